// 2-D P-SV elastic velocity-stress propagator for gfx950 (MI355X): forward, snapshot save and
// the exact discrete adjoint with material-gradient accumulation.  Memory-bound staggered-grid
// stencil (4th order space, leapfrog time, C-PML memory variables); MFMA unused.
//
// Replaces (reference tree): the DENISE-Black-Edition runs behind pyapi_denise's
// `d.forward(...)` / `d.grad(...)` as called at models/networks.py:7787, 9853-9877 (mpirun of 30
// MPI ranks + files on disk in the reference).
//
// Forward: two launches per time step (V: velocities from stresses, S: stresses from the new
// velocities).  A thread owns 4 consecutive cells (16-B lane accesses) and marches RZ rows with
// the z windows in registers; x neighbours come from 8-B halo loads that hit the lines the
// neighbouring lanes fetch.  C-PML memory variables live in compact strips (group-aligned in x,
// row-aligned in z) so that interior tiles never touch them.  Source injection (LDS image of
// the tile, only where the shot's bounding box intersects it) and receiver sampling (extra
// workgroups) ride in the S launch.
// Adjoint: two launches per step (S^T then V^T).  The transposed C-PML acts on the
// material-scaled adjoint fields BEFORE the spatial derivative, so each workgroup first
// evaluates those four per-cell quantities on its tile + 2-cell halo into LDS, then applies the
// stencils from LDS.  All five material-gradient accumulators are updated in the S^T launch and
// stay in registers across the shots of a group (one read-modify-write per group, not per shot).
// Tiles are taken in an XCD-contiguous order (xcd_tile) so that neighbouring tiles share an L2, and the
// drivers sweep the time range a few shots at a time when all shots together would not stay inside the
// 256 MiB Infinity Cache (plan->pass_shots / pass_groups).  Grids whose shot fits the LDS of a few CUs run
// the whole time loop in one launch instead (mifwi_elastic_cluster.h); point forces (source_type 1 / 2) and
// the opt-in fused V+S launch (el_step_fused) live on the per-step path only.
// Arithmetic = the explicit fmaf chain of oracle/elastic.c (build with -ffp-contract=off).
#include "mifwi_common.h"

#include <vector>

#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace {

constexpr int kThreads = 256;
#ifndef MIFWI_EL_MINWAVES
#define MIFWI_EL_MINWAVES 1
#endif
// staggered-grid first-derivative weights (mifwi_elastic_desc.fd_order): Taylor order 4 = (9/8, -1/24),
// order 2 = (1, 0) - the same four-point form, so every kernel serves both
struct FdK { float c1, c2; };
inline FdK fd_weights(int order)
{
    return order == 2 ? FdK{1.0f, 0.0f} : FdK{(float)(9.0 / 8.0), (float)(-1.0 / 24.0)};
}

enum { F_VX = 0, F_VZ = 1, F_SXX = 2, F_SZZ = 3, F_SXZ = 4 };
enum { M_L = 0, M_M = 1, M_MU = 2, M_BX = 3, M_BZ = 4 };
enum { PA = 0, PB = 1, PK = 2, PAH = 3, PBH = 4, PKH = 5 };
// psi index: x strips hold (0: d1 sxx_x @half, 1: d3 sxz_x @int, 2: e1 vx_x @int, 3: e4 vz_x @half)
//            z strips hold (0: d2 sxz_z @int, 1: d4 szz_z @half, 2: e2 vz_z @int, 3: e3 vx_z @half)

struct ElParams {
    int nz, nx, ng, gp, pitch;
    unsigned field_stride;       // (nz+4)*pitch
    long long shot_stride;       // 5*field_stride
    int nshot, gs;
    int s0;                      // first shot of this pass over the time range (blockIdx.z counts from it)
    int W, wl, xr0, wx;          // C-PML strip geometry; W = 0 disables the layer
    int fsurf;                   // 1: stress-imaging free surface on row 0
    long long psix_shot, psiz_shot;   // floats per shot: 4*nz*wx, 4*2W*gp
    const float *mat, *pz, *px;
    float *fields;
    float *fields_out;           // fused forward step: the copy of the state this launch writes
    float *psix, *psiz;          // forward: updated in place; adjoint: read side
    float *psix_out, *psiz_out;  // adjoint: write side (ping-pong)
    float *S;                    // snapshot slice of this step: f32 [nshot][5][nz][gp]; bf16: snap_shot floats per shot
    long long snap_shot;         // floats per shot of one snapshot step (5 * splane for f32)
    unsigned splane;             // floats per snapshot plane: nz*gp, or nz * 64 * ceil(ng / 16) in the column-blocked layout
    int sblk;                    // 1: snapshot planes stored as [column block of 16 groups][row][64 floats] (snap_cell)
    float *acc;                  // [ngroups][5][nz][gp]
    // injection (S launch: sxx,szz += a ; S^T launch: vx += ax, vz += az)
    int ninj, ntap_inj;
    const int *inj_cell;
    const float *inj_w;
    const float *inj_amp0, *inj_amp1;     // [nshot][ninj] of this step (amp1 only in S^T)
    const int *inj_bbox;
    const int *tile_start, *tile_list;   // el_adj_s: receiver taps of each (shot, tile): [nshot][ntiles + 1], [nshot][ninj * ntap_inj]
    // sampling (S launch: vx, vz ; S^T launch: sxx+szz)
    int nsmp, ntap_smp;
    const int *smp_cell;
    const float *smp_w;
    float *smp_out0, *smp_out1;
    int tiles_z;
    int walk_rows;               // el_adj_walk: rows of a column chunk (a multiple of 14)
    float *trash;                // el_adj_walk: 1024 floats that lanes without an owned cell store to (stores stay branch-free)
    long long *walk_trace;       // -DMIFWI_ABLATIONS builds: [block][8] cycles per phase (tools/walk_trace.sh)
    int walk_dbg;                // -DMIFWI_ABLATIONS builds: streams switched off for traffic / timing experiments (wrong results)
    int xcd;                     // XCD-aware tile order (xcd_tile): 1 contiguous runs per shot, 2 whole shots, 3 tile-major over the shots
    FdK K;                       // stencil weights
};

__device__ __forceinline__ float comp(const float4 &v, int c)
{
    return c == 0 ? v.x : c == 1 ? v.y : c == 2 ? v.z : v.w;
}
__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
// uniform base + 32-bit lane offset (in floats): selects the `global_* v_off, s[base]` addressing form - no 64-bit
// vector address per plane and lane
__device__ __forceinline__ const float *el_at(const float *base, unsigned off)
{
    return reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + 4u * off);
}
__device__ __forceinline__ float *el_at(float *base, unsigned off)
{
    return reinterpret_cast<float *>(reinterpret_cast<char *>(base) + 4u * off);
}
// a per-lane value the compiler must treat as unknown here: what is derived from it is recomputed at the use instead of
// being hoisted out of a loop and kept (or spilled) across it
__device__ __forceinline__ int f_opaque(int x)
{
    asm volatile("" : "+v"(x));
    return x;
}
__device__ __forceinline__ float2 ld2(const float *p) { return *reinterpret_cast<const float2 *>(p); }
__device__ __forceinline__ void st4(float *p, const float4 &v) { *reinterpret_cast<float4 *>(p) = v; }
__device__ __forceinline__ float dfw(const FdK &K, float fm1, float f0, float f1, float f2)   // Dp at "0"
{
    return fmaf(K.c1, f1 - f0, K.c2 * (f2 - fm1));
}
__device__ __forceinline__ float dbw(const FdK &K, float fm2, float fm1, float f0, float f1)  // Dm at "0"
{
    return fmaf(K.c1, f0 - fm1, K.c2 * (f1 - fm2));
}
// forward C-PML: psi <- b psi + a d ; returns d*ik + psi
__device__ __forceinline__ float pml(float &psi, float a, float b, float ik, float d)
{
    psi = fmaf(b, psi, a * d);
    return fmaf(d, ik, psi);
}
// transposed C-PML: P = psib + db ; returns ik*db + a*P ; psib <- b*P
__device__ __forceinline__ float pmlT(float psib, float a, float b, float ik, float db,
                                      float &psib_new)
{
    const float P = psib + db;
    psib_new = b * P;
    return fmaf(ik, db, a * P);
}

// ---- bf16 snapshot planes (plan snapshot_format = 1): the five per-step planes the material gradient needs are
// rounded to bf16 (round to nearest even, v_cvt_pk_bf16_f32) on their way to memory and widened again by the
// adjoint: 10 instead of 20 B per cell-step each way.  Arithmetic stays f32.  Layout of one shot-step, in
// 4-cell groups g = (j*gp + 4*i)/4:  [g][S1 x4, S2 x4] 16 B | [g][S4 x4, S5 x4] 16 B | [g][S3 x4] 8 B.
typedef __bf16 mifwi_bf2 __attribute__((ext_vector_type(2)));
typedef float mifwi_f2 __attribute__((ext_vector_type(2)));
typedef unsigned mifwi_u4 __attribute__((ext_vector_type(4)));
typedef unsigned mifwi_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned bf_pack(float a, float b)
{
    return __builtin_bit_cast(unsigned, __builtin_convertvector(mifwi_f2{a, b}, mifwi_bf2));
}
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ void bf_store2(float *base, unsigned grp, const float *a, const float *b)
{
    const mifwi_u4 v{bf_pack(a[0], a[1]), bf_pack(a[2], a[3]), bf_pack(b[0], b[1]), bf_pack(b[2], b[3])};
    __builtin_nontemporal_store(v, reinterpret_cast<mifwi_u4 *>(base) + grp);
}
__device__ __forceinline__ void bf_store1(float *base, unsigned grp, const float *a)
{
    const mifwi_u2 v{bf_pack(a[0], a[1]), bf_pack(a[2], a[3])};
    __builtin_nontemporal_store(v, reinterpret_cast<mifwi_u2 *>(base) + grp);
}
// offsets (in floats) of the three regions of a bf16 shot-step
__device__ __forceinline__ long long bf_reg_de(unsigned ncell) { return (long long)ncell; }
__device__ __forceinline__ long long bf_reg_c(unsigned ncell) { return 2LL * ncell; }
// The packed planes of one group.  Loading and widening are two steps: the loads are requested before a barrier
// and consumed after it, and the widening must stay behind the barrier too (`bf_pin`) - scheduled in front of it,
// the wave would sit out the latency of the snapshot stream, the one stream that always comes from HBM, before
// every barrier (measured: el_adj_s 80 instead of 72 us per launch on 350x1700 although it reads 10 B less).
struct BfPlanes { mifwi_u4 ab, de; mifwi_u2 c; };
__device__ __forceinline__ void bf_request(const float *shot_base, unsigned ncell, unsigned grp, BfPlanes &q)
{
    q.ab = __builtin_nontemporal_load(reinterpret_cast<const mifwi_u4 *>(shot_base) + grp);
    q.de = __builtin_nontemporal_load(reinterpret_cast<const mifwi_u4 *>(shot_base + bf_reg_de(ncell)) + grp);
    q.c = __builtin_nontemporal_load(reinterpret_cast<const mifwi_u2 *>(shot_base + bf_reg_c(ncell)) + grp);
}
__device__ __forceinline__ void bf_pin(BfPlanes &q)
{
    asm volatile("" : "+v"(q.ab), "+v"(q.de), "+v"(q.c));
}
__device__ __forceinline__ void bf_widen(const BfPlanes &q, float4 &S1, float4 &S2, float4 &S3, float4 &S4, float4 &S5)
{
    S1 = make_float4(bf_lo(q.ab.x), bf_hi(q.ab.x), bf_lo(q.ab.y), bf_hi(q.ab.y));
    S2 = make_float4(bf_lo(q.ab.z), bf_hi(q.ab.z), bf_lo(q.ab.w), bf_hi(q.ab.w));
    S4 = make_float4(bf_lo(q.de.x), bf_hi(q.de.x), bf_lo(q.de.y), bf_hi(q.de.y));
    S5 = make_float4(bf_lo(q.de.z), bf_hi(q.de.z), bf_lo(q.de.w), bf_hi(q.de.w));
    S3 = make_float4(bf_lo(q.c.x), bf_hi(q.c.x), bf_lo(q.c.y), bf_hi(q.c.y));
}

// Offset (floats) of group g of row j inside a snapshot plane.  Row-major [nz][gp] (the layout the single-launch
// kernels share), or - per-step family, plans without a single-launch kernel - blocked by columns of 16 groups:
// [g / 16][j][64 floats].  Every kernel of the family works on tiles 16 groups wide, so a tile row is 256 contiguous,
// 256-byte aligned bytes of the plane and consecutive rows follow each other: whole 128-byte lines on the way out
// and on the way back in.  (Row-major rows are gp * 4 bytes apart, not a multiple of 128 in general: a 256-byte tile
// row straddles three lines, and the adjoint read 33 B per cell-step of the 20 B snapshot stream.)
__device__ __forceinline__ unsigned snap_cell(const ElParams &p, int j, int g)
{
    return p.sblk ? ((unsigned)(g >> 4) * (unsigned)p.nz + (unsigned)j) * 64u + 4u * (unsigned)(g & 15)
                  : (unsigned)j * p.gp + 4u * (unsigned)g;
}

// x-strip column offset of group g (or -1)
__device__ __forceinline__ int xstrip(const ElParams &p, int g)
{
    if (p.W == 0) return -1;
    const int c0 = 4 * g;
    if (c0 < p.wl) return c0;
    if (c0 >= p.xr0) return p.wl + (c0 - p.xr0);
    return -1;
}
// z-strip row index of row j (or -1)
__device__ __forceinline__ int zstrip(const ElParams &p, int j)
{
    if (p.W == 0) return -1;
    if (j < p.W) return j;
    if (j >= p.nz - p.W) return j - (p.nz - 2 * p.W);
    return -1;
}

// Workgroups are dealt to the 8 XCDs round-robin in launch order, each XCD with an L2 of its own: with
// the plain blockIdx -> tile map, neighbouring tiles never share an L2 and every halo row/column is
// fetched from memory again.  Remap (a bijection on the launch grid):
//  * z slices (shots / shot groups) in full sets of eight: XCD c takes the slices c, c+8, ... WHOLE, walking
//    each in row-major tile order - every halo a tile reads was fetched by the same L2 one tile row earlier
//    (measured on 350x1700, 8 shots per launch: the halo rows of a 2.7-tile-row run per XCD came from the
//    other XCDs' runs, i.e. from beyond L2);
//  * the remaining slices: each XCD walks one contiguous, row-major run of the (x, y) tiles of the slice.
__device__ __forceinline__ void xcd_tile(const ElParams &p, int &bx, int &by, int &bz)
{
    bx = (int)blockIdx.x; by = (int)blockIdx.y; bz = (int)blockIdx.z;
    if (!p.xcd) return;
    const unsigned gx = gridDim.x, n2 = gx * gridDim.y;
    const unsigned z8 = gridDim.z & ~7u;          // p.xcd == 2: whole slices per XCD
    unsigned T;
    if (p.xcd == 3) {
        // tile-major, slice-minor: the sequence (tile 0: slices 0 .. gz-1), (tile 1: ...), ... is cut into eight
        // contiguous pieces, one per XCD - all shots of the launch pass through a tile back to back on ONE XCD, so the
        // tile's material planes come from memory once per launch, not once per shot (grids whose five planes outgrow the
        // L2s: 60 MB at 1000x3000), and neighbouring tiles still follow each other on that XCD
        const unsigned gz = gridDim.z, total = n2 * gz;
        const unsigned L = blockIdx.x + gx * blockIdx.y + n2 * blockIdx.z;
        const unsigned c = L & 7u, idx = L >> 3, q = total >> 3, r = total & 7u;
        const unsigned e = c * q + (c < r ? c : r) + idx;
        T = e / gz;
        bz = (int)(e - T * gz);
    } else if (p.xcd == 2 && blockIdx.z < z8) {
        const unsigned L = blockIdx.x + gx * blockIdx.y + n2 * blockIdx.z;
        const unsigned c = L & 7u, idx = L >> 3, zl = idx / n2;
        T = idx - zl * n2;
        bz = (int)(c + 8u * zl);
    } else {
        const unsigned L = blockIdx.x + gx * blockIdx.y;
        const unsigned c = L & 7u, idx = L >> 3, q = n2 >> 3, r = n2 & 7u;
        T = c * q + (c < r ? c : r) + idx;
    }
    by = (int)(T / gx); bx = (int)(T - (unsigned)by * gx);
}

// ------------------------------------------------------------------------------------------------
// sampling workgroups.  mode 0: out0 = sum w vx, out1 = sum w vz ; mode 1: out0 = sum w (sxx+szz)
template <int MODE>
__device__ void sample_points(const ElParams &p, int bx, int by, int bz)
{
    if (p.smp_out0 == nullptr) return;
    const int nrb = (int)(gridDim.y - p.tiles_z) * (int)gridDim.x;
    const int rb = (by - p.tiles_z) * (int)gridDim.x + bx;
    const int total = p.gs * p.nsmp;
    for (int e = rb * (int)blockDim.x + (int)threadIdx.x; e < total; e += nrb * (int)blockDim.x) {
        const int si = e / p.nsmp, ip = e - si * p.nsmp;
        const int s = p.s0 + bz * p.gs + si;
        if (s >= p.nshot) continue;
        const float *fl = p.fields + (long long)s * p.shot_stride;
        float a0 = 0.f, a1 = 0.f;
        for (int t = 0; t < p.ntap_smp; ++t) {
            const long long ee = ((long long)s * p.nsmp + ip) * p.ntap_smp + t;
            const int cell = p.smp_cell[ee];
            if (cell < 0) continue;
            const int j = cell / p.nx, i = cell - j * p.nx;
            const unsigned off = (unsigned)(j + 2) * p.pitch + 4 + i;
            const float w = p.smp_w[ee];
            if (MODE == 0) {
                a0 = fmaf(w, fl[F_VX * p.field_stride + off], a0);
                a1 = fmaf(w, fl[F_VZ * p.field_stride + off], a1);
            } else {
                const float zz = (p.fsurf && j == 0) ? 0.f : fl[F_SZZ * p.field_stride + off];
                a0 = fmaf(w, fl[F_SXX * p.field_stride + off] + zz, a0);
            }
        }
        p.smp_out0[(long long)s * p.nsmp + ip] = a0;
        if (MODE == 0) p.smp_out1[(long long)s * p.nsmp + ip] = a1;
    }
}

// stage the shot's injection amplitudes that fall into this tile (block-uniform decision)
template <int TZ, int TX, int NCOMP>
__device__ bool stage_injection(const ElParams &p, int s, int tile_j, int tile_i, float *inj)
{
    if (p.ninj <= 0) return false;
    const int b0 = p.inj_bbox[4 * s + 0], b1 = p.inj_bbox[4 * s + 1];
    const int b2 = p.inj_bbox[4 * s + 2], b3 = p.inj_bbox[4 * s + 3];
    const bool has = (b0 < tile_j + TZ) && (b1 >= tile_j) && (b2 < tile_i + TX) && (b3 >= tile_i);
    if (!has) return false;
    for (int e = (int)threadIdx.x; e < NCOMP * TZ * TX; e += kThreads) inj[e] = 0.f;
    __syncthreads();
    const int total = p.ninj * p.ntap_inj;
    for (int e = (int)threadIdx.x; e < total; e += kThreads) {
        const long long ee = (long long)s * total + e;
        const int cell = p.inj_cell[ee];
        if (cell < 0) continue;
        const int j = cell / p.nx, i = cell - j * p.nx;
        const int tj = j - tile_j, ti = i - tile_i;
        if (tj >= 0 && tj < TZ && ti >= 0 && ti < TX) {
            const long long ai = (long long)s * p.ninj + e / p.ntap_inj;
            const float w = p.inj_w[ee];
            atomicAdd(&inj[tj * TX + ti], w * p.inj_amp0[ai]);
            if (NCOMP == 2) atomicAdd(&inj[TZ * TX + tj * TX + ti], w * p.inj_amp1[ai]);
        }
    }
    __syncthreads();
    return true;
}

// ================================================================================================
// forward V launch:  vx += bxs (Dp_x sxx' + Dm_z sxz'),  vz += bzs (Dm_x sxz' + Dp_z szz')
// ================================================================================================
template <int LX, int RZ, int SAVE>       // SAVE 0: no snapshots, 1: f32 planes, 2: bf16 planes
__global__ __launch_bounds__(kThreads, MIFWI_EL_MINWAVES) void el_step_v(const ElParams p)
{
    const FdK K = p.K;
    constexpr int LZ = kThreads / LX;
    const int lx = (int)threadIdx.x % LX, lz = (int)threadIdx.x / LX;
    int bx, by, bz;
    xcd_tile(p, bx, by, bz);
    const int g = bx * LX + lx;
    const int j0 = (by * LZ + lz) * RZ;
    if (g >= p.ng || j0 >= p.nz) return;
    const int s = p.s0 + bz;
    const unsigned fs = p.field_stride;
    float *fl = p.fields + (long long)s * p.shot_stride;
    const float *sxx = fl + F_SXX * fs, *szz = fl + F_SZZ * fs, *sxz = fl + F_SXZ * fs;
    float *vx = fl + F_VX * fs, *vz = fl + F_VZ * fs;
    const unsigned col = 4 + 4 * g;
    const unsigned ncell = (unsigned)p.nz * p.gp;
    const int xs_off = xstrip(p, g);
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 pxa = zero4, pxb = zero4, pxk = zero4, pxah = zero4, pxbh = zero4, pxkh = zero4;
    if (xs_off >= 0) {
        pxa = ld4(p.px + PA * p.gp + 4 * g); pxb = ld4(p.px + PB * p.gp + 4 * g);
        pxk = ld4(p.px + PK * p.gp + 4 * g); pxah = ld4(p.px + PAH * p.gp + 4 * g);
        pxbh = ld4(p.px + PBH * p.gp + 4 * g); pxkh = ld4(p.px + PKH * p.gp + 4 * g);
    }
    // windows: sxz rows j-2..j+1 (a0..a3), szz rows j-1..j+2 (b0..b3)
    float4 a0, a1, a2, a3, b0, b1, b2, b3;
    {
        const unsigned o = (unsigned)(j0 + 2) * p.pitch + col;     // row j0
        a0 = ld4(sxz + o - 2 * p.pitch); a1 = ld4(sxz + o - p.pitch); a2 = ld4(sxz + o);
        b0 = ld4(szz + o - p.pitch); b1 = ld4(szz + o); b2 = ld4(szz + o + p.pitch);
    }
#pragma unroll
    for (int rz = 0; rz < RZ; ++rz) {
        const int j = j0 + rz;
        if (j < p.nz) {
            const unsigned o = (unsigned)(j + 2) * p.pitch + col;
            const unsigned cc = (unsigned)j * p.gp + 4 * g;
            // every global request of this row up front (one memory round trip per row): the
            // memory variables of the C-PML strips are fetched together with the fields
            const int zs = zstrip(p, j);
            float *q1 = nullptr, *q3 = nullptr, *q2 = nullptr, *q4 = nullptr;
            float4 s1 = zero4, s3 = zero4, s2 = zero4, s4 = zero4;
            float za = 0.f, zb = 0.f, zk = 1.f, zah = 0.f, zbh = 0.f, zkh = 1.f;
            if (xs_off >= 0) {
                q1 = p.psix + (long long)s * p.psix_shot + ((long long)0 * p.nz + j) * p.wx + xs_off;
                q3 = p.psix + (long long)s * p.psix_shot + ((long long)1 * p.nz + j) * p.wx + xs_off;
                s1 = ld4(q1); s3 = ld4(q3);
            }
            if (zs >= 0) {
                q2 = p.psiz + (long long)s * p.psiz_shot + ((long long)0 * 2 * p.W + zs) * p.gp + 4 * g;
                q4 = p.psiz + (long long)s * p.psiz_shot + ((long long)1 * 2 * p.W + zs) * p.gp + 4 * g;
                s2 = ld4(q2); s4 = ld4(q4);
                za = p.pz[PA * p.nz + j]; zb = p.pz[PB * p.nz + j]; zk = p.pz[PK * p.nz + j];
                zah = p.pz[PAH * p.nz + j]; zbh = p.pz[PBH * p.nz + j]; zkh = p.pz[PKH * p.nz + j];
            }
            a3 = ld4(sxz + o + p.pitch);
            b3 = ld4(szz + o + 2 * p.pitch);
            if (p.fsurf && j < 2) {
                // odd mirroring about row 0: sxz(-m) = -sxz(m-1), szz(-m) = -szz(m)
                if (j == 0) {
                    a1 = make_float4(-a2.x, -a2.y, -a2.z, -a2.w);
                    a0 = make_float4(-a3.x, -a3.y, -a3.z, -a3.w);
                    b0 = make_float4(-b2.x, -b2.y, -b2.z, -b2.w);
                } else {
                    a0 = make_float4(-a1.x, -a1.y, -a1.z, -a1.w);
                }
            }
            const float4 cxx = ld4(sxx + o);
            const float2 Lxx = ld2(sxx + o - 2), Rxx = ld2(sxx + o + 4);
            const float2 Lxz = ld2(sxz + o - 2), Rxz = ld2(sxz + o + 4);
            float4 vxv = ld4(vx + o), vzv = ld4(vz + o);
            const float4 bxs = ld4(p.mat + M_BX * ncell + cc), bzs = ld4(p.mat + M_BZ * ncell + cc);
            const float xx[8] = {Lxx.x, Lxx.y, cxx.x, cxx.y, cxx.z, cxx.w, Rxx.x, Rxx.y};
            const float xz[8] = {Lxz.x, Lxz.y, a2.x, a2.y, a2.z, a2.w, Rxz.x, Rxz.y};
            float d1[4], d2[4], d3[4], d4[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                d1[c] = dfw(K, xx[c + 1], xx[c + 2], xx[c + 3], xx[c + 4]);
                d2[c] = dbw(K, comp(a0, c), comp(a1, c), comp(a2, c), comp(a3, c));
                d3[c] = dbw(K, xz[c], xz[c + 1], xz[c + 2], xz[c + 3]);
                d4[c] = dfw(K, comp(b0, c), comp(b1, c), comp(b2, c), comp(b3, c));
            }
            if (xs_off >= 0) {
                float t1[4] = {s1.x, s1.y, s1.z, s1.w}, t3[4] = {s3.x, s3.y, s3.z, s3.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    d1[c] = pml(t1[c], comp(pxah, c), comp(pxbh, c), comp(pxkh, c), d1[c]);
                    d3[c] = pml(t3[c], comp(pxa, c), comp(pxb, c), comp(pxk, c), d3[c]);
                }
                st4(q1, make_float4(t1[0], t1[1], t1[2], t1[3]));
                st4(q3, make_float4(t3[0], t3[1], t3[2], t3[3]));
            }
            if (zs >= 0) {
                float t2[4] = {s2.x, s2.y, s2.z, s2.w}, t4[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    d2[c] = pml(t2[c], za, zb, zk, d2[c]);
                    d4[c] = pml(t4[c], zah, zbh, zkh, d4[c]);
                }
                st4(q2, make_float4(t2[0], t2[1], t2[2], t2[3]));
                st4(q4, make_float4(t4[0], t4[1], t4[2], t4[3]));
            }
            float s4v[4], s5v[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) { s4v[c] = d1[c] + d2[c]; s5v[c] = d3[c] + d4[c]; }
            vxv.x = fmaf(bxs.x, s4v[0], vxv.x); vxv.y = fmaf(bxs.y, s4v[1], vxv.y);
            vxv.z = fmaf(bxs.z, s4v[2], vxv.z); vxv.w = fmaf(bxs.w, s4v[3], vxv.w);
            vzv.x = fmaf(bzs.x, s5v[0], vzv.x); vzv.y = fmaf(bzs.y, s5v[1], vzv.y);
            vzv.z = fmaf(bzs.z, s5v[2], vzv.z); vzv.w = fmaf(bzs.w, s5v[3], vzv.w);
            st4(vx + o, vxv);
            st4(vz + o, vzv);
            if (SAVE == 1) {
                float *Sp = p.S + (long long)s * p.snap_shot + snap_cell(p, j, g);
                mifwi::stnt4(Sp + 3 * (long long)p.splane, make_float4(s4v[0], s4v[1], s4v[2], s4v[3]));
                mifwi::stnt4(Sp + 4 * (long long)p.splane, make_float4(s5v[0], s5v[1], s5v[2], s5v[3]));
            } else if (SAVE == 2) {
                bf_store2(p.S + (long long)s * p.snap_shot + bf_reg_de(p.splane), snap_cell(p, j, g) >> 2, s4v, s5v);
            }
            a0 = a1; a1 = a2; a2 = a3;
            b0 = b1; b1 = b2; b2 = b3;
        }
    }
}

// ================================================================================================
// forward S launch: stresses from the new velocities + source injection + receiver sampling
// ================================================================================================
template <int LX, int RZ, int SAVE>
__global__ __launch_bounds__(kThreads, MIFWI_EL_MINWAVES) void el_step_s(const ElParams p)
{
    const FdK K = p.K;
    constexpr int LZ = kThreads / LX;
    constexpr int TZ = LZ * RZ, TX = LX * 4;
    int bx, by, bz;
    xcd_tile(p, bx, by, bz);
    if (by >= p.tiles_z) {
        sample_points<0>(p, bx, by, bz);
        return;
    }
    __shared__ float inj[TZ * TX];
    const int lx = (int)threadIdx.x % LX, lz = (int)threadIdx.x / LX;
    const int g = bx * LX + lx;
    const int tile_j = by * TZ, tile_i = bx * TX;
    const int j0 = tile_j + lz * RZ;
    const bool active = (g < p.ng) && (j0 < p.nz);
    const unsigned fs = p.field_stride;
    const unsigned col = 4 + 4 * g;
    const unsigned ncell = (unsigned)p.nz * p.gp;
    const int xs_off = active ? xstrip(p, g) : -1;
    float4 pxa, pxb, pxk, pxah, pxbh, pxkh;
    if (xs_off >= 0) {
        pxa = ld4(p.px + PA * p.gp + 4 * g); pxb = ld4(p.px + PB * p.gp + 4 * g);
        pxk = ld4(p.px + PK * p.gp + 4 * g); pxah = ld4(p.px + PAH * p.gp + 4 * g);
        pxbh = ld4(p.px + PBH * p.gp + 4 * g); pxkh = ld4(p.px + PKH * p.gp + 4 * g);
    }
    for (int si = 0; si < p.gs; ++si) {
        const int s = p.s0 + bz * p.gs + si;
        if (s >= p.nshot) break;
        const bool has_inj = stage_injection<TZ, TX, 1>(p, s, tile_j, tile_i, inj);
        if (active) {
            float *fl = p.fields + (long long)s * p.shot_stride;
            const float *vx = fl + F_VX * fs, *vz = fl + F_VZ * fs;
            float *sxx = fl + F_SXX * fs, *szz = fl + F_SZZ * fs, *sxz = fl + F_SXZ * fs;
            // windows: vz rows j-2..j+1 (a0..a3), vx rows j-1..j+2 (b0..b3)
            float4 a0, a1, a2, a3, b0, b1, b2, b3;
            {
                const unsigned o = (unsigned)(j0 + 2) * p.pitch + col;
                a0 = ld4(vz + o - 2 * p.pitch); a1 = ld4(vz + o - p.pitch); a2 = ld4(vz + o);
                b0 = ld4(vx + o - p.pitch); b1 = ld4(vx + o); b2 = ld4(vx + o + p.pitch);
            }
#pragma unroll
            for (int rz = 0; rz < RZ; ++rz) {
                const int j = j0 + rz;
                if (j < p.nz) {
                    const unsigned o = (unsigned)(j + 2) * p.pitch + col;
                    const unsigned cc = (unsigned)j * p.gp + 4 * g;
                    a3 = ld4(vz + o + p.pitch);
                    b3 = ld4(vx + o + 2 * p.pitch);
                    const float2 Lvx = ld2(vx + o - 2), Rvx = ld2(vx + o + 4);
                    const float2 Lvz = ld2(vz + o - 2), Rvz = ld2(vz + o + 4);
                    float4 vxx = ld4(sxx + o), vzz = ld4(szz + o), vxz = ld4(sxz + o);
                    const float4 Ls = ld4(p.mat + M_L * ncell + cc), Ms = ld4(p.mat + M_M * ncell + cc);
                    const float4 mus = ld4(p.mat + M_MU * ncell + cc);
                    const float xv[8] = {Lvx.x, Lvx.y, b1.x, b1.y, b1.z, b1.w, Rvx.x, Rvx.y};
                    const float zv[8] = {Lvz.x, Lvz.y, a2.x, a2.y, a2.z, a2.w, Rvz.x, Rvz.y};
                    float e1[4], e2[4], e3[4], e4[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        e1[c] = dbw(K, xv[c], xv[c + 1], xv[c + 2], xv[c + 3]);
                        e2[c] = dbw(K, comp(a0, c), comp(a1, c), comp(a2, c), comp(a3, c));
                        e3[c] = dfw(K, comp(b0, c), comp(b1, c), comp(b2, c), comp(b3, c));
                        e4[c] = dfw(K, zv[c + 1], zv[c + 2], zv[c + 3], zv[c + 4]);
                    }
                    if (xs_off >= 0) {
                        float *q5 = p.psix + (long long)s * p.psix_shot + ((long long)2 * p.nz + j) * p.wx + xs_off;
                        float *q8 = p.psix + (long long)s * p.psix_shot + ((long long)3 * p.nz + j) * p.wx + xs_off;
                        float4 s5 = ld4(q5), s8 = ld4(q8);
                        float t5[4] = {s5.x, s5.y, s5.z, s5.w}, t8[4] = {s8.x, s8.y, s8.z, s8.w};
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            e1[c] = pml(t5[c], comp(pxa, c), comp(pxb, c), comp(pxk, c), e1[c]);
                            e4[c] = pml(t8[c], comp(pxah, c), comp(pxbh, c), comp(pxkh, c), e4[c]);
                        }
                        st4(q5, make_float4(t5[0], t5[1], t5[2], t5[3]));
                        st4(q8, make_float4(t8[0], t8[1], t8[2], t8[3]));
                    }
                    const int zs = zstrip(p, j);
                    if (zs >= 0) {
                        const float za = p.pz[PA * p.nz + j], zb = p.pz[PB * p.nz + j], zk = p.pz[PK * p.nz + j];
                        const float zah = p.pz[PAH * p.nz + j], zbh = p.pz[PBH * p.nz + j], zkh = p.pz[PKH * p.nz + j];
                        float *q6 = p.psiz + (long long)s * p.psiz_shot + ((long long)2 * 2 * p.W + zs) * p.gp + 4 * g;
                        float *q7 = p.psiz + (long long)s * p.psiz_shot + ((long long)3 * 2 * p.W + zs) * p.gp + 4 * g;
                        float4 s6 = ld4(q6), s7 = ld4(q7);
                        float t6[4] = {s6.x, s6.y, s6.z, s6.w}, t7[4] = {s7.x, s7.y, s7.z, s7.w};
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            e2[c] = pml(t6[c], za, zb, zk, e2[c]);
                            e3[c] = pml(t7[c], zah, zbh, zkh, e3[c]);
                        }
                        st4(q6, make_float4(t6[0], t6[1], t6[2], t6[3]));
                        st4(q7, make_float4(t7[0], t7[1], t7[2], t7[3]));
                    }
                    float nxx[4], nzz[4], nxz[4], s3v[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        s3v[c] = e3[c] + e4[c];
                        nxx[c] = fmaf(comp(Ms, c), e1[c], fmaf(comp(Ls, c), e2[c], comp(vxx, c)));
                        nzz[c] = fmaf(comp(Ls, c), e1[c], fmaf(comp(Ms, c), e2[c], comp(vzz, c)));
                        nxz[c] = fmaf(comp(mus, c), s3v[c], comp(vxz, c));
                        if (has_inj) {
                            const float a = inj[(j - tile_j) * TX + 4 * lx + c];
                            nxx[c] += a;
                            nzz[c] += a;
                        }
                        if (p.fsurf && j == 0) nzz[c] = 0.f;
                    }
                    st4(sxx + o, make_float4(nxx[0], nxx[1], nxx[2], nxx[3]));
                    st4(szz + o, make_float4(nzz[0], nzz[1], nzz[2], nzz[3]));
                    st4(sxz + o, make_float4(nxz[0], nxz[1], nxz[2], nxz[3]));
                    if (SAVE == 1) {
                        float *Sp = p.S + (long long)s * p.snap_shot + snap_cell(p, j, g);
                        mifwi::stnt4(Sp, make_float4(e1[0], e1[1], e1[2], e1[3]));
                        mifwi::stnt4(Sp + (long long)p.splane, make_float4(e2[0], e2[1], e2[2], e2[3]));
                        mifwi::stnt4(Sp + 2 * (long long)p.splane, make_float4(s3v[0], s3v[1], s3v[2], s3v[3]));
                    } else if (SAVE == 2) {
                        float *Sp = p.S + (long long)s * p.snap_shot;
                        bf_store2(Sp, snap_cell(p, j, g) >> 2, e1, e2);
                        bf_store1(Sp + bf_reg_c(p.splane), snap_cell(p, j, g) >> 2, s3v);
                    }
                    a0 = a1; a1 = a2; a2 = a3;
                    b0 = b1; b1 = b2; b2 = b3;
                }
            }
        }
        if (has_inj) __syncthreads();
    }
}

// ================================================================================================
// adjoint launches.  Tile = TZ x (4*GXO) owned cells, staged on (TZ+4) x 4*(GXO+2) in LDS.
// ================================================================================================
constexpr int ATZ = 16;          // owned rows
constexpr int AGO = 16;          // owned groups per row: a quarter wave reads one contiguous LDS row
constexpr int AGX = AGO + 2;     // staged groups per row (1 halo group each side)
constexpr int ASZ = ATZ + 4;     // staged rows
constexpr int ASX = 4 * AGX;     // staged columns (72 floats: consecutive rows start 8 banks apart)
static_assert(ATZ * AGO == kThreads, "one owned 4-cell group per thread");

// x stencils on the 8 values {L.z, L.w, C.x .. C.w, R.x, R.y} around an owned group
struct Row8 { float v[8]; };
__device__ __forceinline__ Row8 row8(const float *row, int cb)
{
    const float4 l = ld4(row + cb - 4), c = ld4(row + cb), r = ld4(row + cb + 4);
    return Row8{{l.z, l.w, c.x, c.y, c.z, c.w, r.x, r.y}};
}

// staged coordinates of the halo group a thread stages besides its own group (threads 0..kHalo-1):
// the two rows above and below the tile, then the left/right halo group of every owned row
constexpr int kHalo = 4 * AGX + 2 * ATZ;
__device__ __forceinline__ void halo_item(int h, int &sr, int &sg)
{
    if (h < 4 * AGX) {
        const int q = h / AGX;
        sr = q < 2 ? q : ATZ + q;
        sg = h - q * AGX;
    } else {
        const int k = h - 4 * AGX;
        sr = 2 + (k >> 1);
        sg = (k & 1) * (AGX - 1);
    }
}

struct AdjIn { float4 a, b, c, m0, m1, m2; };

// E1..E4 of one group: C^T sigma_bar through the transposed C-PML (memory variables written by the owner)
__device__ __forceinline__ void stage_E(const ElParams &p, int s, int j, int g, const AdjIn &in, bool mine,
                                        float4 &E1, float4 &E2, float4 &E3, float4 &E4)
{
    float e1[4], e2[4], e3[4], e4[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        e1[c] = fmaf(comp(in.m1, c), comp(in.a, c), comp(in.m0, c) * comp(in.b, c));
        e2[c] = fmaf(comp(in.m0, c), comp(in.a, c), comp(in.m1, c) * comp(in.b, c));
        e3[c] = comp(in.m2, c) * comp(in.c, c);
        e4[c] = e3[c];
    }
    const int xs_off = xstrip(p, g);
    if (xs_off >= 0) {
        const float4 pxa = ld4(p.px + PA * p.gp + 4 * g), pxb = ld4(p.px + PB * p.gp + 4 * g);
        const float4 pxk = ld4(p.px + PK * p.gp + 4 * g), pxah = ld4(p.px + PAH * p.gp + 4 * g);
        const float4 pxbh = ld4(p.px + PBH * p.gp + 4 * g), pxkh = ld4(p.px + PKH * p.gp + 4 * g);
        const long long q5 = (long long)s * p.psix_shot + ((long long)2 * p.nz + j) * p.wx + xs_off;
        const long long q8 = (long long)s * p.psix_shot + ((long long)3 * p.nz + j) * p.wx + xs_off;
        const float4 s5 = ld4(p.psix + q5), s8 = ld4(p.psix + q8);
        float n5[4], n8[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            e1[c] = pmlT(comp(s5, c), comp(pxa, c), comp(pxb, c), comp(pxk, c), e1[c], n5[c]);
            e4[c] = pmlT(comp(s8, c), comp(pxah, c), comp(pxbh, c), comp(pxkh, c), e4[c], n8[c]);
        }
        if (mine) {
            st4(p.psix_out + q5, make_float4(n5[0], n5[1], n5[2], n5[3]));
            st4(p.psix_out + q8, make_float4(n8[0], n8[1], n8[2], n8[3]));
        }
    }
    const int zs = zstrip(p, j);
    if (zs >= 0) {
        const float za = p.pz[PA * p.nz + j], zb = p.pz[PB * p.nz + j], zk = p.pz[PK * p.nz + j];
        const float zah = p.pz[PAH * p.nz + j], zbh = p.pz[PBH * p.nz + j], zkh = p.pz[PKH * p.nz + j];
        const long long q6 = (long long)s * p.psiz_shot + ((long long)2 * 2 * p.W + zs) * p.gp + 4 * g;
        const long long q7 = (long long)s * p.psiz_shot + ((long long)3 * 2 * p.W + zs) * p.gp + 4 * g;
        const float4 s6 = ld4(p.psiz + q6), s7 = ld4(p.psiz + q7);
        float n6[4], n7[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            e2[c] = pmlT(comp(s6, c), za, zb, zk, e2[c], n6[c]);
            e3[c] = pmlT(comp(s7, c), zah, zbh, zkh, e3[c], n7[c]);
        }
        if (mine) {
            st4(p.psiz_out + q6, make_float4(n6[0], n6[1], n6[2], n6[3]));
            st4(p.psiz_out + q7, make_float4(n7[0], n7[1], n7[2], n7[3]));
        }
    }
    E1 = make_float4(e1[0], e1[1], e1[2], e1[3]); E2 = make_float4(e2[0], e2[1], e2[2], e2[3]);
    E3 = make_float4(e3[0], e3[1], e3[2], e3[3]); E4 = make_float4(e4[0], e4[1], e4[2], e4[3]);
}

// Receiver taps (adjoint sources) of every tile of el_adj_s, one block per shot: start[s][tile .. tile+1) indexes
// list[s][], which holds tap numbers (receiver * ntap + tap) sorted by tile, ascending inside a tile.  Built once
// per backward call; the time loop then injects v_bar += R^T g inside el_adj_s, by the tile that owns the cells
// (workgroup-uniform branch, taken by the few tiles that hold receivers), instead of a launch of its own per step
// and shot pass (4.7 us each: 5 % of a step where the passes are small, e.g. the chunks of a shot-chunked gradient).
__global__ void el_build_tile_taps(const int *cell, int ntaps, int nx, int tx, int ntiles, int *start, int *cursor,
                                   int *list)
{
    const int s = (int)blockIdx.x, t = (int)threadIdx.x, T = (int)blockDim.x;
    start += (long long)s * (ntiles + 1); cursor += (long long)s * ntiles; list += (long long)s * ntaps;
    cell += (long long)s * ntaps;
    for (int k = t; k < ntiles; k += T) cursor[k] = 0;
    __syncthreads();
    auto tile_of = [&](int c) { const int j = c / nx, i = c - j * nx; return (j / ATZ) * tx + (i >> 2) / AGO; };
    for (int e = t; e < ntaps; e += T)
        if (cell[e] >= 0) atomicAdd(cursor + tile_of(cell[e]), 1);
    __syncthreads();
    if (t == 0) {
        int run = 0;
        for (int k = 0; k < ntiles; ++k) { start[k] = run; run += cursor[k]; cursor[k] = start[k]; }
        start[ntiles] = run;
    }
    __syncthreads();
    for (int e = t; e < ntaps; e += T)
        if (cell[e] >= 0) list[atomicAdd(cursor + tile_of(cell[e]), 1)] = e;
    __syncthreads();
    for (int k = t; k < ntiles; k += T)                   // ascending tap order inside a tile (a few entries each)
        for (int a = start[k] + 1; a < start[k + 1]; ++a) {
            const int v = list[a];
            int b = a - 1;
            while (b >= start[k] && list[b] > v) { list[b + 1] = list[b]; --b; }
            list[b + 1] = v;
        }
}

// S^T:  E = C^T sigma_bar through the transposed C-PML;  v_bar -= stencils(E);  all five material-gradient
//       accumulators.  The adjoint sources of the tile's cells (v_bar += R^T g, which the oracle applies at the head
//       of an adjoint step) are added to the loaded v_bar first, through the E planes while they are still empty
//       (p.tile_start set), or have been applied to the state by el_inject_adjsrc.
// Every global load of a shot (own group, halo group, adjoint velocities, the five snapshot planes) is
// requested before the first use: one memory round trip per shot instead of three.
// Register budget pinned at three waves per SIMD (<= 168 VGPRs; 146 in use since the plane loads address as uniform
// base + 32-bit lane offset and the C-PML branches derive their addresses from opaque indices - 207 before): three
// workgroups per CU, 768 workgroup slots.  1000x3000 adjoint pair 885 -> 853-864 us per step.
template <bool BF16>
__global__ __launch_bounds__(kThreads, 3) void el_adj_s(const ElParams p)
{
    const FdK K = p.K;
    int bx, by, bz;
    xcd_tile(p, bx, by, bz);
    if (by >= p.tiles_z) {
        sample_points<1>(p, bx, by, bz);
        return;
    }
    __shared__ float E[4][ASZ][ASX];
    const int tile_j = by * ATZ;
    const int tile_g = bx * AGO;           // first owned group
    const unsigned fs = p.field_stride;
    const unsigned ncell = (unsigned)p.nz * p.gp;
    const int t = (int)threadIdx.x;
    const int orow = t / AGO, ogrp = t % AGO;
    const int oj = tile_j + orow, og = tile_g + ogrp;
    const bool own_ok = oj < p.nz && og < p.ng;
    const unsigned occ = (unsigned)oj * p.gp + 4 * og;
    const unsigned oo = (unsigned)(oj + 2) * p.pitch + 4 + 4 * og;
    const unsigned osc = snap_cell(p, oj, og);          // snapshot and accumulator planes
    int hr = 0, hg = 0;
    if (t < kHalo) halo_item(t, hr, hg);
    const int hj = tile_j - 2 + hr, hgg = tile_g - 1 + hg;
    const bool halo_ok = t < kHalo && hj >= 0 && hj < p.nz && hgg >= 0 && hgg < p.ng;
    const unsigned hcc = (unsigned)hj * p.gp + 4 * hgg;
    const unsigned ho = (unsigned)(hj + 2) * p.pitch + 4 + 4 * hgg;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 acc[5];
    AdjIn own, halo;
    own.m0 = own.m1 = own.m2 = halo.m0 = halo.m1 = halo.m2 = zero4;
    if (own_ok) {
#pragma unroll
        for (int k = 0; k < 5; ++k)          // accumulator planes share the snapshot planes' layout (snap_cell)
            acc[k] = ld4(el_at(p.acc + ((long long)(p.s0 / p.gs + bz) * 5 + k) * p.splane, osc));
        own.m0 = ld4(el_at(p.mat + M_L * ncell, occ)); own.m1 = ld4(el_at(p.mat + M_M * ncell, occ));
        own.m2 = ld4(el_at(p.mat + M_MU * ncell, occ));
    }
    if (halo_ok) {
        halo.m0 = ld4(el_at(p.mat + M_L * ncell, hcc)); halo.m1 = ld4(el_at(p.mat + M_M * ncell, hcc));
        halo.m2 = ld4(el_at(p.mat + M_MU * ncell, hcc));
    }
    for (int si = 0; si < p.gs; ++si) {
        const int s = p.s0 + bz * p.gs + si;
        if (s >= p.nshot) break;
        float *fl = p.fields + (long long)s * p.shot_stride;
        float4 vxb = zero4, vzb = zero4, S1 = zero4, S2 = zero4, S3 = zero4, S4 = zero4, S5 = zero4;
        BfPlanes packed;
        packed.ab = packed.de = mifwi_u4{0u, 0u, 0u, 0u};
        packed.c = mifwi_u2{0u, 0u};
        own.a = own.b = own.c = halo.a = halo.b = halo.c = zero4;
        if (own_ok) {
            own.a = ld4(el_at(fl + F_SXX * fs, oo)); own.b = ld4(el_at(fl + F_SZZ * fs, oo)); own.c = ld4(el_at(fl + F_SXZ * fs, oo));
        }
        if (halo_ok) {
            halo.a = ld4(el_at(fl + F_SXX * fs, ho)); halo.b = ld4(el_at(fl + F_SZZ * fs, ho)); halo.c = ld4(el_at(fl + F_SXZ * fs, ho));
        }
        if (own_ok) { vxb = ld4(el_at(fl + F_VX * fs, oo)); vzb = ld4(el_at(fl + F_VZ * fs, oo)); }
        // ---- adjoint sources of this tile (workgroup-uniform branch; E[0], E[1] are free until the staging) -----
        if (p.tile_start != nullptr) {
            const int ntiles = (int)gridDim.x * p.tiles_z, per = p.ninj * p.ntap_inj;
            const int *ts = p.tile_start + (long long)s * (ntiles + 1) + by * (int)gridDim.x + bx;
            const int e0 = ts[0], e1 = ts[1];
            if (e1 > e0) {
                const int x = 4 * (ogrp + 1);
                st4(&E[0][orow + 2][x], zero4); st4(&E[1][orow + 2][x], zero4);
                __syncthreads();
                for (int e = e0 + t; e < e1; e += kThreads) {
                    const int tap = p.tile_list[(long long)s * per + e];
                    const int cell = p.inj_cell[(long long)s * per + tap];
                    const int j = cell / p.nx, i = cell - j * p.nx;
                    const float w = p.inj_w[(long long)s * per + tap];
                    const long long ai = (long long)s * p.ninj + tap / p.ntap_inj;
                    atomicAdd(&E[0][j - tile_j + 2][i - 4 * tile_g + 4], w * p.inj_amp0[ai]);
                    atomicAdd(&E[1][j - tile_j + 2][i - 4 * tile_g + 4], w * p.inj_amp1[ai]);
                }
                __syncthreads();
                const float4 ix = ld4(&E[0][orow + 2][x]), iz = ld4(&E[1][orow + 2][x]);
                vxb = make_float4(vxb.x + ix.x, vxb.y + ix.y, vxb.z + ix.z, vxb.w + ix.w);
                vzb = make_float4(vzb.x + iz.x, vzb.y + iz.y, vzb.z + iz.z, vzb.w + iz.w);
                __syncthreads();
            }
        }
        // ---- stage E1..E4 on the tile + halo -------------------------------------------------
        if (p.fsurf && oj == 0) own.b = zero4;            // adjoint of szz(0,.) is discarded
        if (p.fsurf && hj == 0) halo.b = zero4;
        {
            float4 E1 = zero4, E2 = zero4, E3 = zero4, E4 = zero4;
            if (own_ok) stage_E(p, s, f_opaque(oj), f_opaque(og), own, true, E1, E2, E3, E4);
            const int x = 4 * (ogrp + 1);
            st4(&E[0][orow + 2][x], E1); st4(&E[1][orow + 2][x], E2);
            st4(&E[2][orow + 2][x], E3); st4(&E[3][orow + 2][x], E4);
        }
        if (t < kHalo) {
            float4 E1 = zero4, E2 = zero4, E3 = zero4, E4 = zero4;
            if (halo_ok) stage_E(p, s, f_opaque(hj), f_opaque(hgg), halo, false, E1, E2, E3, E4);
            st4(&E[0][hr][4 * hg], E1); st4(&E[1][hr][4 * hg], E2);
            st4(&E[2][hr][4 * hg], E3); st4(&E[3][hr][4 * hg], E4);
        }
        // the snapshot planes are only needed after the barrier: requested last, they do not delay the staging
        if (own_ok && !BF16) {
            const float *Sp = p.S + (long long)s * p.snap_shot;
            const long long sp = p.splane;
            S1 = mifwi::ldnt4(el_at(Sp, osc)); S2 = mifwi::ldnt4(el_at(Sp + sp, osc)); S3 = mifwi::ldnt4(el_at(Sp + 2 * sp, osc));
            S4 = mifwi::ldnt4(el_at(Sp + 3 * sp, osc)); S5 = mifwi::ldnt4(el_at(Sp + 4 * sp, osc));
        } else if (own_ok) {
            bf_request(p.S + (long long)s * p.snap_shot, p.splane, osc >> 2, packed);
        }
        __syncthreads();
        // ---- stencils from LDS + injection + gradient accumulation ---------------------------
        if (own_ok) {
            const int r = orow + 2, cb = 4 * (ogrp + 1);
            const float4 bxx = own.a, bzz = own.b, bxz = own.c;
            float nvx[4], nvz[4];
            const Row8 x1 = row8(&E[0][r][0], cb), x4 = row8(&E[3][r][0], cb);
            const float4 z3a = ld4(&E[2][r - 2][cb]), z3b = ld4(&E[2][r - 1][cb]);
            const float4 z3c = ld4(&E[2][r][cb]), z3d = ld4(&E[2][r + 1][cb]);
            const float4 z2a = ld4(&E[1][r - 1][cb]), z2b = ld4(&E[1][r][cb]);
            const float4 z2c = ld4(&E[1][r + 1][cb]), z2d = ld4(&E[1][r + 2][cb]);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float dx1 = dfw(K, x1.v[c + 1], x1.v[c + 2], x1.v[c + 3], x1.v[c + 4]);
                const float dz3 = dbw(K, comp(z3a, c), comp(z3b, c), comp(z3c, c), comp(z3d, c));
                const float dz2 = dfw(K, comp(z2a, c), comp(z2b, c), comp(z2c, c), comp(z2d, c));
                const float dx4 = dbw(K, x4.v[c], x4.v[c + 1], x4.v[c + 2], x4.v[c + 3]);
                float ax = comp(vxb, c) - (dx1 + dz3);
                float az = comp(vzb, c) - (dz2 + dx4);
                if (4 * og + c >= p.nx) { ax = 0.f; az = 0.f; }
                nvx[c] = ax; nvz[c] = az;
            }
            st4(el_at(fl + F_VX * fs, oo), make_float4(nvx[0], nvx[1], nvx[2], nvx[3]));
            st4(el_at(fl + F_VZ * fs, oo), make_float4(nvz[0], nvz[1], nvz[2], nvz[3]));
            // gradients (oracle order): Ms, Ls, mus from sigma_bar; bxs, bzs from the new v_bar
            if (BF16) {
                bf_pin(packed);
                bf_widen(packed, S1, S2, S3, S4, S5);
            }
#define ACC3(dst, a, b, c_, d) dst = fmaf(a, b, fmaf(c_, d, dst))
            ACC3(acc[M_M].x, S1.x, bxx.x, S2.x, bzz.x); ACC3(acc[M_M].y, S1.y, bxx.y, S2.y, bzz.y);
            ACC3(acc[M_M].z, S1.z, bxx.z, S2.z, bzz.z); ACC3(acc[M_M].w, S1.w, bxx.w, S2.w, bzz.w);
            ACC3(acc[M_L].x, S2.x, bxx.x, S1.x, bzz.x); ACC3(acc[M_L].y, S2.y, bxx.y, S1.y, bzz.y);
            ACC3(acc[M_L].z, S2.z, bxx.z, S1.z, bzz.z); ACC3(acc[M_L].w, S2.w, bxx.w, S1.w, bzz.w);
#undef ACC3
            acc[M_MU].x = fmaf(S3.x, bxz.x, acc[M_MU].x); acc[M_MU].y = fmaf(S3.y, bxz.y, acc[M_MU].y);
            acc[M_MU].z = fmaf(S3.z, bxz.z, acc[M_MU].z); acc[M_MU].w = fmaf(S3.w, bxz.w, acc[M_MU].w);
            acc[M_BX].x = fmaf(S4.x, nvx[0], acc[M_BX].x); acc[M_BX].y = fmaf(S4.y, nvx[1], acc[M_BX].y);
            acc[M_BX].z = fmaf(S4.z, nvx[2], acc[M_BX].z); acc[M_BX].w = fmaf(S4.w, nvx[3], acc[M_BX].w);
            acc[M_BZ].x = fmaf(S5.x, nvz[0], acc[M_BZ].x); acc[M_BZ].y = fmaf(S5.y, nvz[1], acc[M_BZ].y);
            acc[M_BZ].z = fmaf(S5.z, nvz[2], acc[M_BZ].z); acc[M_BZ].w = fmaf(S5.w, nvz[3], acc[M_BZ].w);
        }
        __syncthreads();          // E is reused by the next shot
    }
    if (own_ok) {
#pragma unroll
        for (int k = 0; k < 5; ++k)
            st4(el_at(p.acc + ((long long)(p.s0 / p.gs + bz) * 5 + k) * p.splane, osc), acc[k]);
    }
}

// D1..D4 of one group: B^T v_bar through the transposed C-PML
__device__ __forceinline__ void stage_D(const ElParams &p, int s, int j, int g, const float4 &vxb, const float4 &vzb,
                                        const float4 &bxs, const float4 &bzs, bool mine,
                                        float4 &D1, float4 &D2, float4 &D3, float4 &D4)
{
    float d1[4], d2[4], d3[4], d4[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        d1[c] = comp(bxs, c) * comp(vxb, c); d2[c] = d1[c];
        d3[c] = comp(bzs, c) * comp(vzb, c); d4[c] = d3[c];
    }
    const int xs_off = xstrip(p, g);
    if (xs_off >= 0) {
        const float4 pxa = ld4(p.px + PA * p.gp + 4 * g), pxb = ld4(p.px + PB * p.gp + 4 * g);
        const float4 pxk = ld4(p.px + PK * p.gp + 4 * g), pxah = ld4(p.px + PAH * p.gp + 4 * g);
        const float4 pxbh = ld4(p.px + PBH * p.gp + 4 * g), pxkh = ld4(p.px + PKH * p.gp + 4 * g);
        const long long q1 = (long long)s * p.psix_shot + ((long long)0 * p.nz + j) * p.wx + xs_off;
        const long long q3 = (long long)s * p.psix_shot + ((long long)1 * p.nz + j) * p.wx + xs_off;
        const float4 s1 = ld4(p.psix + q1), s3 = ld4(p.psix + q3);
        float n1[4], n3[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            d1[c] = pmlT(comp(s1, c), comp(pxah, c), comp(pxbh, c), comp(pxkh, c), d1[c], n1[c]);
            d3[c] = pmlT(comp(s3, c), comp(pxa, c), comp(pxb, c), comp(pxk, c), d3[c], n3[c]);
        }
        if (mine) {
            st4(p.psix_out + q1, make_float4(n1[0], n1[1], n1[2], n1[3]));
            st4(p.psix_out + q3, make_float4(n3[0], n3[1], n3[2], n3[3]));
        }
    }
    const int zs = zstrip(p, j);
    if (zs >= 0) {
        const float za = p.pz[PA * p.nz + j], zb = p.pz[PB * p.nz + j], zk = p.pz[PK * p.nz + j];
        const float zah = p.pz[PAH * p.nz + j], zbh = p.pz[PBH * p.nz + j], zkh = p.pz[PKH * p.nz + j];
        const long long q2 = (long long)s * p.psiz_shot + ((long long)0 * 2 * p.W + zs) * p.gp + 4 * g;
        const long long q4 = (long long)s * p.psiz_shot + ((long long)1 * 2 * p.W + zs) * p.gp + 4 * g;
        const float4 s2 = ld4(p.psiz + q2), s4 = ld4(p.psiz + q4);
        float n2[4], n4[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            d2[c] = pmlT(comp(s2, c), za, zb, zk, d2[c], n2[c]);
            d4[c] = pmlT(comp(s4, c), zah, zbh, zkh, d4[c], n4[c]);
        }
        if (mine) {
            st4(p.psiz_out + q2, make_float4(n2[0], n2[1], n2[2], n2[3]));
            st4(p.psiz_out + q4, make_float4(n4[0], n4[1], n4[2], n4[3]));
        }
    }
    D1 = make_float4(d1[0], d1[1], d1[2], d1[3]); D2 = make_float4(d2[0], d2[1], d2[2], d2[3]);
    D3 = make_float4(d3[0], d3[1], d3[2], d3[3]); D4 = make_float4(d4[0], d4[1], d4[2], d4[3]);
}

// V^T:  D = B^T v_bar through the transposed C-PML;  sigma_bar -= stencils(D)
__global__ __launch_bounds__(kThreads) void el_adj_v(const ElParams p)
{
    const FdK K = p.K;
    __shared__ float D[4][ASZ][ASX];
    int bx, by, bz;
    xcd_tile(p, bx, by, bz);
    const int tile_j = by * ATZ;
    const int tile_g = bx * AGO;
    const int s = p.s0 + bz;
    const unsigned fs = p.field_stride;
    const unsigned ncell = (unsigned)p.nz * p.gp;
    const int t = (int)threadIdx.x;
    float *fl = p.fields + (long long)s * p.shot_stride;
    const int orow = t / AGO, ogrp = t % AGO;
    const int oj = tile_j + orow, og = tile_g + ogrp;
    const bool own_ok = oj < p.nz && og < p.ng;
    const unsigned occ = (unsigned)oj * p.gp + 4 * og;
    const unsigned oo = (unsigned)(oj + 2) * p.pitch + 4 + 4 * og;
    int hr = 0, hg = 0;
    if (t < kHalo) halo_item(t, hr, hg);
    const int hj = tile_j - 2 + hr, hgg = tile_g - 1 + hg;
    const bool halo_ok = t < kHalo && hj >= 0 && hj < p.nz && hgg >= 0 && hgg < p.ng;
    const unsigned hcc = (unsigned)hj * p.gp + 4 * hgg;
    const unsigned ho = (unsigned)(hj + 2) * p.pitch + 4 + 4 * hgg;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 ovx = zero4, ovz = zero4, obx = zero4, obz = zero4, hvx = zero4, hvz = zero4, hbx = zero4, hbz = zero4;
    float4 bxx = zero4, bzz = zero4, bxz = zero4;
    if (own_ok) {
        ovx = ld4(fl + F_VX * fs + oo); ovz = ld4(fl + F_VZ * fs + oo);
        obx = ld4(p.mat + M_BX * ncell + occ); obz = ld4(p.mat + M_BZ * ncell + occ);
    }
    if (halo_ok) {
        hvx = ld4(fl + F_VX * fs + ho); hvz = ld4(fl + F_VZ * fs + ho);
        hbx = ld4(p.mat + M_BX * ncell + hcc); hbz = ld4(p.mat + M_BZ * ncell + hcc);
    }
    if (own_ok) {
        bxx = ld4(fl + F_SXX * fs + oo); bzz = ld4(fl + F_SZZ * fs + oo); bxz = ld4(fl + F_SXZ * fs + oo);
    }
    {
        float4 D1 = zero4, D2 = zero4, D3 = zero4, D4 = zero4;
        if (own_ok) stage_D(p, s, oj, og, ovx, ovz, obx, obz, true, D1, D2, D3, D4);
        const int x = 4 * (ogrp + 1);
        st4(&D[0][orow + 2][x], D1); st4(&D[1][orow + 2][x], D2);
        st4(&D[2][orow + 2][x], D3); st4(&D[3][orow + 2][x], D4);
    }
    if (t < kHalo) {
        float4 D1 = zero4, D2 = zero4, D3 = zero4, D4 = zero4;
        if (halo_ok) stage_D(p, s, hj, hgg, hvx, hvz, hbx, hbz, false, D1, D2, D3, D4);
        st4(&D[0][hr][4 * hg], D1); st4(&D[1][hr][4 * hg], D2);
        st4(&D[2][hr][4 * hg], D3); st4(&D[3][hr][4 * hg], D4);
    }
    __syncthreads();
    if (own_ok) {
        const int r = orow + 2, cb = 4 * (ogrp + 1);
        float nxx[4], nzz[4], nxz[4];
        const Row8 x1 = row8(&D[0][r][0], cb), x3 = row8(&D[2][r][0], cb);
        const float4 z2a = ld4(&D[1][r - 1][cb]), z2b = ld4(&D[1][r][cb]);
        const float4 z2c = ld4(&D[1][r + 1][cb]), z2d = ld4(&D[1][r + 2][cb]);
        const float4 z4a = ld4(&D[3][r - 2][cb]), z4b = ld4(&D[3][r - 1][cb]);
        const float4 z4c = ld4(&D[3][r][cb]), z4d = ld4(&D[3][r + 1][cb]);
        float4 m12 = zero4, m13 = zero4, m32 = zero4;
        if (p.fsurf && oj < 2) { m12 = ld4(&D[1][2][cb]); m13 = ld4(&D[1][3][cb]); m32 = ld4(&D[3][2][cb]); }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float dx1 = dbw(K, x1.v[c], x1.v[c + 1], x1.v[c + 2], x1.v[c + 3]);
            const float dz2 = dfw(K, comp(z2a, c), comp(z2b, c), comp(z2c, c), comp(z2d, c));
            const float dx3 = dfw(K, x3.v[c + 1], x3.v[c + 2], x3.v[c + 3], x3.v[c + 4]);
            const float dz4 = dbw(K, comp(z4a, c), comp(z4b, c), comp(z4c, c), comp(z4d, c));
            nxx[c] = comp(bxx, c) - dx1;
            nxz[c] = comp(bxz, c) - (dz2 + dx3);
            nzz[c] = comp(bzz, c) - dz4;
            if (p.fsurf && oj < 2) {
                // transposed odd mirroring (tile_j == 0 here: staged row 2 is grid row 0)
                if (oj == 0) nxz[c] = nxz[c] + fmaf(K.c1, comp(m12, c), K.c2 * comp(m13, c));
                else { nxz[c] = nxz[c] + K.c2 * comp(m12, c); nzz[c] = nzz[c] + K.c2 * comp(m32, c); }
            }
            if (4 * og + c >= p.nx) { nxx[c] = 0.f; nxz[c] = 0.f; nzz[c] = 0.f; }
        }
        st4(fl + F_SXX * fs + oo, make_float4(nxx[0], nxx[1], nxx[2], nxx[3]));
        st4(fl + F_SZZ * fs + oo, make_float4(nzz[0], nzz[1], nzz[2], nzz[3]));
        st4(fl + F_SXZ * fs + oo, make_float4(nxz[0], nxz[1], nxz[2], nxz[3]));
    }
}

#include "mifwi_elastic_fused.h"
#include "mifwi_elastic_walk.h"

// receivers of the state in p.fields (the last step of a fused range)
__global__ __launch_bounds__(kThreads) void el_sample_v(const ElParams p)
{
    sample_points<0>(p, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z);
}

// ---- point forces (source_type 1 / 2): a handful of cells per shot, launches of their own ---------------------
// forward: v[cell] += w * f[n] between V and S.  One thread per shot walks its sources in order (several
// sources may share a cell: no atomics, the sum order is fixed).
__global__ void el_inject_force(const ElParams p, int comp)
{
    const int s = p.s0 + (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (s >= p.nshot || s >= p.s0 + p.gs) return;
    float *v = p.fields + (long long)s * p.shot_stride + (comp == 1 ? F_VX : F_VZ) * (long long)p.field_stride;
    for (int e = 0; e < p.ninj * p.ntap_inj; ++e) {
        const long long ee = (long long)s * p.ninj * p.ntap_inj + e;
        const int cell = p.inj_cell[ee];
        if (cell < 0) continue;
        const int j = cell / p.nx, i = cell - j * p.nx;
        v[(unsigned)(j + 2) * p.pitch + 4 + i] += p.inj_w[ee] * p.inj_amp0[(long long)s * p.ninj + e / p.ntap_inj];
    }
}

// adjoint: grad_f[n] = sum w * v_bar[cell], sampled between S^T and V^T
__global__ void el_sample_force(const ElParams p, int comp)
{
    const int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x);     // over the p.gs shots of this pass
    if (idx >= p.gs * p.nsmp) return;
    const int s = p.s0 + idx / p.nsmp;
    if (s >= p.nshot) return;
    const int e = s * p.nsmp + idx % p.nsmp;
    const float *v = p.fields + (long long)s * p.shot_stride + (comp == 1 ? F_VX : F_VZ) * (long long)p.field_stride;
    float a = 0.f;
    for (int t = 0; t < p.ntap_smp; ++t) {
        const long long ee = (long long)e * p.ntap_smp + t;
        const int cell = p.smp_cell[ee];
        if (cell < 0) continue;
        const int j = cell / p.nx, i = cell - j * p.nx;
        a = fmaf(p.smp_w[ee], v[(unsigned)(j + 2) * p.pitch + 4 + i], a);
    }
    p.smp_out0[e] = a;
}

// ---- pressure receivers (desc.record_pressure): launches of their own on the per-step path -----------------
// forward: rec_p[n] = sum w (sxx + szz)[cell] after S and the source term, for the p.gs shots of this pass
__global__ void el_sample_pressure(const ElParams p)
{
    const int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (idx >= p.gs * p.nsmp) return;
    const int s = p.s0 + idx / p.nsmp;
    if (s >= p.nshot) return;
    const int e = s * p.nsmp + idx % p.nsmp;
    const float *fl = p.fields + (long long)s * p.shot_stride;
    float a = 0.f;
    for (int t = 0; t < p.ntap_smp; ++t) {
        const long long ee = (long long)e * p.ntap_smp + t;
        const int cell = p.smp_cell[ee];
        if (cell < 0) continue;
        const int j = cell / p.nx, i = cell - j * p.nx;
        const unsigned off = (unsigned)(j + 2) * p.pitch + 4 + i;
        a = fmaf(p.smp_w[ee], fl[F_SXX * (long long)p.field_stride + off] + fl[F_SZZ * (long long)p.field_stride + off], a);
    }
    p.smp_out0[e] = a;
}

// adjoint: sxx_bar, szz_bar [cell] += w g_p[n] before S^T (receivers normally sit on distinct cells; taps that
// share a cell add in hardware order, as in the LDS staging of the velocity receivers)
__global__ void el_inject_pressure(const ElParams p)
{
    const int per = p.ninj * p.ntap_inj;
    const int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (idx >= p.gs * per) return;
    const int s = p.s0 + idx / per, r = idx % per;
    if (s >= p.nshot) return;
    const long long ee = (long long)s * per + r;
    const int cell = p.inj_cell[ee];
    if (cell < 0) return;
    const int j = cell / p.nx, i = cell - j * p.nx;
    const unsigned off = (unsigned)(j + 2) * p.pitch + 4 + i;
    float *fl = p.fields + (long long)s * p.shot_stride;
    const float a = p.inj_w[ee] * p.inj_amp0[(long long)s * p.ninj + r / p.ntap_inj];
    atomicAdd(fl + F_SXX * (long long)p.field_stride + off, a);
    atomicAdd(fl + F_SZZ * (long long)p.field_stride + off, a);
}

// adjoint of the velocity receivers: vx_bar, vz_bar [cell] += w g[n] at the head of adjoint step n, before S^T
// (oracle/elastic.c, "a. receivers^T").  Receivers normally sit on distinct cells; taps that share a cell add
// in hardware order.
__global__ void el_inject_adjsrc(const ElParams p)
{
    const int per = p.ninj * p.ntap_inj;
    const int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (idx >= p.gs * per) return;
    const int s = p.s0 + idx / per, r = idx % per;
    if (s >= p.nshot) return;
    const long long ee = (long long)s * per + r;
    const int cell = p.inj_cell[ee];
    if (cell < 0) return;
    const int j = cell / p.nx, i = cell - j * p.nx;
    const unsigned off = (unsigned)(j + 2) * p.pitch + 4 + i;
    float *fl = p.fields + (long long)s * p.shot_stride;
    const float w = p.inj_w[ee];
    const long long ai = (long long)s * p.ninj + r / p.ntap_inj;
    atomicAdd(fl + F_VX * (long long)p.field_stride + off, w * p.inj_amp0[ai]);
    atomicAdd(fl + F_VZ * (long long)p.field_stride + off, w * p.inj_amp1[ai]);
}

__global__ void el_points_bbox(const int *cell, int npts_per_shot, int n1, int *bbox)
{
    __shared__ int red[4][kThreads];
    const int s = blockIdx.x;
    int a0 = 0x7fffffff, a1 = -1, b0 = 0x7fffffff, b1 = -1;
    for (int e = threadIdx.x; e < npts_per_shot; e += kThreads) {
        const int c = cell[(long long)s * npts_per_shot + e];
        if (c < 0) continue;
        const int i0 = c / n1, i1 = c - i0 * n1;
        a0 = min(a0, i0); a1 = max(a1, i0); b0 = min(b0, i1); b1 = max(b1, i1);
    }
    red[0][threadIdx.x] = a0; red[1][threadIdx.x] = a1;
    red[2][threadIdx.x] = b0; red[3][threadIdx.x] = b1;
    __syncthreads();
    for (int st = kThreads / 2; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) {
            red[0][threadIdx.x] = min(red[0][threadIdx.x], red[0][threadIdx.x + st]);
            red[1][threadIdx.x] = max(red[1][threadIdx.x], red[1][threadIdx.x + st]);
            red[2][threadIdx.x] = min(red[2][threadIdx.x], red[2][threadIdx.x + st]);
            red[3][threadIdx.x] = max(red[3][threadIdx.x], red[3][threadIdx.x + st]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        bbox[4 * s + 0] = red[0][0]; bbox[4 * s + 1] = red[1][0];
        bbox[4 * s + 2] = red[2][0]; bbox[4 * s + 3] = red[3][0];
    }
}

// grad[k][cell] = sum over shot groups of acc[group][k][cell]; grad is row-major [5][nz][gp] (the C-ABI's layout), the
// accumulator planes are `aplane` floats apart and column-blocked when sblk (snap_cell)
__global__ void el_finalize(const float *acc, int ngroups, int nz, int gp, long long aplane, int sblk, float *grad)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x, ncell = (long long)nz * gp;
    if (idx >= 5 * ncell) return;
    const int k = (int)(idx / ncell);
    const long long c = idx - (long long)k * ncell;
    const int j = (int)(c / gp), i = (int)(c - (long long)j * gp), g = i >> 2;
    const long long off = sblk ? ((long long)(g >> 4) * nz + j) * 64 + 4 * (g & 15) + (i & 3) : c;
    float a = 0.f;
    for (int gidx = 0; gidx < ngroups; ++gidx) a += acc[((long long)gidx * 5 + k) * aplane + off];
    grad[idx] = a;
}

#include "mifwi_elastic_cluster.h"

int env_int(const char *name, int dflt)
{
    const char *v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

}  // namespace

// ================================================================================================
struct mifwi_elastic_plan {
    mifwi_elastic_desc d;
    int device;
    int ng, gp, pitch, lx, rz, gs, ngroups;
    int W, wl, xr0, wx, xcd;
    float *rec_p;          // pressure receivers bound by mifwi_elastic_plan_bind_pressure (or null)
    const float *g_p;
    int snap_bf16;         // snapshot planes as bf16 (per-step kernels on both sides only)
    long long snap_shot;   // floats per shot of one snapshot step
    long long splane;      // floats per snapshot plane
    int sblk;              // snapshot planes blocked by columns of 16 groups (snap_cell)
    int pass_shots;        // forward per-step family: shots per pass over the time range
    int pass_groups;       // adjoint per-step family: shot groups per pass
    int fused;             // forward V+S in one launch (second copy of the state in the work buffer)
    int fused_pass_shots;  // shots per pass of the fused forward (both copies of the state in the Infinity Cache)
    int fused_adj;         // adjoint S^T+V^T in one launch (second copy of the adjoint fields in the work buffer):
                           // 1 = el_adj_fused (16 x 64 tiles, recomputed z halo), 2 = el_adj_walk (column walk)
    int walk_rows, walk_chunks;   // el_adj_walk: rows per column chunk, chunks per column
    long long field_stride, shot_stride, fields_elems, psix_elems, psiz_elems, coef_elems;
    long long psi_elems;  // psix+psiz rounded up to 64
    // cluster path (LDS-resident time loop); 0 when a shot does not fit
    int cluster, NW, PL, fwd_PL, adj_PL, cl_shots, cl_lds, cl_ng;
    int cl_xh, adj_xh;           // single-launch kernels fetch the x-stencils' outer cells from the neighbouring lanes (ec_xhalo)
    // adjoint cluster kernel (its own slab count: different LDS footprint)
    int cl_adj, adj_NW, adj_shots, adj_lds, adj_ng, adj_zrows;
    long long xbuf_elems, list_elems, xcc_elems;
    long long tile_elems;                  // per-tile receiver lists of el_adj_s (ints, in floats)
};

namespace {

ElParams el_base(const mifwi_elastic_plan *pl, const float *mat, const float *pz, const float *px)
{
    ElParams p;
    memset(&p, 0, sizeof(p));
    p.nz = pl->d.nz; p.nx = pl->d.nx; p.ng = pl->ng; p.gp = pl->gp; p.pitch = pl->pitch;
    p.field_stride = (unsigned)pl->field_stride; p.shot_stride = pl->shot_stride;
    p.nshot = pl->d.nshot; p.gs = 1;
    p.W = pl->W; p.wl = pl->wl; p.xr0 = pl->xr0; p.wx = pl->wx;
    p.fsurf = pl->d.free_surface;
    p.psix_shot = 4LL * pl->d.nz * pl->wx; p.psiz_shot = 4LL * 2 * pl->W * pl->gp;
    p.mat = mat; p.pz = pz; p.px = px;
    p.xcd = pl->xcd;
    p.snap_shot = pl->snap_shot;
    p.splane = (unsigned)pl->splane; p.sblk = pl->sblk;
    p.K = fd_weights(pl->d.fd_order);
    return p;
}

template <int SAVE>
void launch_v(const mifwi_elastic_plan *pl, const ElParams &p, int nshot, hipStream_t st)
{
    const int lz = kThreads / pl->lx;
    dim3 grid(mifwi::ceil_div(pl->ng, pl->lx), mifwi::ceil_div(pl->d.nz, lz), nshot);
    dim3 block(kThreads);
    switch (pl->lx) {
        case 64: hipLaunchKernelGGL((el_step_v<64, 1, SAVE>), grid, block, 0, st, p); break;
        case 32: hipLaunchKernelGGL((el_step_v<32, 1, SAVE>), grid, block, 0, st, p); break;
        default: hipLaunchKernelGGL((el_step_v<16, 1, SAVE>), grid, block, 0, st, p); break;
    }
}

template <int SAVE>
void launch_s(const mifwi_elastic_plan *pl, const ElParams &p0, int nshot, hipStream_t st)
{
    const int lz = kThreads / pl->lx;
    const int tiles_x = mifwi::ceil_div(pl->ng, pl->lx);
    ElParams p = p0;
    p.tiles_z = mifwi::ceil_div(pl->d.nz, lz);
    int extra = 0;
    if (p.smp_out0 != nullptr && p.nsmp > 0)
        extra = mifwi::ceil_div(mifwi::ceil_div(p.gs * p.nsmp, kThreads), tiles_x);
    dim3 grid(tiles_x, p.tiles_z + extra, mifwi::ceil_div(nshot, p.gs)), block(kThreads);
    switch (pl->lx) {
        case 64: hipLaunchKernelGGL((el_step_s<64, 1, SAVE>), grid, block, 0, st, p); break;
        case 32: hipLaunchKernelGGL((el_step_s<32, 1, SAVE>), grid, block, 0, st, p); break;
        default: hipLaunchKernelGGL((el_step_s<16, 1, SAVE>), grid, block, 0, st, p); break;
    }
}

void el_cluster_setup(mifwi_elastic_plan *pl)
{
    pl->cluster = 0; pl->NW = 0; pl->PL = 4 * (pl->ng + 2); pl->cl_shots = 0;
    pl->cl_xh = 0; pl->adj_xh = 0;
    pl->adj_PL = pl->PL; pl->fwd_PL = pl->PL;
    // LDS row pitch: with the dense group -> thread deal (lane v <-> group v of the slab, row-major) a wave's 64 lanes
    // cross a row break almost always; the 16-byte slot of lane v stays congruent to v (mod 16) across the break - and
    // every ds_read_b128 lane group conflict-free - iff the pitch in slots is congruent to the groups per row
    // (measured on 100x300: SQ_LDS_BANK_CONFLICT of the forward loop -22 %, step 6.20 -> 6.07 us; the adjoint's LDS
    // has no room for the wider rows at 8 slabs).  MIFWI_EL_PL_SKEW: bit 0 forward (default on), bit 1 adjoint.
    const int skew = env_int("MIFWI_EL_PL_SKEW", 1);
    int Pskew = pl->ng + 2;
    while ((Pskew - pl->ng) % 16) ++Pskew;
    const int PL_plain = pl->PL;
    if (skew & 1) pl->PL = 4 * Pskew;
    if (skew & 2) pl->adj_PL = 4 * Pskew;
    pl->cl_lds = 0; pl->xbuf_elems = 0; pl->xcc_elems = 0;
    pl->cl_adj = 0; pl->adj_NW = 0; pl->adj_shots = 0; pl->adj_lds = 0; pl->adj_ng = 0; pl->adj_zrows = 0; pl->list_elems = 0;
    {   // el_build_tile_taps: start [nshot][ntiles + 1], cursor [nshot][ntiles], list [nshot][nrec * ntap]
        const long long ntiles = (long long)mifwi::ceil_div(pl->ng, AGO) * mifwi::ceil_div(pl->d.nz, ATZ);
        pl->tile_elems = std::max<long long>(1024, mifwi::round_up64(pl->d.nshot * (2 * ntiles + 1 + (long long)pl->d.nrec * pl->d.ntap), 64));
    }
    if (pl->d.ntap != 1 || pl->d.source_type != 0 || pl->d.record_pressure) return;   // per-step kernels only
    const bool want_fwd = env_int("MIFWI_EL_CLUSTER", 1) != 0;
    int ncu = 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, pl->device) != hipSuccess) return;
    const int forced = env_int("MIFWI_EL_NW", 0);
    // Slab count (both loops): cost of a step ~ fixed part (barriers, two hand-off flights) + groups per
    // thread, times the launches the shot batch needs (measured on 100x300: 6 shots 8.7 us at 8 slabs,
    // 7.2 us at 20).  Few shots -> many thin slabs; a full batch -> the fewest slabs that fit.
    double best = 1e30;
    for (int nw = 1; nw <= 32 && want_fwd; ++nw) {
        if (forced > 0 && nw != forced) continue;
        const int rows = mifwi::ceil_div(pl->d.nz, nw);
        if (pl->d.nz / nw < 4) break;
        long long lds = (5LL * (rows + 4) * pl->PL + 6LL * pl->gp + 8LL * rows + 8) * sizeof(float);
        int PLq = pl->PL;
        if (lds > 150 * 1024 && pl->PL != PL_plain) {          // the wider rows do not fit this slab height: plain pitch
            PLq = PL_plain;
            lds = (5LL * (rows + 4) * PLq + 6LL * pl->gp + 8LL * rows + 8) * sizeof(float);
        }
        if (lds > 150 * 1024) continue;
        if ((long long)rows * pl->ng > 2 * kEcThreads || kEcRowFields * pl->gp > kEcGr * kEcThreads) continue;
        const int per_launch = 8 * (ncu / (8 * nw));
        if (per_launch < 8) break;
        const double cost = mifwi::ceil_div(pl->d.nshot, per_launch) * (4.8 + (double)rows * pl->ng / kEcThreads);
        if (cost < best - 1e-9) {
            best = cost;
            pl->cluster = 1; pl->NW = nw; pl->cl_shots = per_launch; pl->cl_lds = (int)lds;
            pl->cl_ng = mifwi::ceil_div(rows * pl->ng, kEcThreads);
            pl->fwd_PL = PLq;
        }
    }
    // every slab's deal must suit the lane-shift form of the x-halo (ec_xhalo_deal_ok); MIFWI_EL_XHALO=0: plain LDS reads
    auto xhalo_ok = [&](int nw, int ngq) {
        if (env_int("MIFWI_EL_XHALO", 1) == 0) return 0;
        for (int w = 0; w < nw; ++w) {
            int r0, R;
            ec_slab_rows(pl->d.nz, nw, w, r0, R);
            if (!ec_xhalo_deal_ok(R, pl->ng, ngq * kEcThreads)) return 0;
        }
        return 1;
    };
    pl->cl_xh = pl->cluster ? xhalo_ok(pl->NW, pl->cl_ng) : 0;
    if (pl->cluster) {
        for (const void *fn : {(const void *)el_cluster_fwd<false, 1, false>, (const void *)el_cluster_fwd<true, 1, false>,
                               (const void *)el_cluster_fwd<false, 2, false>, (const void *)el_cluster_fwd<true, 2, false>,
                               (const void *)el_cluster_fwd<false, 1, true>, (const void *)el_cluster_fwd<true, 1, true>,
                               (const void *)el_cluster_fwd<false, 2, true>, (const void *)el_cluster_fwd<true, 2, true>,
                               (const void *)el_cluster_fwd<false, 1, false, true>, (const void *)el_cluster_fwd<true, 1, false, true>,
                               (const void *)el_cluster_fwd<false, 2, false, true>, (const void *)el_cluster_fwd<true, 2, false, true>})
            if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, mifwi::kClusterLdsLimit) != hipSuccess) {
                (void)hipGetLastError();       // not sticky: fall back to one launch per half step
                pl->cluster = 0;
            }
    }
    // adjoint: four E/D planes + the adjoint memory variables of the slab's C-PML cells in LDS; the
    // gradient accumulators are per shot, so it needs shot groups of one (not used when the caller
    // asked for a group size)
    const int forced_adj = env_int("MIFWI_EL_ADJ_NW", forced);
    if (env_int("MIFWI_EL_CLUSTER_ADJ", 1) != 0 && pl->d.shots_per_group <= 0) {
        double best_adj = 1e30;
        const int adj_skewed = pl->adj_PL;             // the candidate pitch (MIFWI_EL_PL_SKEW bit 1); pl->adj_PL = what the chosen slabs use
        for (int nw = 1; nw <= 32; ++nw) {
            if (forced_adj > 0 && nw != forced_adj) continue;
            if (pl->d.nz / nw < 4) break;
            const int rows = mifwi::ceil_div(pl->d.nz, nw);
            int zmax = 0;
            for (int w = 0; w < nw && pl->W > 0; ++w) {
                const int base = pl->d.nz / nw, rem = pl->d.nz - base * nw;
                const int R = base + (w < rem ? 1 : 0), r0 = w * base + std::min(w, rem);
                const int ntop = std::max(0, std::min(r0 + R, pl->W) - r0);
                const int nbot = std::max(0, r0 + R - std::max(r0, pl->d.nz - pl->W));
                zmax = std::max(zmax, ntop + nbot);
            }
            auto adj_lds = [&](int PLq) {
                return (long long)((4LL * (rows + 4) * PLq + 6LL * pl->gp + 8LL * rows + 4LL * rows * pl->wx + 4LL * zmax * pl->gp +
                                    (rows + 4) + 2LL * kEaRcvRows * PLq) * sizeof(float));   // + row map, receiver rows
            };
            int PLq = adj_skewed;
            long long lds = adj_lds(PLq);
            if (lds > kEaLdsLimit && PLq != PL_plain) {         // the wider rows do not fit this slab height: plain pitch
                PLq = PL_plain;
                lds = adj_lds(PLq);
            }
            if (lds > kEaLdsLimit) continue;
            if ((long long)rows * pl->ng > 2 * kEcThreads || kEcRowFields * pl->gp > kEcGr * kEcThreads) continue;
            const int per_launch = 8 * (ncu / (8 * nw));
            if (per_launch < 8) break;
            const double cost = mifwi::ceil_div(pl->d.nshot, per_launch) * (4.8 + (double)rows * pl->ng / kEcThreads);
            if (cost < best_adj - 1e-9) {
                best_adj = cost;
                pl->cl_adj = 1; pl->adj_NW = nw; pl->adj_shots = per_launch; pl->adj_lds = (int)lds; pl->adj_PL = PLq;
                pl->adj_ng = mifwi::ceil_div(rows * pl->ng, kEcThreads); pl->adj_zrows = zmax;
            }
        }
        pl->adj_xh = pl->cl_adj ? xhalo_ok(pl->adj_NW, pl->adj_ng) : 0;
        if (pl->cl_adj)
            for (const void *fn : {(const void *)el_cluster_adj<1, false>, (const void *)el_cluster_adj<2, false>,
                                   (const void *)el_cluster_adj<1, true>, (const void *)el_cluster_adj<2, true>,
                                   (const void *)el_cluster_adj<1, false, true>, (const void *)el_cluster_adj<2, false, true>})
                if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kEaLdsLimit) != hipSuccess) {
                    (void)hipGetLastError();
                    pl->cl_adj = 0;
                }
    }
    const int nwmax = std::max(pl->cluster ? pl->NW : 0, pl->cl_adj ? pl->adj_NW : 0);
    // granules, the XCC_ID table of mifwi::same_xcd ([nshot][nwmax] ints) and the block of the error word
    pl->xcc_elems = nwmax > 0 ? mifwi::round_up64((long long)pl->d.nshot * nwmax, 64) : 0;
    if (nwmax > 0)
        pl->xbuf_elems = mifwi::round_up64(2LL * pl->d.nshot * nwmax * 4 * kEcRowFields * pl->gp, 64) + pl->xcc_elems + 64;
    if (pl->cl_adj)
        pl->list_elems = mifwi::round_up64((long long)pl->d.nshot * pl->adj_NW * (1 + pl->d.nrec), 64);
}

template <bool SAVE, bool AG>
int el_cluster_run(const mifwi_elastic_plan *pl, EcParams c, float *xbuf, hipStream_t st)
{
    if (mifwi::fake_timeout() == 1) return mifwi::kClusterTimedOut;
    MIFWI_HIP_TRY(hipMemsetAsync(xbuf, 0, sizeof(float) * pl->xbuf_elems, st));
    c.xbuf = reinterpret_cast<unsigned long long *>(xbuf);
    c.err = reinterpret_cast<int *>(xbuf + pl->xbuf_elems - 64);
    c.xcc_tab = reinterpret_cast<int *>(xbuf + pl->xbuf_elems - 64 - pl->xcc_elems);
#ifdef MIFWI_ABLATIONS
    // MIFWI_EL_CL_TRACE=<file>: phase time stamps (EC_STAMP) of one workgroup, steps 64..127, appended as text
    const char *trace_path = getenv("MIFWI_EL_CL_TRACE");
    const size_t trace_n = 64 * 8 * 16;
    if (trace_path && *trace_path) {
        MIFWI_HIP_TRY(hipMalloc(&c.trace, trace_n * sizeof(long long)));
        MIFWI_HIP_TRY(hipMemsetAsync(c.trace, 0, trace_n * sizeof(long long), st));
    }
#endif
    for (int s0 = 0; s0 < pl->d.nshot; s0 += pl->cl_shots) {
        c.shot0 = s0;
        c.shot1 = std::min(pl->d.nshot, s0 + pl->cl_shots);
        const int nsl8 = mifwi::ceil_div(c.shot1 - s0, 8);
        const dim3 grid(8 * pl->NW * nsl8), block(kEcThreads);
        if constexpr (!AG) {                 // (the agent-scope tier keeps the plain reads: one variant less to build)
            if (pl->cl_xh) {
                if (pl->cl_ng == 1) hipLaunchKernelGGL((el_cluster_fwd<SAVE, 1, false, true>), grid, block, pl->cl_lds, st, c);
                else hipLaunchKernelGGL((el_cluster_fwd<SAVE, 2, false, true>), grid, block, pl->cl_lds, st, c);
                continue;
            }
        }
        if (pl->cl_ng == 1) hipLaunchKernelGGL((el_cluster_fwd<SAVE, 1, AG>), grid, block, pl->cl_lds, st, c);
        else hipLaunchKernelGGL((el_cluster_fwd<SAVE, 2, AG>), grid, block, pl->cl_lds, st, c);
    }
    MIFWI_HIP_TRY(hipGetLastError());
    int err[4] = {0, 0, 0, 0};
    MIFWI_HIP_TRY(hipMemcpyAsync(err, c.err, sizeof(err), hipMemcpyDeviceToHost, st));
    MIFWI_HIP_TRY(hipStreamSynchronize(st));
#ifdef MIFWI_ABLATIONS
    if (c.trace) {
        std::vector<long long> h(trace_n);
        MIFWI_HIP_TRY(hipMemcpy(h.data(), c.trace, trace_n * sizeof(long long), hipMemcpyDeviceToHost));
        MIFWI_HIP_TRY(hipFree(c.trace));
        if (FILE *fp = fopen(trace_path, "a")) {
            fprintf(fp, "# el_cluster_fwd save=%d steps=%d\n", SAVE ? 1 : 0, c.n_last - c.n_first);
            for (size_t i = 0; i < trace_n; i += 16) {
                for (int k = 0; k < 16; ++k) fprintf(fp, "%lld ", h[i + k]);
                fprintf(fp, "\n");
            }
            fclose(fp);
        }
    }
#endif
    const int verdict = mifwi::cluster_verdict(err, "elastic forward");
    return mifwi::fake_timeout() == 2 ? mifwi::kClusterTimedOut : verdict;
}

// A timed-out single-launch attempt has advanced the state by an unknown number of steps: a call that started from
// the zero state is zeroed again, a resumed call (time checkpointing) gets back the copy of its input state taken
// before the attempt; then the per-step kernels run the range.
int el_cluster_backup(float *work, long long state_elems, float *backup, int32_t flags, hipStream_t st)
{
    if (flags & MIFWI_ZERO_STATE) return MIFWI_OK;
    MIFWI_HIP_TRY(hipMemcpyAsync(backup, work, sizeof(float) * state_elems, hipMemcpyDeviceToDevice, st));
    return MIFWI_OK;
}
int el_cluster_restore(float *work, long long state_elems, const float *backup, int32_t flags, hipStream_t st)
{
    if (flags & MIFWI_ZERO_STATE) MIFWI_HIP_TRY(hipMemsetAsync(work, 0, sizeof(float) * state_elems, st));
    else MIFWI_HIP_TRY(hipMemcpyAsync(work, backup, sizeof(float) * state_elems, hipMemcpyDeviceToDevice, st));
    return MIFWI_OK;
}

}  // namespace

extern "C" {

int mifwi_elastic_plan_create(mifwi_elastic_plan **plan, int device, const mifwi_elastic_desc *d)
{
    if (!plan || !d) return mifwi::fail(MIFWI_EINVAL, "null plan/desc");
    if (d->nz < 1 || d->nx < 1 || d->nt < 1 || d->nshot < 1 || d->nsrc < 0 || d->nrec < 0)
        return mifwi::fail(MIFWI_EINVAL, "bad sizes nz=%d nx=%d nt=%d nshot=%d", d->nz, d->nx, d->nt,
                           d->nshot);
    if (d->ntap != 1 && d->ntap != 4) return mifwi::fail(MIFWI_EINVAL, "ntap must be 1 or 4");
    if (d->free_surface != 0 && d->free_surface != 1)
        return mifwi::fail(MIFWI_EINVAL, "free_surface must be 0 or 1");
    if (d->free_surface && d->nz < 4) return mifwi::fail(MIFWI_EINVAL, "free surface needs nz >= 4");
    if (d->pml_width < 0) return mifwi::fail(MIFWI_EINVAL, "pml_width < 0");
    if (d->record_pressure != 0 && d->record_pressure != 1)
        return mifwi::fail(MIFWI_EINVAL, "record_pressure must be 0 or 1");
    if (d->source_type < 0 || d->source_type > 2)
        return mifwi::fail(MIFWI_EINVAL, "source_type must be 0 (explosive), 1 (force x) or 2 (force z)");
    if (d->fd_order != 0 && d->fd_order != 2 && d->fd_order != 4)
        return mifwi::fail(MIFWI_EINVAL, "fd_order %d: the staggered-grid stencils are built for order 2 and 4", d->fd_order);
    if (d->snapshot_format != MIFWI_SNAPSHOT_F32 && d->snapshot_format != MIFWI_SNAPSHOT_BF16)
        return mifwi::fail(MIFWI_EINVAL, "snapshot_format must be MIFWI_SNAPSHOT_F32 or MIFWI_SNAPSHOT_BF16");
    int rc = mifwi::check_device(device);
    if (rc) return rc;
    // function attributes and CU counts queried during set-up belong to THIS device (one process per
    // GPU sees all eight devices)
    MIFWI_HIP_TRY(hipSetDevice(device));
    mifwi_elastic_plan *pl = new mifwi_elastic_plan;
    pl->d = *d;
    if (pl->d.fd_order == 0) pl->d.fd_order = 4;
    pl->device = device;
    pl->rec_p = nullptr; pl->g_p = nullptr;
    pl->ng = mifwi::ceil_div(d->nx, 4);
    pl->gp = 4 * pl->ng;
    pl->pitch = (int)mifwi::round_up64(4 * (pl->ng + 3), 32);
    pl->field_stride = (long long)(d->nz + 4) * pl->pitch;
    pl->shot_stride = 5 * pl->field_stride;
    pl->fields_elems = pl->shot_stride * d->nshot;
    pl->coef_elems = (long long)d->nz * pl->gp;
    // strip geometry: W = pml_width + 1 nodes per side (half-node profiles reach one node further)
    pl->W = d->pml_width > 0 ? d->pml_width + 1 : 0;
    if (pl->W > 0) {
        pl->wl = (int)mifwi::round_up64(pl->W, 4);
        pl->xr0 = ((d->nx - pl->W) / 4) * 4;
        if (pl->xr0 < pl->wl || 2 * pl->W > d->nz) {
            delete pl;
            return mifwi::fail(MIFWI_EINVAL, "grid %dx%d too small for a %d-node C-PML", d->nz,
                               d->nx, d->pml_width);
        }
        pl->wx = pl->wl + (pl->gp - pl->xr0);
    } else {
        pl->wl = pl->xr0 = pl->wx = 0;
    }
    pl->psix_elems = 4LL * d->nz * pl->wx * d->nshot;
    pl->psiz_elems = 4LL * 2 * pl->W * pl->gp * d->nshot;
    pl->lx = 16;
    for (int cand : {64, 32}) {
        const int padded = mifwi::ceil_div(pl->ng, cand) * cand;
        if (padded * 10 <= pl->ng * 12) { pl->lx = cand; break; }
    }
    { const int v = env_int("MIFWI_EL_LX", 0); if (v == 16 || v == 32 || v == 64) pl->lx = v; }
    pl->rz = 1;
    // 3: tile-major, shots of the launch back to back through every tile (xcd_tile); 1: every XCD walks one contiguous run of
    // the tiles of each shot; 2: whole shots per XCD.  Measured (r04): 350x1700 x 6 shots forward 43.5 -> 40.5 us per step
    // with 3 against 1, adjoint pair and the 1000x3000 passes (2-3 shots per launch) unchanged
    pl->xcd = env_int("MIFWI_EL_XCD", 3);
    int gs = d->shots_per_group;
    if (gs <= 0) gs = env_int("MIFWI_EL_GS", 4);
    if (gs <= 0) gs = 1;
    if (gs > d->nshot) gs = d->nshot;
    // groups of equal size where a slightly smaller one gives them (a 6-shot chunk of a shot-chunked gradient pass: 3 + 3
    // instead of 4 + 2 - the workgroups of the short group would idle while the others finish their fourth shot)
    if (d->shots_per_group <= 0 && env_int("MIFWI_EL_GS", 0) <= 0 && gs == 4 && d->nshot % 4 != 0 && d->nshot % 3 == 0) gs = 3;
    pl->gs = gs;
    pl->ngroups = mifwi::ceil_div(d->nshot, gs);
    pl->psi_elems = mifwi::round_up64(pl->psix_elems + pl->psiz_elems, 64);
    el_cluster_setup(pl);
    // bf16 snapshot planes: the per-step kernels only (the single-launch time loops are not bound by the
    // snapshot stream, and their grids' snapshots fit in memory many times over)
    pl->snap_bf16 = d->snapshot_format == MIFWI_SNAPSHOT_BF16 && !pl->cluster && !pl->cl_adj;
    // column-blocked snapshot planes (snap_cell): plans that never run a single-launch kernel (those write and read
    // row-major planes, and a call may change family between checkpoint segments); MIFWI_EL_SNAP_BLOCKED=0: row-major
    pl->sblk = (!pl->cluster && !pl->cl_adj && env_int("MIFWI_EL_SNAP_BLOCKED", 1) != 0) ? 1 : 0;
    pl->splane = pl->sblk ? 64LL * d->nz * mifwi::ceil_div(pl->ng, 16) : pl->coef_elems;
    pl->snap_shot = pl->snap_bf16 ? mifwi::round_up64(5 * pl->splane / 2, 4) : 5 * pl->splane;
    pl->fused_adj = (!pl->cl_adj && d->source_type == 0 && !d->record_pressure) ? env_int("MIFWI_EL_FUSED_ADJ", 0) : 0;
    if (pl->fused_adj < 0 || pl->fused_adj > 2) pl->fused_adj = 0;
    pl->walk_rows = 0; pl->walk_chunks = 0;
    {
        // Infinity Cache residency (per-step family, large grids): a pass over the time range takes only as
        // many shots as keep state + materials (+ gradient accumulators) under kResident bytes; measured on
        // 1000x3000: forward 3 shots (240 MB) 0.61 ms, 4 shots (300 MB) 0.82 ms = the all-shots time
        constexpr double kResident = 250e6;
        const double cells = (double)pl->coef_elems;
        const double mats = 4.0 * 5.0 * cells;
        const double psi1 = 4.0 * (double)(pl->psix_elems + pl->psiz_elems) / d->nshot;
        const double fstate = 4.0 * (double)pl->shot_stride + psi1;
        int ps = d->nshot;
        // forward a little below the adjoint's bound: 1000x3000 runs 0.61 ms with 3 shots (240 MB) and 0.62 ms with 2
        // (180 MB), but a set that close to the cache size fell back to the all-shots time on some boxes (acoustic plan)
        if (d->nshot * fstate + mats > kResident) ps = (int)std::max(1.0, std::floor((230e6 - mats) / fstate));
        ps = env_int("MIFWI_EL_PASS_SHOTS", ps);
        pl->pass_shots = std::min(d->nshot, std::max(1, ps));
        // adjoint: groups of gs shots share one accumulator set; when the groups do not all fit, smaller groups
        // (more accumulator traffic, 40/gs B per cell-step) can still pay: 1000x3000, gs 2, one group per pass
        // 0.87 ms against 0.98 ms for gs 4 over all shots
        const double astate = (pl->fused_adj ? 8.0 : 4.0) * (double)pl->shot_stride + 2.0 * psi1, accg = 4.0 * 5.0 * cells;
        if (!pl->cl_adj && d->shots_per_group <= 0 && env_int("MIFWI_EL_GS", 0) <= 0 &&
            pl->ngroups * (pl->gs * astate + accg) + mats > kResident) {
            int g = pl->gs;
            while (g > 2 && g * astate + accg + mats > kResident) g /= 2;
            if (g * astate + accg + mats <= kResident && g != pl->gs) {
                pl->gs = g;
                pl->ngroups = mifwi::ceil_div(d->nshot, g);
            }
        }
        const double group = pl->gs * astate + accg;
        int k = pl->ngroups;
        if (pl->ngroups * group + mats > kResident) {
            k = (int)std::floor((kResident - mats) / group);
            if (k < 1) k = pl->ngroups;                 // nothing fits: one pass over all groups
        }
        k = env_int("MIFWI_EL_PASS_GROUPS", k);
        pl->pass_groups = std::min(pl->ngroups, std::max(1, k));
    }
    // V+S in one launch pays where the time loop has to be cut into Infinity-Cache-sized passes anyway (measured,
    // 1000x3000 x 16 shots: 622 -> 558 us per step, 516 with bf16 planes; 350x1700 x 32: 232 -> 232, 214 -> 209) and
    // on any launch of a million cells or more (350x1700 x 4 shots, the chunks of a shot-chunked gradient pass:
    // 286 -> 260 us per step of all 32 shots)
    pl->fused = !pl->cluster && d->source_type == 0 && !d->record_pressure &&
                env_int("MIFWI_EL_FUSED", (pl->pass_shots < d->nshot || (double)d->nz * d->nx * d->nshot >= 1e6) ? 1 : 0) != 0;
    {
        const double cells = (double)pl->coef_elems, mats = 4.0 * 5.0 * cells;
        const double fstate = 4.0 * (double)pl->shot_stride + 4.0 * (double)(pl->psix_elems + pl->psiz_elems) / d->nshot;
        int ps = d->nshot;
        if (d->nshot * 2.0 * fstate + mats > 250e6) ps = (int)std::max(1.0, std::floor((230e6 - mats) / (2.0 * fstate)));
        ps = env_int("MIFWI_EL_FUSED_PASS_SHOTS", ps);
        pl->fused_pass_shots = std::min(d->nshot, std::max(1, ps));
    }
    if (pl->cl_adj) {                    // the adjoint cluster kernel keeps one accumulator set per shot
        pl->gs = 1;
        pl->ngroups = d->nshot;
    }
    if (pl->fused_adj == 2) {
        // el_adj_walk: columns of 16 groups are cut into chunks of rows; every chunk pays one start-up iteration (the
        // rows above it that its first rows depend on), so as few chunks as still fill the chip a few times over
        const size_t lds = walk_lds_bytes(pl->gs);
        bool ok = lds <= 160 * 1024;
        for (const void *fn : {(const void *)el_adj_walk<false>, (const void *)el_adj_walk<true>})
            if (ok && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
                (void)hipGetLastError();
                ok = false;
            }
        if (!ok) {
            pl->fused_adj = 0;            // a group too large for the carry area: the two-launch form
        } else {
            int ncu = 256;
            (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, pl->device);
            const int per_cu = std::max(1, std::min(MIFWI_WALK_WAVES, (int)(160 * 1024 / lds)));
            const int cols = mifwi::ceil_div(pl->ng, WOG), tiles = mifwi::ceil_div(d->nz, WTZ);
            const int launch_groups = std::min(pl->pass_groups, pl->ngroups);
            const int want = mifwi::ceil_div(2 * per_cu * ncu, std::max(1, cols * launch_groups));
            int chunks = std::min(tiles, std::max(1, want));
            chunks = std::min(chunks, std::max(1, tiles / 2));          // at least two iterations per start-up iteration
            int rows = mifwi::ceil_div(mifwi::ceil_div(d->nz, chunks), WTZ) * WTZ;
            rows = env_int("MIFWI_EL_WALK_ROWS", rows);
            rows = std::max(WTZ, rows / WTZ * WTZ);
            pl->walk_rows = rows;
            pl->walk_chunks = mifwi::ceil_div(d->nz, rows);
        }
    }
    *plan = pl;
    return MIFWI_OK;
}

int mifwi_elastic_plan_cluster_slabs(const mifwi_elastic_plan *plan, int32_t adjoint)
{
    if (!plan) return 0;
    return adjoint ? (plan->cl_adj ? plan->adj_NW : 0) : (plan->cluster ? plan->NW : 0);
}

int mifwi_elastic_plan_pass_sizes(const mifwi_elastic_plan *plan, int32_t *forward_shots, int32_t *adjoint_groups)
{
    if (!plan) return mifwi::fail(MIFWI_EINVAL, "null plan");
    if (forward_shots) *forward_shots = plan->pass_shots;
    if (adjoint_groups) *adjoint_groups = plan->pass_groups;
    return MIFWI_OK;
}

int mifwi_elastic_plan_bind_pressure(mifwi_elastic_plan *plan, float *rec_p, const float *g_p)
{
    if (!plan) return mifwi::fail(MIFWI_EINVAL, "null plan");
    if ((rec_p || g_p) && !plan->d.record_pressure)
        return mifwi::fail(MIFWI_EINVAL, "the plan was created with record_pressure = 0");
    plan->rec_p = rec_p;
    plan->g_p = g_p;
    return MIFWI_OK;
}

int mifwi_elastic_plan_destroy(mifwi_elastic_plan *plan)
{
    delete plan;
    return MIFWI_OK;
}

int mifwi_elastic_plan_layout(const mifwi_elastic_plan *pl, mifwi_elastic_layout *out)
{
    if (!pl || !out) return mifwi::fail(MIFWI_EINVAL, "null plan/layout");
    out->gp = pl->gp; out->pitch = pl->pitch; out->ngroups = pl->ngroups;
    out->shots_per_group = pl->gs;
    out->coef_elems = pl->coef_elems;
    out->snap_step_elems = pl->snap_shot * pl->d.nshot;
    out->snapshot_format = pl->snap_bf16 ? MIFWI_SNAPSHOT_BF16 : MIFWI_SNAPSHOT_F32;
    out->kernel_flags = (pl->cluster ? MIFWI_EL_KERNEL_FWD_SINGLE_LAUNCH : 0) | (pl->cl_adj ? MIFWI_EL_KERNEL_ADJ_SINGLE_LAUNCH : 0) |
                        (pl->fused ? MIFWI_EL_KERNEL_FWD_FUSED_STEP : 0) | (pl->fused_adj ? MIFWI_EL_KERNEL_ADJ_FUSED_STEP : 0) |
                        (pl->cluster && pl->cl_xh ? MIFWI_EL_KERNEL_FWD_LANE_HALO : 0) |
                        (pl->cl_adj && pl->adj_xh ? MIFWI_EL_KERNEL_ADJ_LANE_HALO : 0);
    const long long psi = pl->psi_elems;
    const long long bbox = mifwi::round_up64(4LL * pl->d.nshot, 64);
    out->state_elems = pl->fields_elems + psi;
    // single-launch plans: room for a copy of a resumed call's input state (el_cluster_backup)
    out->work_forward_elems = out->state_elems + bbox + (pl->cluster ? pl->xbuf_elems + out->state_elems : 0) +
                              (pl->fused ? out->state_elems : 0);
    const long long adj_state = pl->fields_elems + 2 * psi + 5LL * pl->ngroups * pl->splane;
    out->work_backward_elems = adj_state + bbox + (pl->cl_adj ? pl->xbuf_elems + pl->list_elems + adj_state : 0) +
                               (pl->fused_adj ? pl->fields_elems : 0) + pl->tile_elems;
    return MIFWI_OK;
}

int mifwi_elastic_forward(mifwi_elastic_plan *pl, const float *mat, const float *pz,
                          const float *px, const float *f, const int32_t *src_cell,
                          const float *src_w, const int32_t *rec_cell, const float *rec_w,
                          float *rec_vx, float *rec_vz, float *snap, float *work, int32_t n_begin,
                          int32_t n_end, int32_t flags, void *stream)
{
    if (!pl || !mat || !pz || !px || !work) return mifwi::fail(MIFWI_EINVAL, "null argument");
    const mifwi_elastic_desc &d = pl->d;
    if (n_begin < 0 || n_end > d.nt || n_begin > n_end)
        return mifwi::fail(MIFWI_EINVAL, "bad step range [%d,%d) for nt=%d", n_begin, n_end, d.nt);
    if (d.nsrc > 0 && (!f || !src_cell || !src_w))
        return mifwi::fail(MIFWI_EINVAL, "sources declared but f/src_cell/src_w is null");
    if ((rec_vx || rec_vz) && (!rec_cell || !rec_w || !rec_vx || !rec_vz))
        return mifwi::fail(MIFWI_EINVAL, "receiver output needs rec_cell, rec_w, rec_vx and rec_vz");
    int rc = mifwi::check_device(pl->device);
    if (rc) return rc;
    MIFWI_HIP_TRY(hipSetDevice(pl->device));
    hipStream_t st = (hipStream_t)stream;
    // state layout: [fields | psi | bbox], updated in place
    const long long psi = pl->psi_elems;
    float *fields = work, *psi_state = work + pl->fields_elems;
    int *bbox = reinterpret_cast<int *>(work + pl->fields_elems + psi);
    if (flags & MIFWI_ZERO_STATE)
        MIFWI_HIP_TRY(hipMemsetAsync(work, 0, sizeof(float) * (pl->fields_elems + psi), st));
    if (d.nsrc > 0)
        hipLaunchKernelGGL(el_points_bbox, dim3(d.nshot), dim3(kThreads), 0, st, src_cell,
                           d.nsrc * d.ntap, d.nx, bbox);
    ElParams p = el_base(pl, mat, pz, px);
    const long long snap_step = pl->snap_shot * d.nshot;
    const int save = !snap ? 0 : pl->snap_bf16 ? 2 : 1;
    const bool want_rec = rec_vx != nullptr && d.nrec > 0;
    if (pl->cluster && n_end > n_begin) {
        float *xbuf = work + pl->fields_elems + psi + mifwi::round_up64(4LL * d.nshot, 64);
        EcParams c;
        memset(&c, 0, sizeof(c));
        c.nz = d.nz; c.nx = d.nx; c.ng = pl->ng; c.gp = pl->gp; c.pitch = pl->pitch;
        c.field_stride = (unsigned)pl->field_stride; c.shot_stride = pl->shot_stride;
        c.nshot = d.nshot; c.NW = pl->NW; c.PL = pl->fwd_PL;
        c.n_first = n_begin; c.n_last = n_end;
        c.W = pl->W; c.wl = pl->wl; c.xr0 = pl->xr0; c.wx = pl->wx; c.fsurf = d.free_surface;
        c.psix_shot = 4LL * d.nz * pl->wx; c.psiz_shot = 4LL * 2 * pl->W * pl->gp;
        c.mat = mat; c.pz = pz; c.px = px;
        c.fields = fields; c.psix = psi_state; c.psiz = psi_state + pl->psix_elems;
        c.S = snap; c.s_first = n_begin; c.s_step = snap_step;
        c.nsrc = d.nsrc; c.nrec = d.nrec; c.src_cell = src_cell; c.src_w = src_w; c.f = f;
        c.rec_cell = rec_cell; c.rec_w = rec_w;
        c.rec_vx = want_rec ? rec_vx : nullptr; c.rec_vz = want_rec ? rec_vz : nullptr;
        c.dbg = env_int("MIFWI_EL_CL_DBG", 0);
        c.nap = env_int("MIFWI_POLL_NAP", mifwi::ceil_div(d.nz, c.NW) >= 8 ? 48 : 1);   // mifwi::poll_nap
        c.K = fd_weights(d.fd_order);
        float *backup = xbuf + pl->xbuf_elems;
        rc = el_cluster_backup(work, pl->fields_elems + psi, backup, flags, st);
        if (rc) return rc;
        rc = snap ? el_cluster_run<true, false>(pl, c, xbuf, st) : el_cluster_run<false, false>(pl, c, xbuf, st);
        if (rc == mifwi::kClusterMisplaced) {          // not on one XCD: once more with hand-offs through the fabric
            mifwi::note_agent_tier("elastic forward");
            rc = el_cluster_restore(work, pl->fields_elems + psi, backup, flags, st);
            if (rc) return rc;
            rc = snap ? el_cluster_run<true, true>(pl, c, xbuf, st) : el_cluster_run<false, true>(pl, c, xbuf, st);
        }
        if (rc != mifwi::kClusterTimedOut && rc != mifwi::kClusterMisplaced) return rc;
        mifwi::note_fallback("elastic");
        rc = el_cluster_restore(work, pl->fields_elems + psi, backup, flags, st);
        if (rc) return rc;
    }
    if (pl->fused && n_end > n_begin) {
        // V+S in one launch: the state ping-pongs between the copy at the head of the work buffer and a second
        // one behind the bounding boxes.  Shots are taken a few at a time (both copies of their state stay in the
        // Infinity Cache); a pass with an odd number of steps ends with a copy back of its shots.
        const long long state = pl->fields_elems + psi;
        float *B = work + state + mifwi::round_up64(4LL * d.nshot, 64);
        MIFWI_HIP_TRY(hipMemsetAsync(B, 0, sizeof(float) * state, st));     // its pad rows / columns stay zero
        ElParams q = p;
        q.ninj = d.nsrc; q.ntap_inj = d.ntap; q.inj_cell = src_cell; q.inj_w = src_w; q.inj_bbox = bbox;
        q.nsmp = want_rec ? d.nrec : 0; q.ntap_smp = d.ntap; q.smp_cell = rec_cell; q.smp_w = rec_w;
        const int tx = mifwi::ceil_div(pl->ng, FTG), tz = mifwi::ceil_div(d.nz, FTZ);
        const int ex = want_rec ? mifwi::ceil_div(mifwi::ceil_div(d.nrec, kThreads), tx) : 0;
        for (int s0 = 0; s0 < d.nshot; s0 += pl->fused_pass_shots) {
            const int cs = std::min(pl->fused_pass_shots, d.nshot - s0);
            q.s0 = s0;
            q.tiles_z = tz;
            for (int n = n_begin; n < n_end; ++n) {
                const bool even = ((n - n_begin) & 1) == 0;
                float *in = even ? work : B, *out = even ? B : work;
                q.fields = in; q.psix = in + pl->fields_elems; q.psiz = q.psix + pl->psix_elems;
                q.fields_out = out; q.psix_out = out + pl->fields_elems; q.psiz_out = q.psix_out + pl->psix_elems;
                q.S = snap ? snap + (long long)(n - n_begin) * snap_step : nullptr;
                q.inj_amp0 = f ? f + (long long)n * d.nshot * d.nsrc : nullptr;
                // the receivers see the input state: the velocities of step n-1
                const bool smp = want_rec && n > n_begin;
                q.smp_out0 = smp ? rec_vx + (long long)(n - 1) * d.nshot * d.nrec : nullptr;
                q.smp_out1 = smp ? rec_vz + (long long)(n - 1) * d.nshot * d.nrec : nullptr;
                const dim3 grid(tx, tz + (smp ? ex : 0), cs);
                if (save == 2) hipLaunchKernelGGL(el_fwd_fused<2>, grid, dim3(kThreads), 0, st, q);
                else if (save == 1) hipLaunchKernelGGL(el_fwd_fused<1>, grid, dim3(kThreads), 0, st, q);
                else hipLaunchKernelGGL(el_fwd_fused<0>, grid, dim3(kThreads), 0, st, q);
            }
            float *last = ((n_end - n_begin) & 1) ? B : work;
            if (want_rec) {
                q.fields = last; q.tiles_z = 0;
                q.smp_out0 = rec_vx + (long long)(n_end - 1) * d.nshot * d.nrec;
                q.smp_out1 = rec_vz + (long long)(n_end - 1) * d.nshot * d.nrec;
                hipLaunchKernelGGL(el_sample_v, dim3(mifwi::ceil_div(d.nrec, kThreads), 1, cs), dim3(kThreads), 0,
                                   st, q);
            }
            if (last != work) {
                const long long psix1 = 4LL * d.nz * pl->wx, psiz1 = 4LL * 2 * pl->W * pl->gp;
                MIFWI_HIP_TRY(hipMemcpyAsync(work + s0 * pl->shot_stride, B + s0 * pl->shot_stride,
                                             sizeof(float) * cs * pl->shot_stride, hipMemcpyDeviceToDevice, st));
                if (psix1 > 0)
                    MIFWI_HIP_TRY(hipMemcpyAsync(work + pl->fields_elems + s0 * psix1, B + pl->fields_elems + s0 * psix1,
                                                 sizeof(float) * cs * psix1, hipMemcpyDeviceToDevice, st));
                if (psiz1 > 0)
                    MIFWI_HIP_TRY(hipMemcpyAsync(work + pl->fields_elems + pl->psix_elems + s0 * psiz1,
                                                 B + pl->fields_elems + pl->psix_elems + s0 * psiz1,
                                                 sizeof(float) * cs * psiz1, hipMemcpyDeviceToDevice, st));
            }
        }
    } else {
        p.fields = fields; p.psix = psi_state; p.psiz = psi_state + pl->psix_elems;
        ElParams ps = p;
        const bool force = d.source_type != 0;       // point force: injected into vx / vz by a launch of its own
        ps.ninj = force ? 0 : d.nsrc; ps.ntap_inj = d.ntap; ps.inj_cell = src_cell; ps.inj_w = src_w;
        ps.inj_bbox = bbox;
        ElParams pf = p;
        pf.ninj = d.nsrc; pf.ntap_inj = d.ntap; pf.inj_cell = src_cell; pf.inj_w = src_w;
        ps.nsmp = want_rec ? d.nrec : 0; ps.ntap_smp = d.ntap; ps.smp_cell = rec_cell; ps.smp_w = rec_w;
        // shots are independent: taken a few at a time, their state and the materials stay inside the
        // 256 MiB Infinity Cache from one launch to the next (pl->pass_shots; all shots when they fit anyway)
        for (int s0 = 0; s0 < d.nshot; s0 += pl->pass_shots) {
            const int cs = std::min(pl->pass_shots, d.nshot - s0);
            p.s0 = ps.s0 = s0;
            for (int n = n_begin; n < n_end; ++n) {
                float *Sn = snap ? snap + (long long)(n - n_begin) * snap_step : nullptr;
                p.S = Sn; ps.S = Sn;
                ps.inj_amp0 = f ? f + (long long)n * d.nshot * d.nsrc : nullptr;
                ps.smp_out0 = want_rec ? rec_vx + (long long)n * d.nshot * d.nrec : nullptr;
                ps.smp_out1 = want_rec ? rec_vz + (long long)n * d.nshot * d.nrec : nullptr;
                if (save == 2) launch_v<2>(pl, p, cs, st);
                else if (save == 1) launch_v<1>(pl, p, cs, st);
                else launch_v<0>(pl, p, cs, st);
                if (force && d.nsrc > 0 && f) {
                    pf.s0 = s0; pf.gs = cs; pf.inj_amp0 = ps.inj_amp0;
                    hipLaunchKernelGGL(el_inject_force, dim3(mifwi::ceil_div(cs, 64)), dim3(64), 0, st, pf,
                                       d.source_type);
                }
                if (save == 2) launch_s<2>(pl, ps, cs, st);
                else if (save == 1) launch_s<1>(pl, ps, cs, st);
                else launch_s<0>(pl, ps, cs, st);
                if (pl->rec_p && want_rec) {
                    ElParams pq = ps;
                    pq.s0 = s0; pq.gs = cs; pq.smp_out0 = pl->rec_p + (long long)n * d.nshot * d.nrec;
                    hipLaunchKernelGGL(el_sample_pressure, dim3(mifwi::ceil_div(cs * d.nrec, 64)), dim3(64), 0, st, pq);
                }
            }
        }
    }
    MIFWI_HIP_TRY(hipGetLastError());
    return MIFWI_OK;
}

int mifwi_elastic_backward(mifwi_elastic_plan *pl, const float *mat, const float *pz,
                           const float *px, const int32_t *src_cell, const float *src_w,
                           const int32_t *rec_cell, const float *rec_w, const float *g_vx,
                           const float *g_vz, const float *snap, int32_t snap_first,
                           float *grad_mat, float *grad_f, float *work, int32_t n_hi, int32_t n_lo,
                           int32_t flags, void *stream)
{
    if (!pl || !mat || !pz || !px || !work || !snap || !g_vx || !g_vz || !rec_cell || !rec_w)
        return mifwi::fail(MIFWI_EINVAL, "null argument");
    const mifwi_elastic_desc &d = pl->d;
    if (n_lo < 0 || n_hi > d.nt - 1 || n_lo > n_hi + 1)
        return mifwi::fail(MIFWI_EINVAL, "bad adjoint range n=%d..%d for nt=%d", n_hi, n_lo, d.nt);
    if (snap_first > n_lo)
        return mifwi::fail(MIFWI_EINVAL, "snapshots start at step %d but step %d is needed",
                           snap_first, n_lo);
    if (grad_f && (!src_cell || !src_w))
        return mifwi::fail(MIFWI_EINVAL, "grad_f requested but src_cell/src_w is null");
    if ((flags & MIFWI_FINALIZE) && !grad_mat)
        return mifwi::fail(MIFWI_EINVAL, "finalize requested but grad_mat is null");
    int rc = mifwi::check_device(pl->device);
    if (rc) return rc;
    MIFWI_HIP_TRY(hipSetDevice(pl->device));
    hipStream_t st = (hipStream_t)stream;
    const long long psi = pl->psi_elems;
    float *fields = work;
    float *psiA = work + pl->fields_elems, *psiB = psiA + psi;
    float *acc = psiB + psi;
    const long long nacc = 5LL * pl->ngroups * pl->splane;          // accumulator planes: the snapshot planes' layout and size
    int *bbox = reinterpret_cast<int *>(acc + nacc);
    float *fieldsB = reinterpret_cast<float *>(bbox) + mifwi::round_up64(4LL * d.nshot, 64);   // fused adjoint only
    if (flags & MIFWI_ZERO_STATE) {
        MIFWI_HIP_TRY(hipMemsetAsync(work, 0, sizeof(float) * (pl->fields_elems + 2 * psi + nacc), st));
        if (pl->fused_adj) MIFWI_HIP_TRY(hipMemsetAsync(fieldsB, 0, sizeof(float) * pl->fields_elems, st));
    }
    ElParams p = el_base(pl, mat, pz, px);
    p.fields = fields;
    p.acc = acc;
    ElParams ps = p;
    ps.gs = pl->gs;
    ps.ninj = d.nrec; ps.ntap_inj = d.ntap; ps.inj_cell = rec_cell; ps.inj_w = rec_w;    // el_inject_adjsrc
    const bool want_f = grad_f != nullptr && d.nsrc > 0;
    ps.nsmp = want_f ? d.nsrc : 0; ps.ntap_smp = d.ntap; ps.smp_cell = src_cell; ps.smp_w = src_w;
    const long long snap_step = pl->snap_shot * d.nshot;
    // per-tile receiver lists of el_adj_s, at the tail of the work buffer (mifwi_elastic_plan_layout)
    int *tile_start = nullptr, *tile_list = nullptr;
    const int tiles_x = mifwi::ceil_div(pl->ng, AGO), ntiles = tiles_x * mifwi::ceil_div(d.nz, ATZ);
    auto build_tile_lists = [&]() {
        mifwi_elastic_layout lay;
        mifwi_elastic_plan_layout(pl, &lay);
        int *base = reinterpret_cast<int *>(work + lay.work_backward_elems - pl->tile_elems);
        tile_start = base;
        int *cursor = base + (long long)d.nshot * (ntiles + 1);
        tile_list = cursor + (long long)d.nshot * ntiles;
        hipLaunchKernelGGL(el_build_tile_taps, dim3(d.nshot), dim3(256), 0, st, rec_cell, d.nrec * d.ntap, d.nx, tiles_x,
                           ntiles, tile_start, cursor, tile_list);
    };
    bool per_step = true;
    if (pl->cl_adj && n_hi >= n_lo) {
        float *xbuf = reinterpret_cast<float *>(bbox) + mifwi::round_up64(4LL * d.nshot, 64);
        int *lists = reinterpret_cast<int *>(xbuf + pl->xbuf_elems);
        hipLaunchKernelGGL(ec_build_slab_lists, dim3(d.nshot), dim3(256), 0, st, rec_cell, d.nrec, d.nz, d.nx,
                           pl->adj_NW, lists, lists + (long long)d.nshot * pl->adj_NW);
        EaParams c;
        memset(&c, 0, sizeof(c));
        c.nz = d.nz; c.nx = d.nx; c.ng = pl->ng; c.gp = pl->gp; c.pitch = pl->pitch;
        c.field_stride = (unsigned)pl->field_stride; c.shot_stride = pl->shot_stride;
        c.nshot = d.nshot; c.NW = pl->adj_NW; c.PL = pl->adj_PL;
        c.n_first = n_hi; c.n_last = n_lo; c.nt = d.nt;
        c.W = pl->W; c.wl = pl->wl; c.xr0 = pl->xr0; c.wx = pl->wx; c.fsurf = d.free_surface;
        c.zrows_max = pl->adj_zrows;
        c.psix_shot = 4LL * d.nz * pl->wx; c.psiz_shot = 4LL * 2 * pl->W * pl->gp; c.psix_elems = pl->psix_elems;
        c.mat = mat; c.pz = pz; c.px = px;
        c.fields = fields; c.psiA = psiA; c.psiB = psiB;
        c.S = snap; c.s_first = snap_first; c.s_step = snap_step;
        c.acc = acc;
        c.nsrc = d.nsrc; c.nrec = d.nrec; c.src_cell = src_cell; c.src_w = src_w;
        c.grad_f = want_f ? grad_f : nullptr;
        c.rec_cell = rec_cell; c.rec_w = rec_w; c.g_vx = g_vx; c.g_vz = g_vz;
        c.slab_cnt = lists; c.slab_list = lists + (long long)d.nshot * pl->adj_NW;
        c.rcv_direct = env_int("MIFWI_EL_ADJ_DIRECT", 1);
        c.dbg = env_int("MIFWI_EL_CL_DBG", 0);
        c.nap = env_int("MIFWI_POLL_NAP", mifwi::ceil_div(d.nz, c.NW) >= 8 ? 48 : 1);   // mifwi::poll_nap
        c.K = fd_weights(d.fd_order);
        c.xbuf = reinterpret_cast<unsigned long long *>(xbuf);
        c.err = reinterpret_cast<int *>(xbuf + pl->xbuf_elems - 64);
        c.xcc_tab = reinterpret_cast<int *>(xbuf + pl->xbuf_elems - 64 - pl->xcc_elems);
        const long long adj_state = pl->fields_elems + 2 * psi + nacc;
        float *backup = reinterpret_cast<float *>(lists) + pl->list_elems;
        rc = el_cluster_backup(work, adj_state, backup, flags, st);
        if (rc) return rc;
        // one attempt on the single-launch kernel; agent: granules published through the fabric (after a failed placement check)
        auto attempt = [&](bool agent) -> int {
            MIFWI_HIP_TRY(hipMemsetAsync(xbuf, 0, sizeof(float) * pl->xbuf_elems, st));
#ifdef MIFWI_ABLATIONS
            const char *trace_path = getenv("MIFWI_EL_CL_TRACE");
            const size_t trace_n = 64 * 8 * 16;
            c.trace = nullptr;
            if (trace_path && *trace_path) {
                MIFWI_HIP_TRY(hipMalloc(&c.trace, trace_n * sizeof(long long)));
                MIFWI_HIP_TRY(hipMemsetAsync(c.trace, 0, trace_n * sizeof(long long), st));
            }
#endif
            for (int s0 = 0; s0 < d.nshot && mifwi::fake_timeout() != 1; s0 += pl->adj_shots) {
                c.shot0 = s0;
                c.shot1 = std::min(d.nshot, s0 + pl->adj_shots);
                const dim3 grid(8 * pl->adj_NW * mifwi::ceil_div(c.shot1 - s0, 8));
                if (!agent && pl->adj_xh) {
                    if (pl->adj_ng == 1) hipLaunchKernelGGL((el_cluster_adj<1, false, true>), grid, dim3(kEcThreads), pl->adj_lds, st, c);
                    else hipLaunchKernelGGL((el_cluster_adj<2, false, true>), grid, dim3(kEcThreads), pl->adj_lds, st, c);
                }
                else if (pl->adj_ng == 1 && !agent) hipLaunchKernelGGL((el_cluster_adj<1, false>), grid, dim3(kEcThreads), pl->adj_lds, st, c);
                else if (pl->adj_ng == 1) hipLaunchKernelGGL((el_cluster_adj<1, true>), grid, dim3(kEcThreads), pl->adj_lds, st, c);
                else if (!agent) hipLaunchKernelGGL((el_cluster_adj<2, false>), grid, dim3(kEcThreads), pl->adj_lds, st, c);
                else hipLaunchKernelGGL((el_cluster_adj<2, true>), grid, dim3(kEcThreads), pl->adj_lds, st, c);
            }
            MIFWI_HIP_TRY(hipGetLastError());
            int err[4] = {0, 0, 0, 0};
            MIFWI_HIP_TRY(hipMemcpyAsync(err, c.err, sizeof(err), hipMemcpyDeviceToHost, st));
            MIFWI_HIP_TRY(hipStreamSynchronize(st));
#ifdef MIFWI_ABLATIONS
            if (c.trace) {
                std::vector<long long> h(trace_n);
                MIFWI_HIP_TRY(hipMemcpy(h.data(), c.trace, trace_n * sizeof(long long), hipMemcpyDeviceToHost));
                MIFWI_HIP_TRY(hipFree(c.trace));
                if (FILE *fp = fopen(trace_path, "a")) {
                    fprintf(fp, "# el_cluster_adj steps=%d\n", n_hi - n_lo + 1);
                    for (size_t i = 0; i < trace_n; i += 16) {
                        for (int k = 0; k < 16; ++k) fprintf(fp, "%lld ", h[i + k]);
                        fprintf(fp, "\n");
                    }
                    fclose(fp);
                }
            }
#endif
            const int verdict = mifwi::cluster_verdict(err, "elastic adjoint");
            return mifwi::fake_timeout() ? mifwi::kClusterTimedOut : verdict;
        };
        rc = attempt(false);
        if (rc == mifwi::kClusterMisplaced) {
            mifwi::note_agent_tier("elastic adjoint");
            rc = el_cluster_restore(work, adj_state, backup, flags, st);
            if (rc) return rc;
            rc = attempt(true);
        }
        if (rc == mifwi::kClusterTimedOut || rc == mifwi::kClusterMisplaced) {
            mifwi::note_fallback("elastic adjoint");
            rc = el_cluster_restore(work, adj_state, backup, flags, st);
            if (rc) return rc;
        } else if (rc != MIFWI_OK) {
            return rc;
        } else {
            per_step = false;
        }
    }
    // the two-launch form adds the adjoint sources inside el_adj_s (per-tile lists); the fused launch keeps the pre-pass
    const bool inj_in_tile = per_step && !pl->fused_adj && d.nrec > 0 && n_hi >= n_lo &&
                             env_int("MIFWI_EL_INJ_PREPASS", 0) == 0;
    if (inj_in_tile) {
        build_tile_lists();
        ps.tile_start = tile_start; ps.tile_list = tile_list;
    }
    // shot groups are independent: a few at a time keep fields, accumulators and materials in the Infinity Cache
    for (int g0 = 0; per_step && g0 < pl->ngroups; g0 += pl->pass_groups)
    for (int n = n_hi; n >= n_lo; --n) {
        const int cg = std::min(pl->pass_groups, pl->ngroups - g0);
        const int cs = std::min(cg * ps.gs, d.nshot - g0 * ps.gs);
        p.s0 = ps.s0 = g0 * ps.gs;
        // ping-pong of the adjoint memory variables is absolute in n (resumable ranges)
        const int par = (d.nt - 1 - n) & 1;
        float *rd = par ? psiB : psiA, *wr = par ? psiA : psiB;
        ps.psix = rd; ps.psiz = rd + pl->psix_elems;
        ps.psix_out = wr; ps.psiz_out = wr + pl->psix_elems;
        ps.S = const_cast<float *>(snap) + (long long)(n - snap_first) * snap_step;
        ps.inj_amp0 = g_vx + (long long)n * d.nshot * d.nrec;
        ps.inj_amp1 = g_vz + (long long)n * d.nshot * d.nrec;
        const bool force = d.source_type != 0;        // grad_f of a point force: v_bar sampled between S^T and V^T
        ps.smp_out0 = (want_f && !force) ? grad_f + (long long)n * d.nshot * d.nsrc : nullptr;
        p.psix = ps.psix; p.psiz = ps.psiz; p.psix_out = ps.psix_out; p.psiz_out = ps.psiz_out;
        if (pl->fused_adj) {
            // S^T + V^T in one launch: the adjoint fields ping-pong like the memory variables (absolute in n)
            ps.fields = par ? fieldsB : fields;
            ps.fields_out = par ? fields : fieldsB;
        }
        if (d.nrec > 0 && !inj_in_tile) {
            ElParams pq = ps;
            pq.gs = cs;
            hipLaunchKernelGGL(el_inject_adjsrc, dim3(mifwi::ceil_div(cs * d.nrec * d.ntap, 64)), dim3(64), 0, st, pq);
        }
        if (pl->fused_adj == 2) {
            const int tx = mifwi::ceil_div(pl->ng, WOG);
            ps.tiles_z = pl->walk_chunks;
            ps.walk_rows = pl->walk_rows;
            {   // the per-tile receiver lists of el_adj_s are not in use with this kernel: their room takes the dummy stores
                mifwi_elastic_layout lay;
                mifwi_elastic_plan_layout(pl, &lay);
                ps.trash = work + lay.work_backward_elems - pl->tile_elems;
            }
#ifdef MIFWI_ABLATIONS
            ps.walk_dbg = env_int("MIFWI_WALK_DBG", 0);
#endif
            const int ex = want_f ? mifwi::ceil_div(mifwi::ceil_div(ps.gs * ps.nsmp, kThreads), tx) : 0;
            const size_t lds = walk_lds_bytes(ps.gs);
#ifdef MIFWI_ABLATIONS
            const char *trace_path = getenv("MIFWI_WALK_TRACE");
            const size_t trace_n = (size_t)tx * pl->walk_chunks * cg * 8;
            ps.walk_trace = nullptr;
            if (trace_path && *trace_path && n == n_lo) {          // the last step of the range: caches warm
                MIFWI_HIP_TRY(hipMalloc(&ps.walk_trace, trace_n * sizeof(long long)));
                MIFWI_HIP_TRY(hipMemsetAsync(ps.walk_trace, 0, trace_n * sizeof(long long), st));
            }
#endif
            if (pl->snap_bf16) hipLaunchKernelGGL(el_adj_walk<true>, dim3(tx, pl->walk_chunks + ex, cg), dim3(kThreads), lds, st, ps);
            else hipLaunchKernelGGL(el_adj_walk<false>, dim3(tx, pl->walk_chunks + ex, cg), dim3(kThreads), lds, st, ps);
#ifdef MIFWI_ABLATIONS
            if (ps.walk_trace) {
                std::vector<long long> h(trace_n);
                MIFWI_HIP_TRY(hipStreamSynchronize(st));
                MIFWI_HIP_TRY(hipMemcpy(h.data(), ps.walk_trace, trace_n * sizeof(long long), hipMemcpyDeviceToHost));
                MIFWI_HIP_TRY(hipFree(ps.walk_trace));
                double sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (size_t i = 0; i < trace_n; ++i) sum[i & 7] += (double)h[i];
                if (FILE *fp = fopen(trace_path, "a")) {
                    // per item, mean over workgroups: A | barrier | B | barrier | C | barrier ; items ; whole kernel per workgroup
                    const double items = sum[6] > 0 ? sum[6] : 1.0, wgs = (double)(trace_n / 8);
                    fprintf(fp, "el_adj_walk wgs %.0f items/wg %.1f cycles/item: A %.0f bar %.0f B %.0f bar %.0f C %.0f bar %.0f | kernel cycles/wg %.0f\n",
                            wgs, items / wgs, sum[0] / items, sum[1] / items, sum[2] / items, sum[3] / items, sum[4] / items,
                            sum[5] / items, sum[7] / wgs);
                    fclose(fp);
                }
                ps.walk_trace = nullptr;
            }
#endif
            continue;
        }
        if (pl->fused_adj) {
            const int tx = mifwi::ceil_div(pl->ng, FTG), tz = mifwi::ceil_div(d.nz, FTZ);
            ps.tiles_z = tz;
            const int ex = want_f ? mifwi::ceil_div(mifwi::ceil_div(ps.gs * ps.nsmp, kThreads), tx) : 0;
            if (pl->snap_bf16) hipLaunchKernelGGL(el_adj_fused<true>, dim3(tx, tz + ex, cg), dim3(kThreads), 0, st, ps);
            else hipLaunchKernelGGL(el_adj_fused<false>, dim3(tx, tz + ex, cg), dim3(kThreads), 0, st, ps);
            continue;
        }
        {
            const int tx = mifwi::ceil_div(pl->ng, AGO), tz = mifwi::ceil_div(d.nz, ATZ);
            ps.tiles_z = tz;
            int ex = 0;
            if (want_f && !force) ex = mifwi::ceil_div(mifwi::ceil_div(ps.gs * ps.nsmp, kThreads), tx);
            if (pl->g_p && d.nrec > 0) {
                ElParams pq = ps;
                pq.gs = cs; pq.inj_amp0 = pl->g_p + (long long)n * d.nshot * d.nrec;
                hipLaunchKernelGGL(el_inject_pressure, dim3(mifwi::ceil_div(cs * d.nrec * d.ntap, 64)), dim3(64), 0,
                                   st, pq);
            }
            if (pl->snap_bf16) hipLaunchKernelGGL(el_adj_s<true>, dim3(tx, tz + ex, cg), dim3(kThreads), 0, st, ps);
            else hipLaunchKernelGGL(el_adj_s<false>, dim3(tx, tz + ex, cg), dim3(kThreads), 0, st, ps);
            if (want_f && force) {
                ElParams pq = ps;
                pq.gs = cs; pq.smp_out0 = grad_f + (long long)n * d.nshot * d.nsrc;
                hipLaunchKernelGGL(el_sample_force, dim3(mifwi::ceil_div(cs * d.nsrc, 64)), dim3(64), 0, st, pq,
                                   d.source_type);
            }
            hipLaunchKernelGGL(el_adj_v, dim3(tx, tz, cs), dim3(kThreads), 0, st, p);
        }
    }
    if (flags & MIFWI_FINALIZE) {
        const long long n5 = 5LL * pl->coef_elems;
        hipLaunchKernelGGL(el_finalize, dim3((unsigned)((n5 + 255) / 256)), dim3(256), 0, st, acc,
                           pl->ngroups, d.nz, pl->gp, pl->splane, pl->cl_adj ? 0 : pl->sblk, grad_mat);
    }
    MIFWI_HIP_TRY(hipGetLastError());
    return MIFWI_OK;
}

}  // extern "C"
