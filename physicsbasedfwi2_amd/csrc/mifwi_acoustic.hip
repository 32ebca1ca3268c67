// 2-D constant-density acoustic propagator for gfx950 (MI355X): forward, snapshot save,
// exact discrete adjoint + imaging.  Memory-bound 4th-order star stencil, MFMA unused.
//
// Replaces (reference tree): deepwave.scalar.Propagator forward/backward as called at
// models/networks.py:5408-5411,5449,5464,5491 and the Devito Forward/Gradient operators of
// seisgan/fwi/pde/seismic/acoustic/operators.py:54-89,127-165.
//
// One kernel per time step does: stencil + damping + source/adjoint-source injection +
// receiver sampling (+ snapshot store | + imaging).  Design points for CDNA4:
//   * a thread owns 4 consecutive cells (one 16-B lane access; a wave row = 1 KiB coalesced)
//     and marches RZ rows down z with the 5-row z window held in registers, so every
//     field value is fetched from L2/HBM once per step (+4/(LZ*RZ) z-halo re-reads);
//   * u+ overwrites u- in place (u- is only read at the centre cell): 3 streams of 4 B/cell
//     + the coefficient = the 16 B/cell-step algorithmic traffic;
//   * shots that share a thread group reuse the coefficient registers and, in the adjoint,
//     one gradient accumulator, so the gradient read-modify-write is amortised over gs shots;
//   * sparse injection is staged through a per-tile LDS image only in the tiles that the
//     shot's point bounding box touches; receiver sampling runs in extra workgroups of the
//     same launch (it reads the field the step only reads), so a step is ONE launch.
// The arithmetic is the same explicit fmaf chain as oracle/acoustic.c (compile with
// -ffp-contract=off) so wavefields can be compared bitwise.
#include "mifwi_common.h"

#include <atomic>

#include <type_traits>
#include <vector>

#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace {

constexpr int kThreads = 256;
constexpr float K0 = -2.5f;
constexpr float K1 = (float)(4.0 / 3.0);
constexpr float K2 = (float)(-1.0 / 12.0);

struct AcParams {
    int n0, n1, ng, pitch, gp;
    long long shot_stride;       // floats per shot wavefield = (n0+4)*pitch
    int nshot, gs;
    float c0, c1;
    const float *r, *q0, *q1;
    const float *cur;            // u^n      [nshot][n0+4][pitch]
    float *prev;                 // u^{n-1} -> u^{n+1}
    float *G;                    // snapshot slice of this step [nshot][n0][gp] (SAVE: w, IMAGE: r)
    float *acc;                  // [ngroups][n0][gp]
    const float *born_dr;        // Born pass (IMAGE launches only): [n0][gp] model perturbation, or NULL
    // injection into the new field
    int ninj, ntap_inj, inj_mode;    // mode 0: += a*r, G += a ; mode 1: += a*r*inv
    const int *inj_cell;
    const float *inj_w;
    const float *inj_amp;        // [nshot][ninj] for this step
    const int *inj_bbox;         // [nshot][4] i0min,i0max,i1min,i1max (max < min => none)
    // sampling of `cur`
    int nsmp, ntap_smp, smp_mode;    // mode 0: sum w*u ; mode 1: sum w*(1+q r)*u
    const int *smp_cell;
    const float *smp_w;
    float *smp_out;              // [nshot][nsmp] for this step (NULL: skip)
    int tiles_z;                 // grid.y rows that are stencil tiles; the rest sample
    int xcd;                     // 1: XCD-contiguous tile order (xcd_tile)
    int g0;                      // first shot group of this pass over the time range (blockIdx.z counts from it)
    // second-order C-PML (ac_step<..., PML = true>, mifwi_acoustic_cpml.h): the layer's term, region-compact
    int pmlW, pml_adj;           // layer width; 0: forward combination fma(c0, e0, c1 e1), 1: adjoint e0 + e1
    const float *pe0, *pe1;      // [nshot][2][W+2][gp], [nshot][n0][2][W+2]
};

__device__ __forceinline__ float comp(const float4 &v, int c)
{
    return c == 0 ? v.x : c == 1 ? v.y : c == 2 ? v.z : v.w;
}

#include "mifwi_acoustic_cpml.h"

// Workgroups go to the 8 XCDs round-robin in launch order and every XCD has its own L2: remap so that
// each XCD walks one contiguous row-major run of the (x, y) tiles of its z slice and halo rows of
// neighbouring tiles are L2 hits (bijection on [0, gridDim.x * gridDim.y); same map as the elastic kernels)
__device__ __forceinline__ void xcd_tile(const AcParams &p, int &bx, int &by)
{
    bx = (int)blockIdx.x; by = (int)blockIdx.y;
    if (!p.xcd) return;
    const unsigned gx = gridDim.x, n2 = gx * gridDim.y;
    const unsigned L = blockIdx.x + gx * blockIdx.y;
    const unsigned c = L & 7u, idx = L >> 3, q = n2 >> 3, r = n2 & 7u;
    const unsigned T = c * q + (c < r ? c : r) + idx;
    by = (int)(T / gx); bx = (int)(T - (unsigned)by * gx);
}

// ---- sampling workgroups (receivers in forward, source-gradient in backward) ----------------
__device__ void sample_points(const AcParams &p, int bx, int by)
{
    if (p.smp_out == nullptr) return;
    const int nrb = (int)(gridDim.y - p.tiles_z) * (int)gridDim.x;
    const int rb = (by - p.tiles_z) * (int)gridDim.x + bx;
    const int total = p.gs * p.nsmp;
    for (int e = rb * kThreads + (int)threadIdx.x; e < total; e += nrb * kThreads) {
        const int si = e / p.nsmp, ip = e - si * p.nsmp;
        const int s = (p.g0 + (int)blockIdx.z) * p.gs + si;
        if (s >= p.nshot) continue;
        const float *cur = p.cur + (long long)s * p.shot_stride;
        float a = 0.f;
        for (int t = 0; t < p.ntap_smp; ++t) {
            const long long ee = ((long long)s * p.nsmp + ip) * p.ntap_smp + t;
            const int cell = p.smp_cell[ee];
            if (cell < 0) continue;
            const int i0 = cell / p.n1, i1 = cell - i0 * p.n1;
            const float u = cur[(long long)(i0 + 2) * p.pitch + 4 + i1];
            float w = p.smp_w[ee];
            if (p.smp_mode == 1) {
                const float q = p.q0[i0] + p.q1[i1];
                w = w * (1.0f + q * p.r[(long long)i0 * p.gp + i1]);
            }
            a = fmaf(w, u, a);
        }
        p.smp_out[(long long)s * p.nsmp + ip] = a;
    }
}

template <int LX, int RZ, bool SAVE, bool IMAGE, bool PML = false>
__global__ __launch_bounds__(kThreads) void ac_step(const AcParams p)
{
    constexpr int LZ = kThreads / LX;
    constexpr int TZ = LZ * RZ;
    constexpr int TX = LX * 4;
    int bx, by;
    xcd_tile(p, bx, by);
    if (by >= p.tiles_z) {
        sample_points(p, bx, by);
        return;
    }
    __shared__ float inj[TZ][TX];

    const int lx = (int)threadIdx.x % LX, lz = (int)threadIdx.x / LX;
    const int g = bx * LX + lx;
    const int tile_i0 = by * TZ, tile_i1 = bx * TX;
    const int j0 = tile_i0 + lz * RZ;
    const bool active = (g < p.ng) && (j0 < p.n0);
    const int col = 4 + 4 * g;                      // first wavefield column of this group

    float4 rr[RZ], acc[RZ];
    float q0v[RZ];
    float4 q1v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (active) {
        q1v = *reinterpret_cast<const float4 *>(p.q1 + 4 * g);
#pragma unroll
        for (int rz = 0; rz < RZ; ++rz) {
            const int j = j0 + rz;
            if (j < p.n0) {
                rr[rz] = *reinterpret_cast<const float4 *>(p.r + (long long)j * p.gp + 4 * g);
                q0v[rz] = p.q0[j];
                if (IMAGE)
                    acc[rz] = p.born_dr ? make_float4(0.f, 0.f, 0.f, 0.f)
                                        : *reinterpret_cast<const float4 *>(
                                              p.acc + ((long long)(p.g0 + (int)blockIdx.z) * p.n0 + j) * p.gp + 4 * g);
            } else {
                rr[rz] = make_float4(0.f, 0.f, 0.f, 0.f);
                q0v[rz] = 0.f;
                if (IMAGE) acc[rz] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }

    for (int si = 0; si < p.gs; ++si) {
        const int s = (p.g0 + (int)blockIdx.z) * p.gs + si;
        if (s >= p.nshot) break;                    // block-uniform

        // ---- stage this shot's injection for this tile in LDS (block-uniform branch) -------
        bool has_inj = false;
        if (p.ninj > 0) {
            const int b0 = p.inj_bbox[4 * s + 0], b1 = p.inj_bbox[4 * s + 1];
            const int b2 = p.inj_bbox[4 * s + 2], b3 = p.inj_bbox[4 * s + 3];
            has_inj = (b0 < tile_i0 + TZ) && (b1 >= tile_i0) && (b2 < tile_i1 + TX) &&
                      (b3 >= tile_i1);
        }
        if (has_inj) {
            for (int e = (int)threadIdx.x; e < TZ * TX; e += kThreads) (&inj[0][0])[e] = 0.f;
            __syncthreads();
            const int total = p.ninj * p.ntap_inj;
            for (int e = (int)threadIdx.x; e < total; e += kThreads) {
                const long long ee = (long long)s * total + e;
                const int cell = p.inj_cell[ee];
                if (cell < 0) continue;
                const int i0 = cell / p.n1, i1 = cell - i0 * p.n1;
                const int t0 = i0 - tile_i0, t1 = i1 - tile_i1;
                if (t0 >= 0 && t0 < TZ && t1 >= 0 && t1 < TX) {
                    const float amp = p.inj_amp[(long long)s * p.ninj + e / p.ntap_inj];
                    atomicAdd(&inj[t0][t1], p.inj_w[ee] * amp);
                }
            }
            __syncthreads();
        }

        if (active) {
            const float *cur = p.cur + (long long)s * p.shot_stride;
            float *prev = p.prev + (long long)s * p.shot_stride;
            // z window rows j-2..j+1 for the first row; row j+2 is loaded in the loop
            float4 w0, w1, w2, w3, w4;
            {
                const float *c0p = cur + (long long)(j0 + 0) * p.pitch + col;   // row j0-2
                w0 = *reinterpret_cast<const float4 *>(c0p);
                w1 = *reinterpret_cast<const float4 *>(c0p + p.pitch);
                w2 = *reinterpret_cast<const float4 *>(c0p + 2 * (long long)p.pitch);
                w3 = *reinterpret_cast<const float4 *>(c0p + 3 * (long long)p.pitch);
            }
#pragma unroll
            for (int rz = 0; rz < RZ; ++rz) {
                const int j = j0 + rz;
                if (j < p.n0) {
                    const float *rowp = cur + (long long)(j + 2) * p.pitch + col;
                    w4 = *reinterpret_cast<const float4 *>(rowp + 2 * (long long)p.pitch);
                    const float2 L = *reinterpret_cast<const float2 *>(rowp - 2);
                    const float2 R = *reinterpret_cast<const float2 *>(rowp + 4);
                    float *prow = prev + (long long)(j + 2) * p.pitch + col;
                    const float4 up = *reinterpret_cast<const float4 *>(prow);
                    float4 Gv = make_float4(0.f, 0.f, 0.f, 0.f), drv = Gv;
                    if (IMAGE)          // snapshot stream: written once, read once -> non-temporal
                        Gv = mifwi::ldnt4(p.G + ((long long)s * p.n0 + j) * p.gp + 4 * g);
                    if (IMAGE && p.born_dr)
                        drv = *reinterpret_cast<const float4 *>(p.born_dr + (long long)j * p.gp + 4 * g);
                    const float xs[8] = {L.x, L.y, w2.x, w2.y, w2.z, w2.w, R.x, R.y};
                    float un[4], gk[4];
                    float pe[4] = {0.f, 0.f, 0.f, 0.f};        // C-PML: the layer's term of the four cells
                    if (PML) {
                        const int W2 = p.pmlW + 2;
                        float4 e0 = make_float4(0.f, 0.f, 0.f, 0.f);
                        const float *pe0 = p.pe0 + (long long)s * 2 * W2 * p.gp;
                        if (j < W2) e0 = *reinterpret_cast<const float4 *>(pe0 + (long long)j * p.gp + 4 * g);
                        else if (j >= p.n0 - W2)
                            e0 = *reinterpret_cast<const float4 *>(pe0 + (long long)(W2 + j - (p.n0 - W2)) * p.gp + 4 * g);
                        const float *pe1 = p.pe1 + ((long long)s * p.n0 + j) * 2 * W2;
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const int i1 = 4 * g + c;
                            float e1 = 0.f;
                            if (i1 < W2) e1 = pe1[i1];
                            else if (i1 >= p.n1 - W2 && i1 < p.n1) e1 = pe1[W2 + i1 - (p.n1 - W2)];
                            const float e0c = comp(e0, c);
                            pe[c] = p.pml_adj ? e0c + e1 : fmaf(p.c0, e0c, p.c1 * e1);
                        }
                    }
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float uc = xs[c + 2];
                        const float s01 = comp(w1, c) + comp(w3, c);
                        const float s02 = comp(w0, c) + comp(w4, c);
                        const float s11 = xs[c + 1] + xs[c + 3];
                        const float s12 = xs[c] + xs[c + 4];
                        const float l0 = fmaf(K1, s01, fmaf(K2, s02, K0 * uc));
                        const float l1 = fmaf(K1, s11, fmaf(K2, s12, K0 * uc));
                        float lap = fmaf(p.c0, l0, p.c1 * l1);
                        if (PML) lap += pe[c];
                        const float rv = comp(rr[rz], c);
                        const float q = q0v[rz] + comp(q1v, c);
                        const float qr = q * rv;
                        const float inv = 1.0f / (1.0f + qr);
                        const float upv = comp(up, c);
                        const float num = fmaf(rv, lap, fmaf(-(1.0f - qr), upv, 2.0f * uc));
                        float v = inv * num;
                        if (SAVE) gk[c] = inv * (fmaf(q, upv, lap) - q * v);
                        if (has_inj) {
                            const float a = inj[j - tile_i0][4 * lx + c];
                            if (p.inj_mode == 0) {
                                v += a * rv;
                                if (SAVE) gk[c] += a;
                            } else {
                                v += a * (rv * inv);
                            }
                        }
                        if (IMAGE && p.born_dr) v = fmaf(comp(drv, c), comp(Gv, c), v);   // Born source G^n dr
                        if (4 * g + c >= p.n1) v = 0.f;     // keep the right halo at zero
                        un[c] = v;
                    }
                    *reinterpret_cast<float4 *>(prow) = make_float4(un[0], un[1], un[2], un[3]);
                    if (SAVE)
                        mifwi::stnt4(p.G + ((long long)s * p.n0 + j) * p.gp + 4 * g,
                                     make_float4(gk[0], gk[1], gk[2], gk[3]));
                    if (IMAGE && !p.born_dr) {
                        acc[rz].x = fmaf(un[0], Gv.x, acc[rz].x);
                        acc[rz].y = fmaf(un[1], Gv.y, acc[rz].y);
                        acc[rz].z = fmaf(un[2], Gv.z, acc[rz].z);
                        acc[rz].w = fmaf(un[3], Gv.w, acc[rz].w);
                    }
                    w0 = w1; w1 = w2; w2 = w3; w3 = w4;
                }
            }
        }
        if (has_inj) __syncthreads();               // LDS image is reused by the next shot
    }

    if (IMAGE && active && !p.born_dr) {
#pragma unroll
        for (int rz = 0; rz < RZ; ++rz) {
            const int j = j0 + rz;
            if (j < p.n0)
                *reinterpret_cast<float4 *>(p.acc + ((long long)(p.g0 + (int)blockIdx.z) * p.n0 + j) * p.gp +
                                            4 * g) = acc[rz];
        }
    }
}

// ---- per-shot bounding box of a point set ------------------------------------------------------
__global__ void points_bbox(const int *cell, int npts_per_shot, int n1, int *bbox)
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

// ---- grad_r = (sum over shot groups of acc) * (1 + q r)/r --------------------------------------
__global__ void ac_finalize(const float *acc, int ngroups, int n0, int n1, int gp, const float *r,
                            const float *q0, const float *q1, float *grad)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)n0 * gp) return;
    const int i0 = (int)(idx / gp), i1 = (int)(idx - (long long)i0 * gp);
    float a = 0.f;
    for (int gidx = 0; gidx < ngroups; ++gidx) a += acc[(long long)gidx * n0 * gp + idx];
    float out = 0.f;
    if (i1 < n1) {
        const float rv = r[idx];
        const float q = q0[i0] + q1[i1];
        out = a * ((1.0f + q * rv) / rv);
    }
    grad[idx] = out;
}


// ================================================================================================
// CLUSTER kernels: wavefields resident in LDS for the whole time loop.
//
// Marmousi-sized shots (the sizes the reference actually runs: 151x200, 174x500, 100x300 ...) are
// far too small to keep 256 CUs busy with one launch per time step: such a step is bounded by the
// launch boundary (1.5 us) and the memory round trip, not by HBM bandwidth.  Here a shot is cut
// into NW row slabs, one workgroup (= one CU, ~130 KB of its 160 KB LDS) per slab, NW*nshot <= the
// CU count so every workgroup is resident, and ONE launch runs all time steps:
//   * both time levels of the slab (+2 halo rows) live in LDS; the coefficient r, the damping and
//     (adjoint) the gradient accumulator of a thread's cells live in registers for the whole run,
//     so the only per-step global traffic is the snapshot stream (4 B/cell) and the halo rows;
//   * after each step the two boundary rows are handed to the neighbouring slab through global
//     memory as self-validating 8-byte granules {epoch, value} written with agent-scope (sc1)
//     stores and polled with agent-scope loads (cdna_hip_programming.md, Guideline 16, form R2):
//     no flag, no fence, no grid barrier.  Granule slots are double-buffered on the epoch parity:
//     a slab may run ONE step ahead of a neighbour (publishing epoch n+1 only needs the
//     neighbour's epoch-n rows, not the neighbour's consumption of ours), but not two (epoch n+2
//     needs the neighbour's n+1, which the neighbour publishes after consuming our epoch n);
//   * every spin is bounded; on a time-out the workgroup raises a global error word that all other
//     workgroups also watch, so the grid always drains.
// The arithmetic per cell is the same fmaf chain as ac_step (bitwise identical results).
// ================================================================================================
constexpr int kClThreads = 1024;

constexpr int kClPmlLdsLimit = 160 * 1024 - 1024;    // dynamic LDS a C-PML launch may ask for (272 B of static LDS next to it)
constexpr int kClMaxNG = 4;                  // groups of 4 cells a thread may own
constexpr unsigned kClMaxSpin = 400000;
// Ablation builds: wave 0..15 of slab NW/2 of the first shot writes s_memtime at phase boundary k of steps 64..127 to
// trace[((it - 64) * 16 + wave) * 16 + k]; tools/cluster_trace.py turns them into a per-phase table.
#ifdef MIFWI_ABLATIONS
#define CL_STAMP(k)                                                                                        \
    do {                                                                                                   \
        if (p.trace && tr_on && it >= 64 && it < 128 && (t & 63) == 0)                                     \
            p.trace[((it - 64) * 16 + (t >> 6)) * 16 + (k)] = (long long)__builtin_readcyclecounter();     \
    } while (0)
// dbg bit 4096 (fault injection): slab 1 of the first shot stalls ~10 ms at every 64th step (see EC_LAGGARD)
#define CL_LAGGARD()                                                                                       \
    do {                                                                                                   \
        if ((kDbg(p) & 4096) && w == 1 && s == p.shot0 && (it & 63) == 1)                                  \
            for (int z_ = 0; z_ < 3000; ++z_) __builtin_amdgcn_s_sleep(127);                               \
    } while (0)
#else
#define CL_STAMP(k) do { } while (0)
#define CL_LAGGARD() do { } while (0)
#endif
// Publishes stay in the XCD's L2 (workgroup-scope stores; mifwi::same_xcd in mifwi_common.h checks the placement that
// makes this correct).  AG = true (last template argument of ac_cluster): agent-scope stores, through the fabric -
// correct on any placement; launched by the host after a failed placement check.
template <bool AG> struct ClScope { static constexpr int value = AG ? __HIP_MEMORY_SCOPE_AGENT : __HIP_MEMORY_SCOPE_WORKGROUP; };

struct ClParams {
    int n0, n1, ng, gp, pitch;
    long long shot_stride;
    int nshot, NW, PL;           // slabs per shot, LDS row pitch (floats)
    int rt;                      // rows of the first / last slab (0: even split), see slab_rows
    int shot0, shot1;            // shots [shot0, shot1) are handled by this launch
    int dbg;                     // timing experiments only: 1 = skip the halo hand-off, 2 = skip snapshots
    int rcv_plain;               // adjoint sources by a plain LDS read-add-write where every tap of a slab has a cell of its own
    int nap;                     // s_sleep units between poll passes (mifwi::poll_nap)
    int nt, n_first, n_last;     // forward: steps n_first..n_last-1 ; adjoint: k = n_first down to n_last
    float c0, c1;
    const float *r, *q0, *q1;
    float *ua, *ub;              // global state (same layout as the per-step kernels)
    float *G;                    // snapshot base; step n at G + (n - g_first) * g_step
    int g_first;
    long long g_step;
    float *acc;                  // adjoint: [nshot][n0][gp]
    const float *born_dr;        // Born pass (MODE 3): [n0][gp] model perturbation
    // few-point list handled by the cell owners (forward: sources -> injection + G term)
    int nsrc, ntap;
    const int *src_cell;
    const float *src_w;
    const float *f;              // forward: [nt][nshot][nsrc]
    // many-point list (forward: receivers sampled from u^n; adjoint: receivers injected)
    int nrec;
    const int *rec_cell;
    const float *rec_w;
    float *rec_out;              // forward: [nt][nshot][nrec]
    const float *grad_rec;       // adjoint: [nt][nshot][nrec]
    float *grad_f;               // adjoint: [nt][nshot][nsrc] or null
    const int *slab_cnt;         // adjoint: [nshot][NW] number of receiver taps in the slab
    const int *slab_list;        // adjoint: [nshot][NW][nrec*ntap] tap ids (shot-local)
    unsigned long long *xbuf;    // granules [nshot][NW][2 epoch slots][2 sides][2 rows][gp]
    int *err;
    int *xcc_tab;                        // [nshot][NW] XCC_ID + 1 of each slab's workgroup (mifwi::same_xcd)
    AcPml pml;                           // second-order C-PML (PML = true variants): strip / region arrays in global memory
    int pml_lds_floats;                  // dynamic LDS of the launch in floats: what a slab may carve layer arrays from (pml_place)
    int pml_own;                         // edge slabs run the layer of axis 0 in the own-group form (pml_own_*) if its planes fit
    int pml_late_e;                      // the groups of axis 1's region in slot 0, no barrier behind the layer's last phase
#ifdef MIFWI_ABLATIONS
    long long *trace;                    // phase time stamps of one workgroup (MIFWI_AC_CL_TRACE), see CL_STAMP
#endif
};

// Row slabs of a shot.  rt == 0: n0 rows split evenly over NW slabs.  rt > 0 (NW >= 3): the first and
// the last slab hold rt rows each - the absorbing layer, whose cells cost an IEEE division per step -
// and the other NW-2 slabs share the rest evenly, so that the all-sponge slabs do not set the pace.
__host__ __device__ __forceinline__ void slab_rows(int n0, int NW, int rt, int w, int &r0, int &rows)
{
    if (rt > 0) {
        if (w == 0) { r0 = 0; rows = rt; return; }
        if (w == NW - 1) { r0 = n0 - rt; rows = rt; return; }
        const int m = n0 - 2 * rt, k = NW - 2, v = w - 1;
        const int base = m / k, rem = m - base * k;
        rows = base + (v < rem ? 1 : 0);
        r0 = rt + v * base + (v < rem ? v : rem);
        return;
    }
    const int base = n0 / NW, rem = n0 - base * NW;
    rows = base + (w < rem ? 1 : 0);
    r0 = w * base + (w < rem ? w : rem);
}

__host__ __device__ __forceinline__ int slab_of_row(int n0, int NW, int rt, int i0)
{
    int m = n0, k = NW, off = 0, first = 0;
    if (rt > 0) {
        if (i0 < rt) return 0;
        if (i0 >= n0 - rt) return NW - 1;
        m = n0 - 2 * rt; k = NW - 2; off = rt; first = 1;
    }
    const int base = m / k, rem = m - base * k, i = i0 - off;
    return first + ((i < rem * (base + 1)) ? i / (base + 1) : rem + (i - rem * (base + 1)) / base);
}

// lists of the receiver taps that fall into each slab (adjoint injection), one block per shot
__global__ void cl_build_slab_lists(const int *rec_cell, int ntaps, int n0, int n1, int NW, int rt,
                                    int *slab_cnt, int *slab_list)
{
    const int s = blockIdx.x;
    __shared__ int cnt[64];
    if ((int)threadIdx.x < NW) cnt[threadIdx.x] = 0;
    __syncthreads();
    for (int e = threadIdx.x; e < ntaps; e += blockDim.x) {
        const int cell = rec_cell[(long long)s * ntaps + e];
        if (cell < 0) continue;
        const int w = slab_of_row(n0, NW, rt, cell / n1);
        const int pos = atomicAdd(&cnt[w], 1);
        slab_list[((long long)s * NW + w) * ntaps + pos] = e;
    }
    __syncthreads();
    if ((int)threadIdx.x < NW) slab_cnt[s * NW + threadIdx.x] = cnt[threadIdx.x];
}

// one stencil update of a group of 4 cells whose operands sit in LDS.  Same fmaf chain as ac_step,
// written on 2-wide vectors so that hipcc emits packed fp32 VALU (v_pk_fma/mul/add_f32: two cells
// per instruction, IEEE per lane, so results stay bitwise identical to the scalar form).
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

template <bool WANT_G, bool PML = false>
__device__ __forceinline__ void cl_update(const float *c, const float *pp, int PL, const float4 &rv4,
                                          float q0, const float4 &q1, bool damped, float c0, float c1,
                                          int ncol_valid, float (&un)[4], float (&gk)[4],
                                          const float4 &pe = make_float4(0.f, 0.f, 0.f, 0.f))
{
    const float4 w2 = *reinterpret_cast<const float4 *>(c);
    const float4 w0 = *reinterpret_cast<const float4 *>(c - 2 * PL);
    const float4 w1 = *reinterpret_cast<const float4 *>(c - PL);
    const float4 w3 = *reinterpret_cast<const float4 *>(c + PL);
    const float4 w4 = *reinterpret_cast<const float4 *>(c + 2 * PL);
    const float2 Lh = *reinterpret_cast<const float2 *>(c - 2);
    const float2 Rh = *reinterpret_cast<const float2 *>(c + 4);
    const float4 up = *reinterpret_cast<const float4 *>(pp);
    const f2 k0 = {K0, K0}, k1 = {K1, K1}, k2 = {K2, K2}, vc0 = {c0, c0}, vc1 = {c1, c1};
    const f2 two = {2.0f, 2.0f}, mone = {-1.0f, -1.0f};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        // cells 2h, 2h+1
        const f2 uc = h == 0 ? f2{w2.x, w2.y} : f2{w2.z, w2.w};
        const f2 zm1 = h == 0 ? f2{w1.x, w1.y} : f2{w1.z, w1.w};
        const f2 zp1 = h == 0 ? f2{w3.x, w3.y} : f2{w3.z, w3.w};
        const f2 zm2 = h == 0 ? f2{w0.x, w0.y} : f2{w0.z, w0.w};
        const f2 zp2 = h == 0 ? f2{w4.x, w4.y} : f2{w4.z, w4.w};
        const f2 xm1 = h == 0 ? f2{Lh.y, w2.x} : f2{w2.y, w2.z};
        const f2 xp1 = h == 0 ? f2{w2.y, w2.z} : f2{w2.w, Rh.x};
        const f2 xm2 = h == 0 ? f2{Lh.x, Lh.y} : f2{w2.x, w2.y};
        const f2 xp2 = h == 0 ? f2{w2.z, w2.w} : f2{Rh.x, Rh.y};
        const f2 rv = h == 0 ? f2{rv4.x, rv4.y} : f2{rv4.z, rv4.w};
        const f2 upv = h == 0 ? f2{up.x, up.y} : f2{up.z, up.w};
        const f2 s01 = zm1 + zp1, s02 = zm2 + zp2, s11 = xm1 + xp1, s12 = xm2 + xp2;
        const f2 l0 = pk_fma(k1, s01, pk_fma(k2, s02, k0 * uc));
        const f2 l1 = pk_fma(k1, s11, pk_fma(k2, s12, k0 * uc));
        f2 lap = pk_fma(vc0, l0, vc1 * l1);
        if (PML) lap = lap + (h == 0 ? f2{pe.x, pe.y} : f2{pe.z, pe.w});       // the layer's term (exact zero outside it)
        f2 v, g;
        if (damped) {
            const f2 q = f2{q0, q0} + (h == 0 ? f2{q1.x, q1.y} : f2{q1.z, q1.w});
            const f2 qr = q * rv;
            const f2 den = f2{1.0f, 1.0f} + qr;
            const f2 inv = {1.0f / den.x, 1.0f / den.y};
            const f2 num = pk_fma(rv, lap, pk_fma(-(f2{1.0f, 1.0f} - qr), upv, two * uc));
            v = inv * num;
            g = inv * (pk_fma(q, upv, lap) - q * v);
        } else {
            // q == 0 on all four cells: inv == 1 exactly, the general formula reduces bit for bit to this
            v = pk_fma(rv, lap, pk_fma(mone, upv, two * uc));
            g = lap;
        }
        // columns >= n1 hold r = 0 and zero fields, so v is exactly 0 there without a mask
        (void)ncol_valid;
        un[2 * h] = v.x;
        un[2 * h + 1] = v.y;
        if (WANT_G) { gk[2 * h] = g.x; gk[2 * h + 1] = g.y; }
    }
}

// Value the compiler must treat as unknown here.  Applied once per time step to a thread's group
// offsets: the global / LDS addresses derived from them are then recomputed where they are used
// (an add or two) instead of being hoisted out of the time loop into registers of their own, which
// spilled - and every spill reload in the loop is a full s_waitcnt vmcnt(0) on the wave.
__device__ __forceinline__ int cl_opaque(int x)
{
    asm volatile("" : "+v"(x));
    return x;
}

// SLOW = false: at most one source per shot and at most kClThreads receivers / sources (decided by the
// host from the sizes): every sparse point has a thread of its own and the rescanning paths are not even
// compiled in - they cost ~8 % of the C2 gradient pass in SGPR/VGPR spills alone.  SLOW = true: general.
// The layer's phases of one step inside a slab (PML variants): exactly the thin launches of the per-step family
// (mifwi_acoustic_cpml.h: same cell functions, same bits), run by the slab's threads over the slab's own strip / region
// cells with the wavefield read from LDS.  Memory variables, exchanged values (psi', P, Q) and the layer's term e
// travel through global memory between the phases.  Writers and readers are waves of ONE workgroup, i.e. of one CU and
// one (write-through) vector L1: workgroup scope is all the ordering they need, which is what __syncthreads() gives
// (an agent-scope release / acquire pair here writes back and invalidates L2 lines at every phase boundary: measured
// 224 us per step instead of 5).  Edge slabs hold W + 2 rows (plan), so everything a phase reads of u lies in the
// slab's own rows - the phases run before the hand-off poll.
__device__ __forceinline__ void cl_pml_sync() { __syncthreads(); }
template <class F>
__device__ __forceinline__ void cl_pml_cells(const AcPml &m, int w, int NW, int r0, int R, int t, bool skip0, F f)
{
    const unsigned W2 = (unsigned)m.W + 2u, ng = (unsigned)m.gp / 4u;
    if ((w == 0 || w == NW - 1) && !skip0) {                       // axis 0: the W + 2 rows of this end of the grid, groups of four cells
        const unsigned base = w == 0 ? 0u : W2 * ng;
        for (unsigned e = (unsigned)t; e < W2 * ng; e += kClThreads) f(pml_cell(m, base + e, 0), 0);
    }
    for (unsigned e = (unsigned)t; e < 2u * W2 * (unsigned)R; e += kClThreads)       // axis 1: two runs per own row
        f(pml_cell(m, (unsigned)r0 * 2u * W2 + e, 1), 1);
}

template <int MODE, bool SLOW, bool AG, bool PML = false>   // MODE 0: forward, 1: forward + snapshots, 2: adjoint + imaging, 3: Born (source G^n dr)
__global__ __launch_bounds__(kClThreads) void ac_cluster(const ClParams p)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // XCD-aware mapping: consecutive linear ids go to different XCDs, so give all slabs of a shot
    // ids that are congruent mod 8 (speed only; correctness never depends on placement)
    const int L = (int)blockIdx.x;
    const int xcd = L & 7, k = L >> 3;
    const int w = k % p.NW, s = p.shot0 + xcd + 8 * (k / p.NW);
    if (s >= p.shot1) return;
    // ablation builds only: slab 1 of the first shot never shows up (a workgroup that was not resident in time) -
    // its neighbours must time out, publish the error word and let the host fall back
    if ((kDbg(p) & 64) && w == 1 && s == p.shot0) return;
    const int t = (int)threadIdx.x;
    if (!AG && !mifwi::same_xcd(p.xcc_tab, s, p.NW, w, t, p.err, kClMaxSpin, kDbg(p) & 128)) return;
    constexpr bool adj = (MODE == 2);
    int r0, R;
    slab_rows(p.n0, p.NW, p.rt, w, r0, R);
    const int PL = p.PL, LR = R + 4;
    float *bufA = lds, *bufB = lds + LR * PL;     // rows: 0,1 top halo | 2..R+1 own | R+2,R+3 bottom halo
    float *ldq1 = bufB + LR * PL;                 // damping tables (constant over the run): q1[gp], q0[R]
    float *ldq0 = ldq1 + p.gp;
    // C-PML: the layer's arrays of this slab that fit behind the planes live in LDS (pml_place: exchanged ones first), the
    // others in global memory.  `m` is p.pml with the pointers of the LDS-resident arrays redirected to the slab's own
    // part of each array; m.lds / m.f* tell the cell functions which index that part starts at (pml_off).
    AcPml m = p.pml;
    bool own = false;                             // edge slab in the own-group form; its planes (Psi | P, Q)
    float *plP = nullptr, *plQ = nullptr;
    if (PML) {
        const long long W = m.W, W2 = m.W + 2;
        const bool edge = w == 0 || w == p.NW - 1;
        long long base = 2LL * LR * PL + p.gp + ((R + 3) & ~3);
        long long used = 0;
        float *q = ldq0 + ((R + 3) & ~3);
        // edge slabs: the own-group form of the layer of axis 0 (mifwi_acoustic_cpml.h) if its planes fit behind the field planes
        own = edge && p.pml_own && R == m.W + 2 && base + pml_own_floats(adj, R, PL) <= (long long)p.pml_lds_floats;
        if (own) {                           // planes start as zeros and only the strip's cells are ever written: zero elsewhere for good
            plP = q; plQ = q + (adj ? LR * PL : 0);
            q += pml_own_floats(adj, R, PL); base += pml_own_floats(adj, R, PL);
            for (int e = t; e < (int)pml_own_floats(adj, R, PL); e += kClThreads) plP[e] = 0.f;
        }
        m.lds = pml_place(adj, edge, R, m.W, m.gp, (long long)p.pml_lds_floats - base, &used, own);
        m.f0s = w == 0 ? 0 : W * m.gp; m.f0r = w == 0 ? 0 : W2 * m.gp;      // first index of this slab's part
        m.f1s = (long long)r0 * 2 * W; m.f1r = (long long)r0 * 2 * W2;
        auto take = [&](int bit, float *&ptr) {
            if (m.lds & (unsigned)bit) { ptr = q; q += pml_lds_floats(bit, R, m.W, m.gp); }
        };
        if (!adj) {                          // the order of pml_place
            take(PML_A1, m.A1); take(PML_E1, m.e1); take(PML_A0, m.A0); take(PML_E0, m.e0); take(PML_B1, m.B1); take(PML_B0, m.B0);
        } else {
            take(PML_P1, m.P1); take(PML_Q1, m.Q1); take(PML_E1, m.e1); take(PML_P0, m.P0); take(PML_Q0, m.Q0);
            take(PML_E0, m.e0); take(PML_A1, m.A1); take(PML_B1, m.B1); take(PML_A0, m.A0); take(PML_B0, m.B0);
        }
        // memory variables that live in LDS for the run: in from the global state (out again at the end)
        auto copy = [&](int bit, float *dst, const float *src, long long first, long long n, long long shot_stride) {
            if (m.lds & (unsigned)bit)
                for (long long e = t; e < n; e += kClThreads) dst[e] = src[(long long)s * shot_stride + first + e];
        };
        copy(PML_A1, m.A1, p.pml.A1, m.f1s, 2 * W * R, m.s1); copy(PML_B1, m.B1, p.pml.B1, m.f1s, 2 * W * R, m.s1);
        if (edge) { copy(PML_A0, m.A0, p.pml.A0, m.f0s, W * m.gp, m.s0); copy(PML_B0, m.B0, p.pml.B0, m.f0s, W * m.gp, m.s0); }
        if (own && !adj) {                   // Psi of the strip's rows into its plane (zeroed above, by other threads: barrier first)
            __syncthreads();
            const float *src = p.pml.A0 + (long long)s * m.s0 + m.f0s;
            for (int e = t; e < (int)(W * m.gp); e += kClThreads) {
                const int row = e / m.gp, col = e - row * m.gp;
                plP[(row + (w == 0 ? 0 : 2) + 2) * PL + 4 + col] = src[e];
            }
        }
    }
    const int ngrp = R * p.ng;
    // row order with the four boundary rows first, so that a thread's slot 0 covers every group the
    // neighbours wait for: 0, 1, R-2, R-1, 2, 3, ..., R-3   (R >= 4 is guaranteed by the plan)
    auto row_of = [&](int k_) { return k_ < 2 ? k_ : (k_ < 4 ? R - 4 + k_ : k_ - 2); };

    // ---- per-thread constants: owned groups, coefficients, accumulators -----------------------
    int loff[kClMaxNG];                 // LDS offset of the group in a level buffer
    float4 rr[kClMaxNG], acc[kClMaxNG];
    int jg[kClMaxNG];                   // (grid row << 12) | group
    auto goff = [&](int i_) { return (long long)(jg[i_] >> 12) * p.gp + 4 * (jg[i_] & 4095); };
    auto goff_of = [&](int jg_) { return (unsigned)((jg_ >> 12) * p.gp + 4 * (jg_ & 4095)); };
    unsigned dampmask = 0;
    int nown = 0;
    // Group -> thread assignment.  After the boundary rows (kept in slot 0 for the early hand-off) the
    // UNDAMPED groups come first and the damped ones (sponge cells: an IEEE division per cell and
    // step) last, so that most waves hold one kind only and run one of the two update paths instead
    // of both.  Which thread updates a group does not change any result.  The permutation lives in
    // the (not yet staged) plane memory.
    int *perm = reinterpret_cast<int *>(lds);
    __shared__ int perm_cnt[3];
    // C-PML plans: the groups at the two ends of a row (the region of axis 1) go right behind the boundary rows, i.e. into
    // slot 0 - the slot that is updated behind barrier A.  If all of them fit there (late_e), the layer's last phase (the
    // term e of axis 1, read by exactly these groups) needs no barrier of its own: A orders it.
    bool late_e = false;
    {
        const int nb = min(4 * p.ng, ngrp);
        int nin1 = 0;
        if (PML) {
            int per_row = 0;
            for (int g = 0; g < p.ng; ++g) per_row += (4 * g < m.W + 2 || 4 * g + 3 >= m.n1 - (m.W + 2)) ? 1 : 0;
            nin1 = per_row * max(R - 4, 0);
            late_e = p.pml_late_e && nb + nin1 <= kClThreads;
            if (!late_e) nin1 = 0;
        }
        if (t < 3) perm_cnt[t] = 0;
        __syncthreads();
        for (int gi = t; gi < ngrp; gi += kClThreads) {
            if (gi < nb) { perm[gi] = gi; continue; }
            const int kr = gi / p.ng, g = gi - kr * p.ng;
            const float4 q1 = *reinterpret_cast<const float4 *>(p.q1 + 4 * g);
            // (C-PML plans have no sponge: there the groups of the layer, whose update adds the layer's term, go last)
            const bool dmp = PML ? pml_layer_group(m, r0 + row_of(kr), g)
                                 : p.q0[r0 + row_of(kr)] != 0.f || q1.x != 0.f || q1.y != 0.f || q1.z != 0.f || q1.w != 0.f;
            if (PML && late_e && (4 * g < m.W + 2 || 4 * g + 3 >= m.n1 - (m.W + 2))) perm[nb + atomicAdd(&perm_cnt[2], 1)] = gi;
            else if (!dmp) perm[nb + nin1 + atomicAdd(&perm_cnt[0], 1)] = gi;
            else perm[ngrp - 1 - atomicAdd(&perm_cnt[1], 1)] = gi;
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < kClMaxNG; ++i) {
        const int gslot = t + i * kClThreads;
        loff[i] = 0; jg[i] = 0;
        rr[i] = acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gslot < ngrp) {
            nown = i + 1;
            const int gi = perm[gslot];
            const int kr = gi / p.ng, g = gi - kr * p.ng;
            const int lrw = row_of(kr), j = r0 + lrw;
            loff[i] = (lrw + 2) * PL + 4 + 4 * g;
            jg[i] = (j << 12) | g;
            rr[i] = *reinterpret_cast<const float4 *>(p.r + (long long)j * p.gp + 4 * g);
            const float4 q1 = *reinterpret_cast<const float4 *>(p.q1 + 4 * g);
            const float q0 = p.q0[j];
            if (PML ? pml_layer_group(m, j, g) : (q0 != 0.f || q1.x != 0.f || q1.y != 0.f || q1.z != 0.f || q1.w != 0.f))
                dampmask |= 1u << i;             // C-PML plans: "a group of the layer" (their damping tables are zero)
            if (PML && (4 * g < m.W + 2 || 4 * g + 3 >= m.n1 - (m.W + 2)))
                dampmask |= 16u << i;            // ... and "some of its cells lie in the region of axis 1" 
            if (adj) acc[i] = *reinterpret_cast<const float4 *>(p.acc + ((long long)s * p.n0 + j) * p.gp + 4 * g);
            if (MODE == 3) acc[i] = *reinterpret_cast<const float4 *>(p.born_dr + (long long)j * p.gp + 4 * g);
        }
    }
    // ---- per-thread sparse points (at most one of each kind per thread in this path) ------------
    // forward: source tap inside one of my groups; receiver sampled by me.  adjoint: receiver tap
    // injected by me (slab list); source sampled by me for grad_f.
    int src_slot = -1, src_comp = 0, src_e = -1;        // forward injection (owner side)
    float src_wt = 0.f;
    int smp_off = -1;                                    // sampling of `cur` (point index = t)
    float smp_w = 0.f;
    // adjoint injection into LDS: LDS offset (16 bits: a level buffer is < 20 000 floats) | turn (2 bits) | receiver id (12 bits, fast variant: nrec <= 1024) |
    // "lies in a row the neighbours wait for" (bit 30), one register instead of three; -1 = none
    int inj_pack = -1;
    float inj_scale = 0.f;
    // unpacked from an opaque copy at every use: a hoisted field would take a register of its own again
    auto inj_off = [&]() { return cl_opaque(inj_pack) & 0xffff; };
    auto inj_turn = [&]() { return (cl_opaque(inj_pack) >> 16) & 3; };
    auto inj_id = [&]() { return (cl_opaque(inj_pack) >> 18) & 0xfff; };
    constexpr bool slow_sparse = SLOW;                   // more points than one per thread: rescan per step
    // adjoint sources without LDS float atomics: inj_turn() = the turn in which this thread adds its tap to its cell (0
    // for a tap that has the cell to itself; taps sharing a cell take turns), tap_rounds = turns needed by the slab
    // (workgroup-uniform; 0: atomics, the general path)
    int tap_rounds = 0;
    if (!adj) {
        for (int e = 0; e < p.nsrc; ++e) {
            const int cell = p.src_cell[(long long)s * p.nsrc + e];
            if (cell < 0) continue;
            const int i0 = cell / p.n1, i1 = cell - i0 * p.n1;
            if (i0 < r0 || i0 >= r0 + R) continue;
            const long long want = (long long)i0 * p.gp + (i1 & ~3);
#pragma unroll
            for (int i = 0; i < kClMaxNG; ++i)
                if (i < nown && goff(i) == want) {
                    src_slot = i; src_comp = i1 & 3; src_e = e;
                    src_wt = p.src_w[(long long)s * p.nsrc + e];
                }
        }
        if (p.rec_out != nullptr && t < p.nrec) {
            const int cell = p.rec_cell[(long long)s * p.nrec + t];
            if (cell >= 0) {
                const int i0 = cell / p.n1, i1 = cell - i0 * p.n1;
                if (i0 >= r0 && i0 < r0 + R) {
                    smp_off = (i0 - r0 + 2) * PL + 4 + i1;
                    smp_w = p.rec_w[(long long)s * p.nrec + t];
                }
            } else if (w == 0) {
                smp_off = -2;                            // inactive tap: slab 0 writes the zero
            }
        }
    } else {
        const int cnt = p.slab_cnt[s * p.NW + w];
        if (t < cnt) {
            const int id = p.slab_list[((long long)s * p.NW + w) * p.nrec + t];
            const int cell = p.rec_cell[(long long)s * p.nrec + id];
            const int i0 = cell / p.n1, i1 = cell - i0 * p.n1;
            const float rv = p.r[(long long)i0 * p.gp + i1];
            const float q = p.q0[i0] + p.q1[i1];
            const float inv = 1.0f / (1.0f + q * rv);
            const bool edge = (i0 - r0 < 2) || (i0 - r0 >= R - 2);
            inj_pack = ((i0 - r0 + 2) * PL + 4 + i1) | ((id & 0xfff) << 18) | (edge ? (1 << 30) : 0);
            inj_scale = rv * inv;
            smp_w = p.rec_w[(long long)s * p.nrec + id];          // reused as the tap weight
        }
        // The injection is a plain LDS read-add-write (the same single rounding) instead of ds_add_f32: a float atomic
        // takes the LDS ~1500 clocks to drain, every step, in the slab that holds the receivers - the one all the others
        // wait for.  Taps that share a cell (bilinear taps of neighbouring receivers) take turns, lowest receiver number
        // first, a barrier between turns: a fixed order of additions, unlike the atomics.  The turns are dealt once,
        // through the (still empty) field plane.
        if (!slow_sparse && cnt > 0 && p.rcv_plain) {
            int *ib = reinterpret_cast<int *>(bufA);
            const int off = inj_pack & 0xffff, key = (inj_pack >> 18) & 0xfff;      // receiver number
            bool waiting = inj_pack >= 0;
            constexpr int kMaxRounds = 4;                 // bilinear taps of neighbouring receivers: 2-4 per cell (two bits)
            int r = 0;
            for (; r < kMaxRounds; ++r) {
                if (waiting) ib[off] = 0x7fffffff;
                __syncthreads();
                if (waiting) atomicMin(&ib[off], key);
                __syncthreads();
                if (waiting && ib[off] == key) { inj_pack |= r << 16; waiting = false; }
                if (!__syncthreads_or(waiting ? 1 : 0)) break;
            }
            tap_rounds = r < kMaxRounds ? r + 1 : 0;      // more than kMaxRounds taps in one cell: atomics
        }
        if (p.grad_f != nullptr && t < p.nsrc) {
            const int cell = p.src_cell[(long long)s * p.nsrc + t];
            if (cell >= 0) {
                const int i0 = cell / p.n1, i1 = cell - i0 * p.n1;
                if (i0 >= r0 && i0 < r0 + R) {
                    const float q = p.q0[i0] + p.q1[i1];
                    smp_off = (i0 - r0 + 2) * PL + 4 + i1;
                    src_wt = p.src_w[(long long)s * p.nsrc + t] * (1.0f + q * p.r[(long long)i0 * p.gp + i1]);
                }
            } else if (w == 0) {
                smp_off = -2;
            }
        }
    }
    __syncthreads();

    for (int e = t; e < p.gp; e += kClThreads) ldq1[e] = p.q1[e];
    for (int e = t; e < R; e += kClThreads) ldq0[e] = p.q0[r0 + e];
    // ---- load the slab (+ halo rows) of both time levels from the global state ------------------
    // buffer parity is absolute in the step index, as in the per-step path
    const int par0 = adj ? ((p.nt - 1 - p.n_first) & 1) : (p.n_first & 1);
    const float *gcur = (par0 ? p.ub : p.ua) + (long long)s * p.shot_stride;
    const float *gprev = (par0 ? p.ua : p.ub) + (long long)s * p.shot_stride;
    for (int e = t; e < LR * (PL / 4); e += kClThreads) {
        const int lr = e / (PL / 4), lg = e - lr * (PL / 4);      // lg = 0 is the left halo group
        const int j = r0 - 2 + lr, g = lg - 1;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        if (j >= 0 && j < p.n0 && g >= 0 && g < p.ng) {
            const long long o = (long long)(j + 2) * p.pitch + 4 + 4 * g;
            a = *reinterpret_cast<const float4 *>(gcur + o);
            b = *reinterpret_cast<const float4 *>(gprev + o);
        }
        *reinterpret_cast<float4 *>(bufA + lr * PL + 4 * lg) = a;
        *reinterpret_cast<float4 *>(bufB + lr * PL + 4 * lg) = b;
    }
    __syncthreads();
    float *cur = bufA, *prv = bufB;

    // hand-off assignments of this thread (constant over the run): granule e = t + k*kClThreads
    constexpr int kGr = 3;
    int rcv_lo[kGr];                     // LDS offset of the halo cell (<0: none); bit 30 of the
    unsigned upmask = 0;                 // mask: granule comes from the upper neighbour
#pragma unroll
    for (int kk = 0; kk < kGr; ++kk) {
        const int e = t + kk * kClThreads;
        rcv_lo[kk] = -1;
        if (e < 4 * p.gp) {
            const int side = e / (2 * p.gp), rem = e - side * 2 * p.gp;
            const int row = rem / p.gp, col = rem - row * p.gp;
            if (!((side == 0 && w == 0) || (side == 1 && w == p.NW - 1))) {
                rcv_lo[kk] = ((side == 0) ? row : R + 2 + row) * PL + 4 + col;
                if (side == 0) upmask |= 1u << kk;
            }
        }
    }
    // published cell = the own boundary cell two rows inside the halo cell; the neighbour's slot holds
    // our halo at the mirrored side: index e +- 2*gp
    auto pub_off = [&](int kk) { return rcv_lo[kk] + (((upmask >> kk) & 1u) ? 2 * PL : -2 * PL); };
    auto src_idx = [&](int kk) { return t + kk * kClThreads + (((upmask >> kk) & 1u) ? 2 * p.gp : -2 * p.gp); };
    const long long xslab = 8LL * p.gp;                                   // 2 slots x 4 rows x gp
    unsigned long long *xmine0 = p.xbuf + ((long long)s * p.NW + w) * xslab;
    const unsigned long long *xup0 = xmine0 - xslab;                      // valid if w > 0
    const unsigned long long *xdn0 = xmine0 + xslab;                      // valid if w < NW-1
    const int nsteps = adj ? (p.n_first - p.n_last + 1) : (p.n_last - p.n_first);
    const long long plane = (long long)s * p.n0 * p.gp;
    bool failed = false;
    const bool do_x = p.NW > 1 && !(kDbg(p) & 1);
    // software prefetch of the next step's global operands (issued before this step's stores, so
    // that they do not queue behind them): source / adjoint-source amplitude and, in the adjoint,
    // the snapshot values the imaging condition multiplies with
    float amp_next = 0.f;
    float4 Gbuf[kClMaxNG];              // adjoint: snapshot values of the coming step; forward: values to store
    auto prefetch = [&](int it_) {
        if (it_ >= nsteps) return;
        const int n_ = adj ? (p.n_first - it_) : (p.n_first + it_);
        if (!adj) {
            if (src_slot >= 0) amp_next = (p.f + ((long long)n_ * p.nshot + s) * p.nsrc)[cl_opaque(src_e)];
        } else {
            if (inj_pack >= 0) amp_next = (p.grad_rec + ((long long)n_ * p.nshot + s) * p.nrec)[inj_id()];
        }
        if (adj || MODE == 3) {
            // adjoint: G^{n-1} for the imaging condition; Born: G^n, the source term of this step
            const float *Gq = p.G + (long long)((adj ? n_ - 1 : n_) - p.g_first) * p.g_step + plane;
#pragma unroll
            for (int i = 0; i < kClMaxNG; ++i)
                if (i < nown) Gbuf[i] = mifwi::ldnt4(Gq + goff_of(cl_opaque(jg[i])));
        }
    };
    prefetch(0);
#ifdef MIFWI_ABLATIONS
    const bool tr_on = w == (((p.dbg >> 8) & 15) ? ((p.dbg >> 8) & 15) - 1 : p.NW / 2) && s == p.shot0;   // dbg bits 8-11: traced slab + 1
#endif

    for (int it = 0; it < nsteps; ++it) {
        const int n = adj ? (p.n_first - it) : (p.n_first + it);
        const float amp = amp_next;
        CL_LAGGARD();
        CL_STAMP(0);
        // ---- sampling of the current field (owner slab writes) -------------------------------
        if (kDbg(p) & 8) {
        } else if (!slow_sparse) {
            // uniform row base + a 32-bit lane offset: no per-lane 64-bit address to keep (or spill)
            float *out_n = adj ? p.grad_f + ((long long)n * p.nshot + s) * p.nsrc
                               : p.rec_out + ((long long)n * p.nshot + s) * p.nrec;
            if (smp_off >= 0) out_n[cl_opaque(t)] = fmaf(adj ? src_wt : smp_w, cur[cl_opaque(smp_off)], 0.f);
            else if (smp_off == -2) out_n[cl_opaque(t)] = 0.f;
        } else if (!adj) {
            if (p.rec_out != nullptr)
                for (int e = t; e < p.nrec; e += kClThreads) {
                    const int cell = p.rec_cell[(long long)s * p.nrec + e];
                    if (cell < 0) {
                        if (w == 0) p.rec_out[((long long)n * p.nshot + s) * p.nrec + e] = 0.f;
                        continue;
                    }
                    const int i0 = cell / p.n1, i1 = cell - i0 * p.n1;
                    if (i0 >= r0 && i0 < r0 + R)
                        p.rec_out[((long long)n * p.nshot + s) * p.nrec + e] =
                            fmaf(p.rec_w[(long long)s * p.nrec + e], cur[(i0 - r0 + 2) * PL + 4 + i1], 0.f);
                }
        } else if (p.grad_f != nullptr) {
            for (int e = t; e < p.nsrc; e += kClThreads) {
                const int cell = p.src_cell[(long long)s * p.nsrc + e];
                if (cell < 0) {
                    if (w == 0) p.grad_f[((long long)n * p.nshot + s) * p.nsrc + e] = 0.f;
                    continue;
                }
                const int i0 = cell / p.n1, i1 = cell - i0 * p.n1;
                if (i0 >= r0 && i0 < r0 + R) {
                    const float q = p.q0[i0] + p.q1[i1];
                    const float wq = p.src_w[(long long)s * p.nsrc + e] * (1.0f + q * p.r[(long long)i0 * p.gp + i1]);
                    p.grad_f[((long long)n * p.nshot + s) * p.nsrc + e] = fmaf(wq, cur[(i0 - r0 + 2) * PL + 4 + i1], 0.f);
                }
            }
        }
        CL_STAMP(1);
        if (PML) {
            const float *ul = cur + (2 - r0) * PL + 4;             // ul[i0 * PL + i1] = the current field at grid cell (i0, i1)
            // edge slabs in the own-group form: axis 0 by the owner of each group (phase 0 / 1; e0 inside the update)
            auto own_phase = [&](int ph) {
#pragma unroll
                for (int i = 0; i < kClMaxNG; ++i)
                    if (i < nown) {
                        const int lo_i = cl_opaque(loff[i]), jg_i = cl_opaque(jg[i]);
                        const int j = jg_i >> 12, g = jg_i & 4095;
                        if (!adj) pml_own_fwd_psi(m, w == 0, j - r0, j, cur + lo_i, plP + lo_i, PL);
                        else if (ph == 0) pml_own_adj_a(m, s, w == 0, j - r0, j, g, cur + lo_i, plP + lo_i);
                        else pml_own_adj_b(m, s, w == 0, j - r0, j, g, cur + lo_i, plP + lo_i, plQ + lo_i, PL);
                    }
            };
            if (!adj) {
                if (own) own_phase(0);
                cl_pml_cells(m, w, p.NW, r0, R, t, own, [&](const PmlCell &c, int ax) { ac_pml_fwd_psi_cell(m, s, ax, c, ul, PL); });
                CL_STAMP(11);
                cl_pml_sync();
                CL_STAMP(12);
                cl_pml_cells(m, w, p.NW, r0, R, t, own, [&](const PmlCell &c, int ax) { ac_pml_fwd_zeta_cell(m, s, ax, c, ul, PL); });
                CL_STAMP(13);
            } else {
                if (own) own_phase(0);
                cl_pml_cells(m, w, p.NW, r0, R, t, own, [&](const PmlCell &c, int ax) { ac_pml_adj_a_cell(m, s, ax, c, ul, PL); });
                CL_STAMP(11);
                cl_pml_sync();
                CL_STAMP(12);
                if (own) own_phase(1);
                cl_pml_cells(m, w, p.NW, r0, R, t, own, [&](const PmlCell &c, int ax) { ac_pml_adj_b_cell(m, s, ax, c, ul, PL); });
                CL_STAMP(13);
                cl_pml_sync();
                CL_STAMP(15);
                cl_pml_cells(m, w, p.NW, r0, R, t, own, [&](const PmlCell &c, int ax) { ac_pml_adj_c_cell(m, s, ax, c, ul, PL); });
            }
            // the last phase's e is read by the update: of every group on an edge slab in the generic form (barrier), of the
            // groups at the ends of the rows otherwise - in slot 0, behind barrier A, when late_e
            if (!(late_e && (own || !(w == 0 || w == p.NW - 1)))) cl_pml_sync();
            CL_STAMP(14);
        }
        // ---- stencil: new field overwrites prv in place (prv is only read at the own cell) --
        float *Gn = (MODE == 1) ? p.G + (long long)(n - p.g_first) * p.g_step + plane : nullptr;
        const unsigned epoch = (unsigned)(it + 1);
        unsigned long long *xmine = xmine0 + (epoch & 1u) * 4 * p.gp;
        // One slot = one 4-cell group of this thread.  Slot 0 holds the groups of the slab's four boundary rows (the
        // only ones whose stencil reaches the halo rows), slots 1.. the interior.
        auto update_slot = [&](auto I) {
            constexpr int i = decltype(I)::value;
            if (i < nown && !(kDbg(p) & 4)) {
                float un[4], gk[4];
                float4 q1 = make_float4(0.f, 0.f, 0.f, 0.f);
                float q0 = 0.f;
                const bool damped = !PML && ((dampmask >> i) & 1u);
                const int lo_i = cl_opaque(loff[i]), jg_i = cl_opaque(jg[i]);
                if (damped) {
                    q1 = *reinterpret_cast<const float4 *>(ldq1 + 4 * (jg_i & 4095));
                    q0 = ldq0[(jg_i >> 12) - r0];
                }
                float4 pe = make_float4(0.f, 0.f, 0.f, 0.f);
                // the layer's term of the group's four cells (an exact zero for the groups away from the layer: not read)
                if (PML && ((dampmask >> i) & 1u)) {
                    if (own) {
                        const int j = jg_i >> 12;
                        const float4 e0 = adj ? pml_own_adj_e0(plP + lo_i, plQ + lo_i, PL)
                                              : pml_own_fwd_e0(m, s, w == 0, j - r0, j, jg_i & 4095, cur + lo_i, plP + lo_i, PL);
                        pe = pml_term(m, s, j, jg_i & 4095, adj, (dampmask >> (4 + i)) & 1u, &e0);
                    } else {
                        pe = pml_term(m, s, jg_i >> 12, jg_i & 4095, adj, (dampmask >> (4 + i)) & 1u);
                    }
                }
                cl_update<MODE == 1, PML>(cur + lo_i, prv + lo_i, PL, rr[i], q0, q1, damped, p.c0, p.c1,
                                          p.n1 - 4 * (jg_i & 4095), un, gk, pe);
                if (!adj && !slow_sparse && i == src_slot) {
                    const float a = src_wt * amp;
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc)
                        if (cc == src_comp) { un[cc] += a * comp(rr[i], cc); if (MODE == 1) gk[cc] += a; }
                }
                if (!adj && slow_sparse && p.nsrc > 0) {
                    for (int e = 0; e < p.nsrc; ++e) {
                        const int cell = p.src_cell[(long long)s * p.nsrc + e];
                        if (cell < 0) continue;
                        const int i0 = cell / p.n1, i1 = cell - i0 * p.n1;
                        if ((long long)i0 * p.gp + (i1 & ~3) == goff(i)) {
                            const float a = p.src_w[(long long)s * p.nsrc + e] *
                                            p.f[((long long)n * p.nshot + s) * p.nsrc + e];
#pragma unroll
                            for (int cc = 0; cc < 4; ++cc)
                                if (cc == (i1 & 3)) { un[cc] += a * comp(rr[i], cc); if (MODE == 1) gk[cc] += a; }
                        }
                    }
                }
                if (MODE == 3) {
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) un[cc] = fmaf(comp(acc[i], cc), comp(Gbuf[i], cc), un[cc]);
                }
                *reinterpret_cast<float4 *>(prv + lo_i) = make_float4(un[0], un[1], un[2], un[3]);
                if (MODE == 1) Gbuf[i] = make_float4(gk[0], gk[1], gk[2], gk[3]);
            }
            __builtin_amdgcn_sched_barrier(0);   // one group at a time: keeps the register peak below 128
        };
        // Interior first: it needs no halo, and the boundary rows the neighbours published at the end of the last step
        // travel meanwhile.  Then the poll, the boundary rows, and the publish of the new boundary rows - whose flight the
        // interior of the NEXT step hides.  (Round 1 updated and published the boundary rows first and polled at the end
        // of the step: same overlap, one barrier more; the adjoint, which can only publish after the receivers have
        // been injected, had no overlap at all.)
        update_slot(std::integral_constant<int, 1>());
        update_slot(std::integral_constant<int, 2>());
        update_slot(std::integral_constant<int, 3>());
        static_assert(kClMaxNG == 4, "slots 1..3 are spelled out");
        CL_STAMP(2);
        // ---- receive the neighbours' boundary rows of the CURRENT field (published as epoch `it`) into the halo rows
        const unsigned long long *xup_in = xup0 + ((unsigned)it & 1u) * 4 * p.gp;
        const unsigned long long *xdn_in = xdn0 + ((unsigned)it & 1u) * 4 * p.gp;
        if (do_x && it > 0) {
            // sweep: each pass re-reads ALL of this thread's granules back to back (one memory round
            // trip per pass, not one per granule) until every tag carries the epoch
            unsigned long long v[kGr];
            const unsigned long long *src[kGr];
#pragma unroll
            for (int kk = 0; kk < kGr; ++kk) {
                // our top halo = the upper neighbour's "down" rows, our bottom halo = the lower one's "up" rows
                src[kk] = (((upmask >> kk) & 1u) ? xup_in : xdn_in) + src_idx(kk);
                v[kk] = 0;
            }
            for (unsigned spins = 0;; ++spins) {
                bool ok = true;
#pragma unroll
                for (int kk = 0; kk < kGr; ++kk)
                    if (rcv_lo[kk] >= 0)
                        v[kk] = __hip_atomic_load(src[kk], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int kk = 0; kk < kGr; ++kk)
                    if (rcv_lo[kk] >= 0) ok = ok && (unsigned)(v[kk] >> 32) == (unsigned)it;
                // a wave that waits this long for a neighbour says so at once (rare path, no state carried through the loop)
                if (spins == mifwi::kSlowPollPasses && (t & 63) == 0) atomicAdd(p.err + mifwi::kErrSlow, 1);
                if (ok || failed) break;            // once failed: one pass per step, garbage forward until the check
                if (spins > kClMaxSpin ||
                    ((spins & 255u) == 255u && __hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                    // publish at once: every other workgroup of the launch bails within 256 spins instead of
                    // running into its own time-out, one hand-off after the other
                    if (spins > kClMaxSpin) __hip_atomic_store(p.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    failed = true;
                    break;
                }
                mifwi::poll_nap(p.nap);
            }
#pragma unroll
            for (int kk = 0; kk < kGr; ++kk)
                if (rcv_lo[kk] >= 0) cur[rcv_lo[kk]] = __uint_as_float((unsigned)v[kk]);
            // the sweep has landed, so (vector memory completes in order) nothing this wave issued before it is in
            // flight: stating vmcnt(0) is free and stops the compiler from guarding the reuse of the sweep registers
            // with waits that would later stall on the snapshot stores / prefetches issued below
            __builtin_amdgcn_s_waitcnt(0x0f70);
        }
        CL_STAMP(3);
        __syncthreads();                     // A: halo rows of the current field are in LDS
        CL_STAMP(4);
        update_slot(std::integral_constant<int, 0>());
        CL_STAMP(5);
        __syncthreads();                     // B: the new field is complete on the own rows; every read of the old one is done
        CL_STAMP(6);
        if (adj) {
            // ---- adjoint sources: receiver taps of this slab, z^k[cell] += (w g) (r inv) -------
            if (!slow_sparse) {
                if (inj_pack >= 0) {
                    float *cell = &prv[inj_off()];
                    if (tap_rounds == 0) atomicAdd(cell, (smp_w * amp) * inj_scale);
                    else if (inj_turn() == 0) *cell = *cell + (smp_w * amp) * inj_scale;
                }
                for (int r = 1; r < tap_rounds; ++r) {       // taps that share a cell: one per turn, in a fixed order
                    __syncthreads();
                    if (inj_pack >= 0 && inj_turn() == r) {
                        float *cell = &prv[inj_off()];
                        *cell = *cell + (smp_w * amp) * inj_scale;
                    }
                }
            } else {
                const int cnt = p.slab_cnt[s * p.NW + w];
                const int *lst = p.slab_list + ((long long)s * p.NW + w) * p.nrec;
                for (int e = t; e < cnt; e += kClThreads) {
                    const int id = lst[e];
                    const int cell = p.rec_cell[(long long)s * p.nrec + id];
                    const int i0 = cell / p.n1, i1 = cell - i0 * p.n1;
                    const float rv = p.r[(long long)i0 * p.gp + i1];
                    const float q = p.q0[i0] + p.q1[i1];
                    const float inv = 1.0f / (1.0f + q * rv);
                    const float a = p.rec_w[(long long)s * p.nrec + id] *
                                    p.grad_rec[((long long)n * p.nshot + s) * p.nrec + id];
                    atomicAdd(&prv[(i0 - r0 + 2) * PL + 4 + i1], a * (rv * inv));
                }
            }
            __syncthreads();
            // ---- imaging with the injected field: acc += z^k * G^{k-1} ----------------------
#pragma unroll
            for (int i = 0; i < kClMaxNG; ++i) {
                if (i < nown) {
                    const float4 z = *reinterpret_cast<const float4 *>(prv + cl_opaque(loff[i]));
                    acc[i].x = fmaf(z.x, Gbuf[i].x, acc[i].x); acc[i].y = fmaf(z.y, Gbuf[i].y, acc[i].y);
                    acc[i].z = fmaf(z.z, Gbuf[i].z, acc[i].z); acc[i].w = fmaf(z.w, Gbuf[i].w, acc[i].w);
                }
            }
        }
        CL_STAMP(7);
        if (do_x) {
#pragma unroll
            for (int kk = 0; kk < kGr; ++kk)
                if (rcv_lo[kk] >= 0)
                    __hip_atomic_store(xmine + t + kk * kClThreads,
                                       ((unsigned long long)epoch << 32) | __float_as_uint(prv[pub_off(kk)]),
                                       __ATOMIC_RELAXED, ClScope<AG>::value);
        }
        CL_STAMP(8);
        // global traffic that nobody waits for goes AFTER the hand-off (vector memory operations
        // retire in order: a poll issued behind these would wait for them)
        if (MODE == 1 && !(kDbg(p) & 2)) {
#pragma unroll
            for (int i = 0; i < kClMaxNG; ++i)
                if (i < nown) mifwi::stnt4(Gn + goff_of(cl_opaque(jg[i])), Gbuf[i]);
        }
        prefetch(it + 1);
        CL_STAMP(9);
        // a timed-out thread carries garbage forward until the next collective check (fatal anyway)
        // No barrier at the end of a step: the next step reads the field just written (complete since B, injected since
        // the barrier behind the injection) and writes the buffer whose last readers finished before B.
        if ((it & 31) == 31 || it == nsteps - 1) {
            if (__syncthreads_or(failed ? 1 : 0)) {
                if (t == 0) __hip_atomic_store(p.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                failed = true;
                break;
            }
        }
        CL_STAMP(10);
        float *tmp = cur; cur = prv; prv = tmp;
    }

    if (PML) {                               // memory variables kept in LDS go back to the global state
        __syncthreads();
        const long long W = m.W;
        const bool edge = w == 0 || w == p.NW - 1;
        auto back = [&](int bit, float *dst, const float *src, long long first, long long n, long long shot_stride) {
            if (m.lds & (unsigned)bit)
                for (long long e = t; e < n; e += kClThreads) dst[(long long)s * shot_stride + first + e] = src[e];
        };
        back(PML_A1, p.pml.A1, m.A1, m.f1s, 2 * W * R, m.s1); back(PML_B1, p.pml.B1, m.B1, m.f1s, 2 * W * R, m.s1);
        if (edge) { back(PML_A0, p.pml.A0, m.A0, m.f0s, W * m.gp, m.s0); back(PML_B0, p.pml.B0, m.B0, m.f0s, W * m.gp, m.s0); }
        if (own && !adj) {                   // Psi: the strip's rows of the plane
            float *dst = p.pml.A0 + (long long)s * m.s0 + m.f0s;
            for (int e = t; e < (int)(W * m.gp); e += kClThreads) {
                const int row = e / m.gp, col = e - row * m.gp;
                dst[e] = plP[(row + (w == 0 ? 0 : 2) + 2) * PL + 4 + col];
            }
        }
    }
    // ---- write the state (own rows of both levels) and the accumulators back --------------------
    const int parE = adj ? ((p.nt - 1 - (p.n_last - 1)) & 1) : (p.n_last & 1);
    float *ocur = (parE ? p.ub : p.ua) + (long long)s * p.shot_stride;
    float *oprev = (parE ? p.ua : p.ub) + (long long)s * p.shot_stride;
#pragma unroll
    for (int i = 0; i < kClMaxNG; ++i) {
        if (i < nown) {
            const long long o = (long long)((jg[i] >> 12) + 2) * p.pitch + 4 + 4 * (jg[i] & 4095);
            *reinterpret_cast<float4 *>(ocur + o) = *reinterpret_cast<const float4 *>(cur + loff[i]);
            *reinterpret_cast<float4 *>(oprev + o) = *reinterpret_cast<const float4 *>(prv + loff[i]);
            if (adj) *reinterpret_cast<float4 *>(p.acc + plane + goff(i)) = acc[i];
        }
    }
}

}  // namespace

// ================================================================================================
struct mifwi_acoustic_plan {
    mifwi_acoustic_desc d;
    int device;
    int ng, gp, pitch, lx, rz, gs, ngroups, xcd;
    int pass_fwd, pass_adj;                // shot groups per pass over the time range (per-step family)
    long long shot_stride, field_elems, coef_elems;
    // cluster path (LDS-resident time loop), 0 when the shot does not fit
    int cluster, NW, PL, cl_shots, cl_lds, rt;
    long long xbuf_elems, list_elems;      // in floats
    // second-order C-PML (desc.cpml_width > 0; one launch per step family only)
    int pmlW;
    long long pml_persist, pml_scratch, zq_elems;      // floats, all shots: memory variables | scratch + e | zero q0, q1
};

namespace {

template <int LX, bool SAVE, bool IMAGE>
void launch_rz(const mifwi_acoustic_plan *pl, dim3 grid, const AcParams &q, hipStream_t st)
{
    dim3 block(kThreads);
    if (q.pmlW > 0) {                  // C-PML plans march two rows per thread (plan_create pins rz)
        hipLaunchKernelGGL((ac_step<LX, 2, SAVE, IMAGE, true>), grid, block, 0, st, q);
        return;
    }
    switch (pl->rz) {
        case 8: hipLaunchKernelGGL((ac_step<LX, 8, SAVE, IMAGE>), grid, block, 0, st, q); break;
        case 2: hipLaunchKernelGGL((ac_step<LX, 2, SAVE, IMAGE>), grid, block, 0, st, q); break;
        case 1: hipLaunchKernelGGL((ac_step<LX, 1, SAVE, IMAGE>), grid, block, 0, st, q); break;
        default: hipLaunchKernelGGL((ac_step<LX, 4, SAVE, IMAGE>), grid, block, 0, st, q); break;
    }
}

template <bool SAVE, bool IMAGE>
void launch_step(const mifwi_acoustic_plan *pl, const AcParams &p, hipStream_t st, int ngroups = -1)
{
    const int lz = kThreads / pl->lx;
    const int tiles_x = mifwi::ceil_div(pl->ng, pl->lx);
    const int tiles_z = mifwi::ceil_div(pl->d.n0, lz * pl->rz);
    AcParams q = p;
    q.tiles_z = tiles_z;
    int extra = 0;
    if (p.smp_out != nullptr && p.nsmp > 0)
        extra = mifwi::ceil_div(mifwi::ceil_div(pl->gs * p.nsmp, kThreads), tiles_x);
    dim3 grid(tiles_x, tiles_z + extra, ngroups > 0 ? ngroups : pl->ngroups);
    switch (pl->lx) {
        case 64: launch_rz<64, SAVE, IMAGE>(pl, grid, q, st); break;
        case 32: launch_rz<32, SAVE, IMAGE>(pl, grid, q, st); break;
        default: launch_rz<16, SAVE, IMAGE>(pl, grid, q, st); break;
    }
}

int env_int(const char *name, int dflt)
{
    const char *v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

AcParams base_params(const mifwi_acoustic_plan *pl, const float *r, const float *q0,
                     const float *q1)
{
    AcParams p;
    memset(&p, 0, sizeof(p));
    p.n0 = pl->d.n0; p.n1 = pl->d.n1; p.ng = pl->ng; p.pitch = pl->pitch; p.gp = pl->gp;
    p.shot_stride = pl->shot_stride; p.nshot = pl->d.nshot; p.gs = pl->gs;
    p.c0 = pl->d.c0; p.c1 = pl->d.c1; p.r = r; p.q0 = q0; p.q1 = q1;
    p.xcd = pl->xcd;
    return p;
}

// ---- C-PML helpers -------------------------------------------------------------------------------------
// persist: [A0 | B0 | A1 | B1], scratch: [P0 | Q0 | P1 | Q1 | e0 | e1], every array shot-major
AcPml pml_params(const mifwi_acoustic_plan *pl, const float *ab0, const float *ab1, float *persist, float *scratch)
{
    AcPml m;
    memset(&m, 0, sizeof(m));
    const mifwi_acoustic_desc &d = pl->d;
    const long long ns = d.nshot;
    m.W = pl->pmlW; m.n0 = d.n0; m.n1 = d.n1; m.gp = pl->gp; m.pitch = pl->pitch; m.shot_stride = pl->shot_stride;
    m.ab0 = ab0; m.ab1 = ab1; m.c0 = d.c0; m.c1 = d.c1; m.nshot = d.nshot;
    m.s0 = 2LL * m.W * pl->gp; m.s1 = 2LL * m.W * d.n0;
    m.r0 = 2LL * (m.W + 2) * pl->gp; m.r1 = 2LL * (m.W + 2) * d.n0;
    m.A0 = persist; m.B0 = m.A0 + ns * m.s0; m.A1 = m.B0 + ns * m.s0; m.B1 = m.A1 + ns * m.s1;
    m.P0 = scratch; m.Q0 = m.P0 + ns * m.s0; m.P1 = m.Q0 + ns * m.s0; m.Q1 = m.P1 + ns * m.s1;
    m.e0 = m.Q1 + ns * m.s1; m.e1 = m.e0 + ns * m.r0;
    m.mg_ng = pml_magic((unsigned)pl->gp / 4u); m.mg_w2 = pml_magic(2u * (unsigned)(m.W + 2));
    return m;
}
// the thin launches of one step for shots [shot0, shot0 + count): forward (Psi, Z, e) or adjoint (P/Zb, Q/Pb, e)
void pml_step(const AcPml &m0, const float *cur, int shot0, int count, bool adjoint, hipStream_t st)
{
    AcPml m = m0;
    m.shot0 = shot0;
    const long long cells = 2LL * (m.W + 2) * std::max(m.gp / 4, m.n0);      // threads per axis (blockIdx.z): groups of four cells on axis 0
    const dim3 grid((unsigned)((cells + kThreads - 1) / kThreads), (unsigned)count, 2), block(kThreads);
    if (!adjoint) {
        hipLaunchKernelGGL(ac_pml_fwd_psi, grid, block, 0, st, m, cur);
        hipLaunchKernelGGL(ac_pml_fwd_zeta, grid, block, 0, st, m, cur);
    } else {
        hipLaunchKernelGGL(ac_pml_adj_a, grid, block, 0, st, m, cur);
        hipLaunchKernelGGL(ac_pml_adj_b, grid, block, 0, st, m, cur);
        hipLaunchKernelGGL(ac_pml_adj_c, grid, block, 0, st, m, cur);
    }
}

// ---- cluster path helpers --------------------------------------------------------------------------
void cluster_setup(mifwi_acoustic_plan *pl)
{
    pl->cluster = 0; pl->NW = 0; pl->PL = 4 * (pl->ng + 2); pl->cl_shots = 0; pl->cl_lds = 0; pl->rt = 0;
    pl->xbuf_elems = 0; pl->list_elems = 0;
    if (env_int("MIFWI_AC_CLUSTER", 1) == 0 || pl->d.ntap != 1) return;
    if (pl->pmlW > 0 && env_int("MIFWI_AC_CLUSTER_PML", 1) == 0) return;
    int ncu = 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, pl->device) != hipSuccess) return;
    const int forced = env_int("MIFWI_AC_NW", 0);
    // C-PML plans: the first and the last slab hold exactly the layer's rows and the two beyond them (W + 2), so that
    // everything the layer's phases read and write along z stays inside one slab
    const int hint = pl->pmlW > 0 ? pl->pmlW + 2 : env_int("MIFWI_AC_EDGE_ROWS", pl->d.edge_rows);
    // Slab count: a step costs a fixed part (barriers, one hand-off flight) plus the groups a thread
    // updates, times the launches the shot batch needs (measured: ~(4.7 + groups/thread) per launch on
    // C1/C2).  Few shots -> many thin slabs use the idle CUs; a full batch -> the fewest slabs that fit.
    double best = 1e30;
    for (int nw = 1; nw <= 32; ++nw) {
        if (forced > 0 && nw != forced) continue;
        if (pl->d.n0 / nw < 4) break;
        if (forced <= 0 && nw > 1 && pl->d.n0 / nw < 6) break;      // thinner slabs are all boundary rows
        const int per_launch = 8 * (ncu / (8 * nw));       // shots per launch (multiple of 8)
        if (per_launch < 8) break;
        // both splits are priced: the absorbing layer in slabs of its own (hint = its width), or even
        for (int uneven = 1; uneven >= (pl->pmlW > 0 ? 1 : 0); --uneven) {
            int rt = 0, rows = mifwi::ceil_div(pl->d.n0, nw);
            if (uneven) {
                if (hint < 4 || nw < 3 || pl->d.n0 - 2 * hint < 4 * (nw - 2)) continue;
                rt = hint;
                rows = std::max(rt, mifwi::ceil_div(pl->d.n0 - 2 * rt, nw - 2));
            }
            long long lds = (2LL * (rows + 4) * pl->PL + pl->gp + rows + 8) * sizeof(float);
            if (lds > 150 * 1024) continue;
            if (pl->pmlW > 0) {
                // the layer's arrays each slab class can keep in LDS (pml_place, same call as in the kernel): the launch asks
                // for the largest need; nothing placed still works (everything through global memory)
                const int rows_int = mifwi::ceil_div(pl->d.n0 - 2 * rt, nw - 2);
                const long long cap = kClPmlLdsLimit / (long long)sizeof(float);
                long long need = lds / (long long)sizeof(float);
                for (int cls = 0; cls < 2; ++cls)
                    for (int adjm = 0; adjm < 2; ++adjm) {
                        const int rws = cls == 0 ? rt : rows_int;
                        long long base = 2LL * (rws + 4) * pl->PL + pl->gp + ((rws + 3) & ~3);
                        long long used = 0;
                        // (the kernel's own decision, from the same numbers: ac_cluster, `own`)
                        const bool own = cls == 0 && env_int("MIFWI_AC_PML_OWN", 1) != 0 &&
                                         base + pml_own_floats(adjm != 0, rws, pl->PL) <= cap;
                        if (own) base += pml_own_floats(adjm != 0, rws, pl->PL);
                        pml_place(adjm != 0, cls == 0, rws, pl->pmlW, pl->gp, cap - base, &used, own);
                        need = std::max(need, base + used);
                    }
                lds = std::min<long long>(need, cap) * (long long)sizeof(float);
            }
            if ((long long)rows * pl->ng > (long long)kClMaxNG * kClThreads || pl->ng > 4095 ||
                4 * pl->gp > 3 * kClThreads) continue;
            // rows of the slowest slab, a sponge row counted 1.5 times (a division per cell and step)
            const double units = uneven ? std::max(1.5 * rt, (double)mifwi::ceil_div(pl->d.n0 - 2 * rt, nw - 2))
                                        : rows + 0.5 * std::min(std::max(hint, 0), rows);
            const double cost = mifwi::ceil_div(pl->d.nshot, per_launch) * (4.7 + units * pl->ng / kClThreads);
            if (cost < best - 1e-9) {
                best = cost;
                pl->cluster = 1; pl->NW = nw; pl->cl_shots = per_launch; pl->cl_lds = (int)lds; pl->rt = rt;
            }
        }
    }
    if (!pl->cluster) return;
    // granules, the XCC_ID table of mifwi::same_xcd ([nshot][NW] ints), the block of the error word
    pl->xbuf_elems = mifwi::round_up64(2LL * pl->d.nshot * pl->NW * 8 * pl->gp, 64) +
                     mifwi::round_up64((long long)pl->d.nshot * pl->NW, 64) + 64;
    pl->list_elems = mifwi::round_up64((long long)pl->d.nshot * pl->NW * (1 + (long long)pl->d.nrec), 64);
    // 28 function attributes: once per device and process, not once per plan (a plan is created on every propagate call)
    static std::atomic<unsigned> attr_done{0};
    if (pl->device < 32 && (attr_done.load() >> pl->device & 1u)) return;
    for (const void *fn : {(const void *)ac_cluster<0, false, false>, (const void *)ac_cluster<1, false, false>,
                           (const void *)ac_cluster<2, false, false>, (const void *)ac_cluster<3, false, false>,
                           (const void *)ac_cluster<0, true, false>, (const void *)ac_cluster<1, true, false>,
                           (const void *)ac_cluster<2, true, false>, (const void *)ac_cluster<3, true, false>,
                           (const void *)ac_cluster<0, false, true>, (const void *)ac_cluster<1, false, true>,
                           (const void *)ac_cluster<2, false, true>, (const void *)ac_cluster<3, false, true>,
                           (const void *)ac_cluster<0, true, true>, (const void *)ac_cluster<1, true, true>,
                           (const void *)ac_cluster<2, true, true>, (const void *)ac_cluster<3, true, true>,
                           (const void *)ac_cluster<0, false, false, true>, (const void *)ac_cluster<1, false, false, true>,
                           (const void *)ac_cluster<2, false, false, true>, (const void *)ac_cluster<0, true, false, true>,
                           (const void *)ac_cluster<1, true, false, true>, (const void *)ac_cluster<2, true, false, true>,
                           (const void *)ac_cluster<0, false, true, true>, (const void *)ac_cluster<1, false, true, true>,
                           (const void *)ac_cluster<2, false, true, true>, (const void *)ac_cluster<0, true, true, true>,
                           (const void *)ac_cluster<1, true, true, true>, (const void *)ac_cluster<2, true, true, true>})
        // one cap for every variant (the attribute is per function and plans of both kinds coexist): C-PML plans ask for up
        // to kClPmlLdsLimit, the others never for more than kClusterLdsLimit
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kClPmlLdsLimit) != hipSuccess) {
            (void)hipGetLastError();           // not sticky: the plan simply uses one launch per step
            pl->cluster = 0;
            return;
        }
    if (pl->device < 32) attr_done.fetch_or(1u << pl->device);
}

ClParams cluster_params(const mifwi_acoustic_plan *pl, const float *r, const float *q0, const float *q1,
                        float *ua, float *ub, float *xbuf)
{
    ClParams c;
    memset(&c, 0, sizeof(c));
    c.n0 = pl->d.n0; c.n1 = pl->d.n1; c.ng = pl->ng; c.gp = pl->gp; c.pitch = pl->pitch;
    c.shot_stride = pl->shot_stride; c.nshot = pl->d.nshot; c.NW = pl->NW; c.PL = pl->PL; c.rt = pl->rt;
    c.nt = pl->d.nt; c.c0 = pl->d.c0; c.c1 = pl->d.c1; c.r = r; c.q0 = q0; c.q1 = q1;
    c.ua = ua; c.ub = ub;
    c.nsrc = pl->d.nsrc; c.ntap = 1; c.nrec = pl->d.nrec;
    c.xbuf = reinterpret_cast<unsigned long long *>(xbuf);
    c.err = reinterpret_cast<int *>(xbuf + pl->xbuf_elems - 64);
    c.xcc_tab = reinterpret_cast<int *>(xbuf + pl->xbuf_elems - 64 - mifwi::round_up64((long long)pl->d.nshot * pl->NW, 64));
    c.dbg = env_int("MIFWI_AC_CL_DBG", 0);
    c.rcv_plain = env_int("MIFWI_AC_ADJ_PLAIN", 1);
    // fat slabs nap long between poll passes, thin ones short (mifwi::poll_nap)
    c.nap = env_int("MIFWI_POLL_NAP", mifwi::ceil_div(pl->d.n0, pl->NW) >= 16 ? 48 : 1);
    c.pml_lds_floats = env_int("MIFWI_AC_PML_LDS", 1) ? pl->cl_lds / (int)sizeof(float) : 0;      // 0: every layer array through global memory
    // tests: pretend the launch has this many KB less LDS, so that pml_place keeps only a prefix of its list in LDS (every
    // partial placement must give the same bits)
    c.pml_lds_floats = std::max(0, c.pml_lds_floats - 256 * env_int("MIFWI_AC_PML_LDS_SHRINK_KB", 0));
    c.pml_own = env_int("MIFWI_AC_PML_OWN", 1);
    c.pml_late_e = env_int("MIFWI_AC_PML_LATE_E", 1);
    return c;
}

template <int MODE, bool AG>
int cluster_attempt(const mifwi_acoustic_plan *pl, ClParams c, float *xbuf, hipStream_t st)
{
    if (mifwi::fake_timeout() == 1) return mifwi::kClusterTimedOut;
    MIFWI_HIP_TRY(hipMemsetAsync(xbuf, 0, sizeof(float) * pl->xbuf_elems, st));
#ifdef MIFWI_ABLATIONS
    // MIFWI_AC_CL_TRACE=<file>: phase time stamps (CL_STAMP) of one workgroup, steps 64..127, appended as text
    const char *trace_path = getenv("MIFWI_AC_CL_TRACE");
    const size_t trace_n = 64 * 16 * 16;
    if (trace_path && *trace_path) {
        MIFWI_HIP_TRY(hipMalloc(&c.trace, trace_n * sizeof(long long)));
        MIFWI_HIP_TRY(hipMemsetAsync(c.trace, 0, trace_n * sizeof(long long), st));
    }
#endif
    for (int s0 = 0; s0 < pl->d.nshot; s0 += pl->cl_shots) {
        c.shot0 = s0;
        c.shot1 = std::min(pl->d.nshot, s0 + pl->cl_shots);
        const int nsl8 = mifwi::ceil_div(c.shot1 - s0, 8);
        // general (rescanning) variant only when a thread may own several sparse points
        const bool general = MODE == 2 ? (c.nrec > kClThreads || c.nsrc > kClThreads)
                                       : (c.nsrc > 1 || c.nrec > kClThreads);
        const dim3 grid(8 * pl->NW * nsl8), block(kClThreads);
        if constexpr (MODE != 3) {
            if (pl->pmlW > 0) {
                if (general) hipLaunchKernelGGL((ac_cluster<MODE, true, AG, true>), grid, block, pl->cl_lds, st, c);
                else hipLaunchKernelGGL((ac_cluster<MODE, false, AG, true>), grid, block, pl->cl_lds, st, c);
                continue;
            }
        }
        if (general) hipLaunchKernelGGL((ac_cluster<MODE, true, AG>), grid, block, pl->cl_lds, st, c);
        else hipLaunchKernelGGL((ac_cluster<MODE, false, AG>), grid, block, pl->cl_lds, st, c);
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
            fprintf(fp, "# ac_cluster mode=%d waves=16\n", MODE);
            for (size_t i = 0; i < trace_n; i += 16) {
                for (int k = 0; k < 16; ++k) fprintf(fp, "%lld ", h[i + k]);
                fprintf(fp, "\n");
            }
            fclose(fp);
        }
    }
#endif
    const int verdict = mifwi::cluster_verdict(err, "acoustic");
    return mifwi::fake_timeout() == 2 ? mifwi::kClusterTimedOut : verdict;
}

int cluster_restore(float *work, long long state_elems, const float *backup, int32_t flags, hipStream_t st);

// The single-launch time loop with its middle tier: a failed placement check (the slabs of a shot were not dealt to one
// XCD) restores the state and repeats the launch with granules published through the fabric; what comes back is OK,
// an error, or kClusterTimedOut (the caller then restores the state and runs one launch per step).
template <int MODE>
int cluster_run(const mifwi_acoustic_plan *pl, ClParams c, float *xbuf, hipStream_t st, float *work, long long state_elems,
                const float *backup, int32_t flags)
{
    int rc = cluster_attempt<MODE, false>(pl, c, xbuf, st);
    if (rc != mifwi::kClusterMisplaced) return rc;
    mifwi::note_agent_tier("acoustic");
    rc = cluster_restore(work, state_elems, backup, flags, st);
    if (rc) return rc;
    rc = cluster_attempt<MODE, true>(pl, c, xbuf, st);
    return rc == mifwi::kClusterMisplaced ? mifwi::kClusterTimedOut : rc;
}

// A single-launch attempt may time out (some workgroup was not resident in time) after it has advanced the state by
// an unknown number of steps.  A call that starts from the zero state is simply zeroed again; a resumed call (time
// checkpointing) keeps a copy of its input state behind the work buffer's other regions and gets it back.  Either
// way the per-step family then runs the range.
int cluster_backup(float *work, long long state_elems, float *backup, int32_t flags, hipStream_t st)
{
    if (flags & MIFWI_ZERO_STATE) return MIFWI_OK;
    MIFWI_HIP_TRY(hipMemcpyAsync(backup, work, sizeof(float) * state_elems, hipMemcpyDeviceToDevice, st));
    return MIFWI_OK;
}
int cluster_restore(float *work, long long state_elems, const float *backup, int32_t flags, hipStream_t st)
{
    if (flags & MIFWI_ZERO_STATE) MIFWI_HIP_TRY(hipMemsetAsync(work, 0, sizeof(float) * state_elems, st));
    else MIFWI_HIP_TRY(hipMemcpyAsync(work, backup, sizeof(float) * state_elems, hipMemcpyDeviceToDevice, st));
    return MIFWI_OK;
}

}  // namespace

extern "C" {

const char *mifwi_last_error(void) { return mifwi::err_buf(); }
int mifwi_version(void) { return MIFWI_VERSION_MAJOR * 1000 + MIFWI_VERSION_MINOR; }
int64_t mifwi_fallback_count(void) { return (int64_t)mifwi::g_fallbacks.load(std::memory_order_relaxed); }
int64_t mifwi_agent_handoff_count(void) { return (int64_t)mifwi::g_agent_tier.load(std::memory_order_relaxed); }
int64_t mifwi_slow_handoff_count(void) { return (int64_t)mifwi::g_slow_handoffs.load(std::memory_order_relaxed); }
int mifwi_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int mifwi_device_info(int device, int32_t *sclk_khz, int32_t *mclk_khz, int32_t *compute_units)
{
    int rc = mifwi::check_device(device);
    if (rc) return rc;
    int v = 0;
    if (sclk_khz) *sclk_khz = hipDeviceGetAttribute(&v, hipDeviceAttributeClockRate, device) == hipSuccess ? v : 0;
    if (mclk_khz) *mclk_khz = hipDeviceGetAttribute(&v, hipDeviceAttributeMemoryClockRate, device) == hipSuccess ? v : 0;
    if (compute_units)
        *compute_units = hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess ? v : 0;
    (void)hipGetLastError();
    return MIFWI_OK;
}

int mifwi_acoustic_plan_create(mifwi_acoustic_plan **plan, int device,
                               const mifwi_acoustic_desc *d)
{
    if (!plan || !d) return mifwi::fail(MIFWI_EINVAL, "null plan/desc");
    if (d->n0 < 1 || d->n1 < 1 || d->nt < 1 || d->nshot < 1 || d->nsrc < 0 || d->nrec < 0)
        return mifwi::fail(MIFWI_EINVAL, "bad sizes n0=%d n1=%d nt=%d nshot=%d", d->n0, d->n1,
                           d->nt, d->nshot);
    if (d->ntap != 1 && d->ntap != 4) return mifwi::fail(MIFWI_EINVAL, "ntap must be 1 or 4");
    if (d->cpml_width < 0 || (d->cpml_width > 0 && (d->n0 < 2 * d->cpml_width + 4 || d->n1 < 2 * d->cpml_width + 4)))
        return mifwi::fail(MIFWI_EINVAL, "grid %dx%d cannot hold a %d-cell C-PML on every side", d->n0, d->n1, d->cpml_width);
    int rc = mifwi::check_device(device);
    if (rc) return rc;
    // function attributes and CU counts queried during set-up belong to THIS device (one process per
    // GPU sees all eight devices)
    MIFWI_HIP_TRY(hipSetDevice(device));
    mifwi_acoustic_plan *pl = new mifwi_acoustic_plan;
    pl->d = *d;
    pl->device = device;
    pl->ng = mifwi::ceil_div(d->n1, 4);
    pl->gp = 4 * pl->ng;
    pl->pmlW = d->cpml_width;
    // every sub-buffer of `work` starts on a 16-byte boundary (the per-shot sizes are even, not multiples of four, when
    // W * n0 is odd: an odd shot count would leave everything behind them on an 8-byte boundary)
    pl->pml_persist = pl->pmlW > 0 ? mifwi::round_up64(d->nshot * pml_persist_per_shot(pl->pmlW, d->n0, pl->gp), 4) : 0;
    pl->pml_scratch = pl->pmlW > 0 ? mifwi::round_up64(d->nshot * pml_scratch_per_shot(pl->pmlW, d->n0, pl->gp), 4) : 0;
    pl->zq_elems = pl->pmlW > 0 ? mifwi::round_up64(mifwi::round_up64(d->n0, 4) + pl->gp, 64) : 0;
    // one halo group left, interior, two spare groups right, rounded to 128-B lines
    pl->pitch = (int)mifwi::round_up64(4 * (pl->ng + 3), 32);
    pl->shot_stride = (long long)(d->n0 + 4) * pl->pitch;
    pl->field_elems = pl->shot_stride * d->nshot;
    pl->coef_elems = (long long)d->n0 * pl->gp;
    // lanes per row: widest tile whose padding waste stays small
    pl->lx = 16;
    for (int cand : {64, 32}) {
        const int padded = mifwi::ceil_div(pl->ng, cand) * cand;
        if (padded * 10 <= pl->ng * 12) { pl->lx = cand; break; }
    }
    pl->rz = 2;
    pl->xcd = env_int("MIFWI_AC_XCD", 1) != 0;
    // tuning overrides (benchmarks only)
    { const int v = env_int("MIFWI_AC_LX", 0); if (v == 16 || v == 32 || v == 64) pl->lx = v; }
    { const int v = env_int("MIFWI_AC_RZ", 0); if (v == 1 || v == 2 || v == 4 || v == 8) pl->rz = v; }
    if (pl->pmlW > 0) pl->rz = 2;       // the C-PML variant of ac_step is built for two rows per thread
    int gs = d->shots_per_group;
    if (gs <= 0) gs = env_int("MIFWI_AC_GS", 2);
    if (gs <= 0) gs = 2;
    if (gs > d->nshot) gs = d->nshot;
    pl->gs = gs;
    pl->ngroups = mifwi::ceil_div(d->nshot, gs);
    cluster_setup(pl);
    if (pl->cluster) {                 // the cluster adjoint keeps one accumulator per shot in registers
        pl->gs = 1;
        pl->ngroups = d->nshot;
    }
    {
        // Infinity Cache residency of the per-step family (same rule as the elastic plan): a pass over the time
        // range takes the shot groups whose wavefields (+ accumulators) fit 250 MB together with the model
        // forward: 200 MB (1000x3000, 2-shot groups: 3 groups = 165 MB run 126 us per step on every box tried,
        // 4 groups = 216 MB between 119 and 141 us, all 8 groups 165-200 us); adjoint: 250 MB as in the elastic plan
        const double model = 4.0 * (double)pl->coef_elems;
        const double gstate = 4.0 * 2.0 * (double)pl->shot_stride * pl->gs;
        auto fit = [&](double per_group, double resident) {
            int k = pl->ngroups;
            if (pl->ngroups * per_group + model > resident) {
                k = (int)std::floor((resident - model) / per_group);
                if (k < 1) k = pl->ngroups;
            }
            return k;
        };
        pl->pass_fwd = std::min(pl->ngroups, std::max(1, env_int("MIFWI_AC_PASS_GROUPS", fit(gstate, 200e6))));
        pl->pass_adj = std::min(pl->ngroups, std::max(1, env_int("MIFWI_AC_PASS_GROUPS", fit(gstate + model, 250e6))));
    }
    *plan = pl;
    return MIFWI_OK;
}

int mifwi_acoustic_plan_pass_sizes(const mifwi_acoustic_plan *plan, int32_t *forward_groups, int32_t *adjoint_groups)
{
    if (!plan) return mifwi::fail(MIFWI_EINVAL, "null plan");
    if (forward_groups) *forward_groups = plan->pass_fwd;
    if (adjoint_groups) *adjoint_groups = plan->pass_adj;
    return MIFWI_OK;
}

int mifwi_acoustic_plan_cluster_slabs(const mifwi_acoustic_plan *plan, int32_t adjoint)
{
    (void)adjoint;                       // one slab count serves both loops
    return plan && plan->cluster ? plan->NW : 0;
}

int mifwi_acoustic_plan_destroy(mifwi_acoustic_plan *plan)
{
    delete plan;
    return MIFWI_OK;
}

int mifwi_acoustic_plan_layout(const mifwi_acoustic_plan *pl, mifwi_acoustic_layout *out)
{
    if (!pl || !out) return mifwi::fail(MIFWI_EINVAL, "null plan/layout");
    out->gp = pl->gp;
    out->pitch = pl->pitch;
    out->ngroups = pl->ngroups;
    out->shots_per_group = pl->gs;
    out->field_elems = pl->field_elems;
    out->coef_elems = pl->coef_elems;
    const long long bbox = mifwi::round_up64(4LL * pl->d.nshot, 64);
    const long long cl = pl->cluster ? pl->xbuf_elems + pl->list_elems : 0;
    // single-launch plans: room for a copy of the input state of a resumed call (cluster_backup)
    const long long pml = pl->pml_persist + pl->pml_scratch + pl->zq_elems;
    out->work_forward_elems = 2 * pl->field_elems + pml + bbox + cl + (pl->cluster ? 2 * pl->field_elems + pl->pml_persist : 0);
    out->work_backward_elems = 2 * pl->field_elems + pl->ngroups * pl->coef_elems + pml + bbox + cl +
                               (pl->cluster ? 2 * pl->field_elems + pl->pml_persist + pl->ngroups * pl->coef_elems : 0);
    out->state_elems = 2 * pl->field_elems + pl->pml_persist;
    return MIFWI_OK;
}

int mifwi_acoustic_forward(mifwi_acoustic_plan *pl, const float *r, const float *q0,
                           const float *q1, const float *f, const int32_t *src_cell,
                           const float *src_w, const int32_t *rec_cell, const float *rec_w,
                           float *rec_out, float *snap, float *work, int32_t n_begin,
                           int32_t n_end, int32_t flags, void *stream)
{
    if (!pl || !r || !q0 || !q1 || !work) return mifwi::fail(MIFWI_EINVAL, "null argument");
    const mifwi_acoustic_desc &d = pl->d;
    if (n_begin < 0 || n_end > d.nt || n_begin > n_end)
        return mifwi::fail(MIFWI_EINVAL, "bad step range [%d,%d) for nt=%d", n_begin, n_end, d.nt);
    if (d.nsrc > 0 && (!f || !src_cell || !src_w))
        return mifwi::fail(MIFWI_EINVAL, "sources declared but f/src_cell/src_w is null");
    if (rec_out && (!rec_cell || !rec_w))
        return mifwi::fail(MIFWI_EINVAL, "rec_out given but rec_cell/rec_w is null");
    int rc = mifwi::check_device(pl->device);
    if (rc) return rc;
    MIFWI_HIP_TRY(hipSetDevice(pl->device));
    hipStream_t st = (hipStream_t)stream;
    float *ua = work, *ub = work + pl->field_elems;
    // C-PML plans: [ua | ub | memory variables | scratch + e | zero q0, q1 | bbox]; q0 / q1 carry the a, b profiles
    float *pml_persist = work + 2 * pl->field_elems, *pml_scratch = pml_persist + pl->pml_persist;
    float *zq = pml_scratch + pl->pml_scratch;
    int *bbox = reinterpret_cast<int *>(zq + pl->zq_elems);
    if (flags & MIFWI_ZERO_STATE)
        MIFWI_HIP_TRY(hipMemsetAsync(work, 0, sizeof(float) * (2 * pl->field_elems + pl->pml_persist), st));
    AcPml pm;
    memset(&pm, 0, sizeof(pm));
    if (pl->pmlW > 0) {
        MIFWI_HIP_TRY(hipMemsetAsync(zq, 0, sizeof(float) * pl->zq_elems, st));
        pm = pml_params(pl, q0, q1, pml_persist, pml_scratch);
        q0 = zq; q1 = zq + mifwi::round_up64(d.n0, 4);      // q1 is read with 16-byte loads
    }
    if (d.nsrc > 0)
        hipLaunchKernelGGL(points_bbox, dim3(d.nshot), dim3(kThreads), 0, st, src_cell,
                           d.nsrc * d.ntap, d.n1, bbox);
    AcParams p = base_params(pl, r, q0, q1);
    p.pmlW = pl->pmlW; p.pml_adj = 0; p.pe0 = pm.e0; p.pe1 = pm.e1;
    p.ninj = d.nsrc; p.ntap_inj = d.ntap; p.inj_mode = 0;
    p.inj_cell = src_cell; p.inj_w = src_w; p.inj_bbox = bbox;
    p.nsmp = rec_out ? d.nrec : 0; p.ntap_smp = d.ntap; p.smp_mode = 0;
    p.smp_cell = rec_cell; p.smp_w = rec_w;
    const long long snap_step = (long long)d.nshot * pl->coef_elems;
    if (pl->cluster && n_end > n_begin) {
        float *xbuf = reinterpret_cast<float *>(bbox) + mifwi::round_up64(4LL * d.nshot, 64);
        ClParams c = cluster_params(pl, r, q0, q1, ua, ub, xbuf);
        c.n_first = n_begin; c.n_last = n_end;
        c.src_cell = src_cell; c.src_w = src_w; c.f = f;
        c.rec_cell = rec_cell; c.rec_w = rec_w; c.rec_out = (rec_out && d.nrec > 0) ? rec_out : nullptr;
        c.G = snap; c.g_first = n_begin; c.g_step = snap_step;
        c.pml = pm;
        const long long fstate = 2 * pl->field_elems + pl->pml_persist;       // fields + the layer's memory variables
        float *backup = xbuf + pl->xbuf_elems + pl->list_elems;
        rc = cluster_backup(work, fstate, backup, flags, st);
        if (rc) return rc;
        rc = snap ? cluster_run<1>(pl, c, xbuf, st, work, fstate, backup, flags)
                  : cluster_run<0>(pl, c, xbuf, st, work, fstate, backup, flags);
        if (rc != mifwi::kClusterTimedOut) return rc;
        mifwi::note_fallback("acoustic");
        rc = cluster_restore(work, fstate, backup, flags, st);
        if (rc) return rc;
    }
    // shot groups are independent: a few at a time keep wavefields and model inside the Infinity Cache
    for (int g0 = 0; g0 < pl->ngroups; g0 += pl->pass_fwd)
    for (int n = n_begin; n < n_end; ++n) {
        const int cg = std::min(pl->pass_fwd, pl->ngroups - g0);
        p.g0 = g0;
        p.cur = (n & 1) ? ub : ua;
        p.prev = (n & 1) ? ua : ub;
        p.inj_amp = f ? f + (long long)n * d.nshot * d.nsrc : nullptr;
        p.smp_out = (rec_out && d.nrec > 0) ? rec_out + (long long)n * d.nshot * d.nrec : nullptr;
        if (pl->pmlW > 0) pml_step(pm, p.cur, g0 * pl->gs, std::min(cg * pl->gs, d.nshot - g0 * pl->gs), false, st);
        if (snap) {
            p.G = snap + (long long)(n - n_begin) * snap_step;
            launch_step<true, false>(pl, p, st, cg);
        } else {
            launch_step<false, false>(pl, p, st, cg);
        }
    }
    MIFWI_HIP_TRY(hipGetLastError());
    return MIFWI_OK;
}

int mifwi_acoustic_born(mifwi_acoustic_plan *pl, const float *r, const float *q0, const float *q1,
                        const float *dr, const int32_t *rec_cell, const float *rec_w, const float *snap,
                        int32_t snap_first, float *drec_out, float *work, int32_t n_begin, int32_t n_end,
                        int32_t flags, void *stream)
{
    if (!pl || !r || !q0 || !q1 || !dr || !snap || !work || !drec_out || !rec_cell || !rec_w)
        return mifwi::fail(MIFWI_EINVAL, "null argument");
    const mifwi_acoustic_desc &d = pl->d;
    if (n_begin < 0 || n_end > d.nt || n_begin > n_end)
        return mifwi::fail(MIFWI_EINVAL, "bad step range [%d,%d) for nt=%d", n_begin, n_end, d.nt);
    if (snap_first > n_begin)
        return mifwi::fail(MIFWI_EINVAL, "snapshots start at step %d but step %d is needed", snap_first, n_begin);
    int rc = mifwi::check_device(pl->device);
    if (rc) return rc;
    MIFWI_HIP_TRY(hipSetDevice(pl->device));
    hipStream_t st = (hipStream_t)stream;
    float *ua = work, *ub = work + pl->field_elems;
    float *pml_persist = work + 2 * pl->field_elems, *pml_scratch = pml_persist + pl->pml_persist;
    float *zq = pml_scratch + pl->pml_scratch;
    if (flags & MIFWI_ZERO_STATE)
        MIFWI_HIP_TRY(hipMemsetAsync(work, 0, sizeof(float) * (2 * pl->field_elems + pl->pml_persist), st));
    AcPml pm;
    memset(&pm, 0, sizeof(pm));
    if (pl->pmlW > 0) {
        MIFWI_HIP_TRY(hipMemsetAsync(zq, 0, sizeof(float) * pl->zq_elems, st));
        pm = pml_params(pl, q0, q1, pml_persist, pml_scratch);
        q0 = zq; q1 = zq + mifwi::round_up64(d.n0, 4);      // q1 is read with 16-byte loads
    }
    const long long snap_step = (long long)d.nshot * pl->coef_elems;
    if (pl->cluster && pl->pmlW == 0 && n_end > n_begin) {          // (the Born pass of a C-PML plan runs one launch per step)
        float *xbuf = zq + pl->zq_elems + mifwi::round_up64(4LL * d.nshot, 64);
        ClParams c = cluster_params(pl, r, q0, q1, ua, ub, xbuf);
        c.n_first = n_begin; c.n_last = n_end;
        c.nsrc = 0; c.src_cell = nullptr; c.src_w = nullptr; c.f = nullptr;      // no point source
        c.rec_cell = rec_cell; c.rec_w = rec_w; c.rec_out = d.nrec > 0 ? drec_out : nullptr;
        c.G = const_cast<float *>(snap); c.g_first = snap_first; c.g_step = snap_step;
        c.born_dr = dr;
        float *backup = xbuf + pl->xbuf_elems + pl->list_elems;
        rc = cluster_backup(work, 2 * pl->field_elems, backup, flags, st);
        if (rc) return rc;
        rc = cluster_run<3>(pl, c, xbuf, st, work, 2 * pl->field_elems, backup, flags);
        if (rc != mifwi::kClusterTimedOut) return rc;
        mifwi::note_fallback("acoustic");
        rc = cluster_restore(work, 2 * pl->field_elems, backup, flags, st);
        if (rc) return rc;
    }
    AcParams p = base_params(pl, r, q0, q1);
    p.ninj = 0;
    p.nsmp = d.nrec; p.ntap_smp = d.ntap; p.smp_mode = 0; p.smp_cell = rec_cell; p.smp_w = rec_w;
    p.born_dr = dr;
    p.pmlW = pl->pmlW; p.pml_adj = 0; p.pe0 = pm.e0; p.pe1 = pm.e1;
    for (int n = n_begin; n < n_end; ++n) {
        p.cur = (n & 1) ? ub : ua;
        p.prev = (n & 1) ? ua : ub;
        p.smp_out = d.nrec > 0 ? drec_out + (long long)n * d.nshot * d.nrec : nullptr;
        p.G = const_cast<float *>(snap) + (long long)(n - snap_first) * snap_step;
        if (pl->pmlW > 0) pml_step(pm, p.cur, 0, d.nshot, false, st);
        launch_step<false, true>(pl, p, st);
    }
    MIFWI_HIP_TRY(hipGetLastError());
    return MIFWI_OK;
}

int mifwi_acoustic_backward(mifwi_acoustic_plan *pl, const float *r, const float *q0,
                            const float *q1, const int32_t *src_cell, const float *src_w,
                            const int32_t *rec_cell, const float *rec_w, const float *grad_rec,
                            const float *snap, int32_t snap_first, float *grad_r, float *grad_f,
                            float *work, int32_t k_hi, int32_t k_lo, int32_t flags, void *stream)
{
    if (!pl || !r || !q0 || !q1 || !work || !snap || !grad_rec || !rec_cell || !rec_w)
        return mifwi::fail(MIFWI_EINVAL, "null argument");
    const mifwi_acoustic_desc &d = pl->d;
    if (k_lo < 1 || k_hi > d.nt - 1 || k_lo > k_hi + 1)
        return mifwi::fail(MIFWI_EINVAL, "bad adjoint range k=%d..%d for nt=%d", k_hi, k_lo, d.nt);
    if (snap_first > k_lo - 1)
        return mifwi::fail(MIFWI_EINVAL, "snapshots start at step %d but step %d is needed",
                           snap_first, k_lo - 1);
    if (grad_f && (!src_cell || !src_w))
        return mifwi::fail(MIFWI_EINVAL, "grad_f requested but src_cell/src_w is null");
    if ((flags & MIFWI_FINALIZE) && !grad_r)
        return mifwi::fail(MIFWI_EINVAL, "finalize requested but grad_r is null");
    int rc = mifwi::check_device(pl->device);
    if (rc) return rc;
    MIFWI_HIP_TRY(hipSetDevice(pl->device));
    hipStream_t st = (hipStream_t)stream;
    float *za = work, *zb = work + pl->field_elems;
    // C-PML plans: [za | zb | adjoint memory variables | acc | scratch + e | zero q0, q1 | bbox]
    float *pml_persist = work + 2 * pl->field_elems;
    float *acc = pml_persist + pl->pml_persist;
    float *pml_scratch = acc + (long long)pl->ngroups * pl->coef_elems;
    float *zq = pml_scratch + pl->pml_scratch;
    int *bbox = reinterpret_cast<int *>(zq + pl->zq_elems);
    if (flags & MIFWI_ZERO_STATE)
        MIFWI_HIP_TRY(hipMemsetAsync(
            work, 0, sizeof(float) * (2 * pl->field_elems + pl->pml_persist + pl->ngroups * pl->coef_elems), st));
    AcPml pm;
    memset(&pm, 0, sizeof(pm));
    if (pl->pmlW > 0) {
        MIFWI_HIP_TRY(hipMemsetAsync(zq, 0, sizeof(float) * pl->zq_elems, st));
        pm = pml_params(pl, q0, q1, pml_persist, pml_scratch);
        q0 = zq; q1 = zq + mifwi::round_up64(d.n0, 4);      // q1 is read with 16-byte loads
    }
    hipLaunchKernelGGL(points_bbox, dim3(d.nshot), dim3(kThreads), 0, st, rec_cell,
                       d.nrec * d.ntap, d.n1, bbox);
    AcParams p = base_params(pl, r, q0, q1);
    p.pmlW = pl->pmlW; p.pml_adj = 1; p.pe0 = pm.e0; p.pe1 = pm.e1;
    p.acc = acc;
    p.ninj = d.nrec; p.ntap_inj = d.ntap; p.inj_mode = 1;
    p.inj_cell = rec_cell; p.inj_w = rec_w; p.inj_bbox = bbox;
    const bool want_f = grad_f != nullptr && d.nsrc > 0;
    p.nsmp = want_f ? d.nsrc : 0; p.ntap_smp = d.ntap; p.smp_mode = 1;
    p.smp_cell = src_cell; p.smp_w = src_w;
    const long long snap_step = (long long)d.nshot * pl->coef_elems;
    // step k computes z^k from cur = z^{k+1}, prev = z^{k+2}; samples grad_f[k] from z^{k+1}.
    // Buffer parity is absolute in k so that a range can be resumed by a later call.
    bool per_step = true;
    if (pl->cluster && k_hi >= k_lo) {
        float *xbuf = reinterpret_cast<float *>(bbox) + mifwi::round_up64(4LL * d.nshot, 64);
        int *lists = reinterpret_cast<int *>(xbuf + pl->xbuf_elems);
        hipLaunchKernelGGL(cl_build_slab_lists, dim3(d.nshot), dim3(256), 0, st, rec_cell, d.nrec, d.n0, d.n1,
                           pl->NW, pl->rt, lists, lists + (long long)d.nshot * pl->NW);
        ClParams c = cluster_params(pl, r, q0, q1, za, zb, xbuf);
        c.n_first = k_hi; c.n_last = k_lo;
        c.src_cell = src_cell; c.src_w = src_w;
        c.rec_cell = rec_cell; c.rec_w = rec_w; c.grad_rec = grad_rec;
        c.grad_f = want_f ? grad_f : nullptr;
        c.G = const_cast<float *>(snap); c.g_first = snap_first; c.g_step = snap_step;
        c.acc = acc;
        c.slab_cnt = lists; c.slab_list = lists + (long long)d.nshot * pl->NW;
        c.pml = pm;
        const long long state = 2 * pl->field_elems + pl->pml_persist + pl->ngroups * pl->coef_elems;     // adjoint fields (+ layer) + accumulators
        float *backup = reinterpret_cast<float *>(lists) + pl->list_elems;
        rc = cluster_backup(work, state, backup, flags, st);
        if (rc) return rc;
        rc = cluster_run<2>(pl, c, xbuf, st, work, state, backup, flags);
        if (rc == mifwi::kClusterTimedOut) {
            mifwi::note_fallback("acoustic");
            rc = cluster_restore(work, state, backup, flags, st);
            if (rc) return rc;
        } else {
            if (rc) return rc;
            per_step = false;
        }
    }
    for (int g0 = 0; per_step && g0 < pl->ngroups; g0 += pl->pass_adj)
    for (int k = k_hi; k >= k_lo; --k) {
        const int par = (d.nt - 1 - k) & 1;
        p.g0 = g0;
        p.cur = par ? zb : za;
        p.prev = par ? za : zb;
        p.inj_amp = grad_rec + (long long)k * d.nshot * d.nrec;
        p.G = const_cast<float *>(snap) + (long long)(k - 1 - snap_first) * snap_step;
        p.smp_out = want_f ? grad_f + (long long)k * d.nshot * d.nsrc : nullptr;
        const int cg = std::min(pl->pass_adj, pl->ngroups - g0);
        if (pl->pmlW > 0) pml_step(pm, p.cur, g0 * pl->gs, std::min(cg * pl->gs, d.nshot - g0 * pl->gs), true, st);
        launch_step<false, true>(pl, p, st, cg);
    }
    p.g0 = 0;
    if (flags & MIFWI_FINALIZE) {
        if (want_f) {
            // grad_f[k_lo-1] from z^{k_lo} (sampling workgroups only)
            AcParams s = p;
            const int par = (d.nt - 1 - (k_lo - 1)) & 1;
            s.cur = par ? zb : za;
            s.smp_out = grad_f + (long long)(k_lo - 1) * d.nshot * d.nsrc;
            s.tiles_z = 0;
            const int tiles_x = mifwi::ceil_div(pl->ng, pl->lx);
            const int extra =
                mifwi::ceil_div(mifwi::ceil_div(pl->gs * s.nsmp, kThreads), tiles_x);
            dim3 grid(tiles_x, extra, pl->ngroups), block(kThreads);
            s.pmlW = 0;                                         // sampling workgroups only: no stencil, no layer term
            hipLaunchKernelGGL((ac_step<16, 4, false, false>), grid, block, 0, st, s);
        }
        const long long ncoef = pl->coef_elems;
        hipLaunchKernelGGL(ac_finalize, dim3((unsigned)((ncoef + 255) / 256)), dim3(256), 0, st,
                           acc, pl->ngroups, d.n0, d.n1, pl->gp, r, q0, q1, grad_r);
    }
    MIFWI_HIP_TRY(hipGetLastError());
    return MIFWI_OK;
}

}  // extern "C"
