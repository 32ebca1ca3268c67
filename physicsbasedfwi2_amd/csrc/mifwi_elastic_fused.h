// Fused per-step kernels of the elastic propagator for grids that do not fit the LDS of a few CUs
// (HBM / Infinity-Cache-bound regime; included by mifwi_elastic.hip inside its anonymous namespace).
//
//   el_fwd_fused<SNAP> : V and S of one time step in ONE launch   (80 instead of 100 B per cell-step with f32
//                        snapshot planes, 70 instead of 90 with bf16 planes)
//   el_adj_fused<BF16> : S^T and V^T of one adjoint step in ONE launch (100 instead of 122 B; 90 instead of 112)
//
// A workgroup owns 16 rows x 64 cells (one 4-cell group per thread).  What the two-launch form exchanges through
// memory between its launches (new velocities; new adjoint velocities) is recomputed on a two-row / one-group halo
// and handed over through LDS, so a step reads its state once and writes it once.  Nothing is updated in place - a
// neighbouring workgroup may still need the old value of a cell this one owns - so a launch reads one copy of the
// state (fields + C-PML memory variables) and writes the other; the drivers ping-pong.
// What makes the form pay (the first version, round 1, did not): every stencil operand is staged ONCE in LDS by
// plain 16-byte loads issued up front (13 per thread instead of 24 through per-thread z windows), the register
// budget is capped at 128 so that four workgroups share a CU, and the time loop runs over as many shots at a time
// as keep BOTH copies of their state inside the Infinity Cache.
// Arithmetic: the same fmaf chains as el_step_v / el_step_s / el_adj_s / el_adj_v, term by term (bit-identical
// results; tests/test_elastic_gpu.py::test_fused_* compare the two forms).

constexpr int FTZ = 16;              // owned rows
constexpr int FTG = 16;              // owned groups per row
constexpr int FSG = FTG + 4;         // LDS groups per row: owned + one halo group + one margin group per side
constexpr int FSW = 4 * FSG;         // floats per LDS row (80: rows start 16 banks apart)
constexpr int FRV = FTZ + 4;         // rows of the planes needed on the two-row halo (new velocities, sxx)
constexpr int FRS = FTZ + 8;         // rows of the planes needed two rows further out (sxz, szz)
constexpr int kFHaloV = 2 * 2 * (FTG + 2) + 2 * FTZ;       // halo groups of the FRV x (FTG+2) region: 104
constexpr int kFHaloS = 2 * 4 * (FTG + 2) + 2 * FTZ;       // halo groups of the FRS x (FTG+2) region: 176
static_assert(FTZ * FTG == kThreads, "one owned 4-cell group per thread");
static_assert(kFHaloS <= kThreads, "one halo item per thread");

// staged coordinates (row, group) of halo item h of a (FTZ + 2*HR) x (FTG + 2) region around the owned block:
// HR rows above, HR rows below (full width), then the left / right group of every owned row
template <int HR>
__device__ __forceinline__ void fhalo_item(int h, int &r, int &g)
{
    constexpr int W = FTG + 2;
    if (h < HR * W) { r = h / W; g = h - r * W; }
    else if (h < 2 * HR * W) { const int k = h - HR * W; r = FTZ + HR + k / W; g = k % W; }
    else { const int k = h - 2 * HR * W; r = HR + (k >> 1); g = (k & 1) * (W - 1); }
}

__device__ __forceinline__ float4 lds4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void sts4(float *p, const float4 &v) { *reinterpret_cast<float4 *>(p) = v; }

struct FwdV { float4 vx, vz, bx, bz; };

// The V update of el_step_v on group (j, g), stencil operands from the LDS planes: `lr` = row of the group in the
// FRV-row planes, `lc` = its first column.  Memory variables read from psi*, written (owner only) to psi*_out.
__device__ __forceinline__ void fwd_v_update(const ElParams &p, int s, int j, int g, const float *Lxx, const float *Lzz,
                                             const float *Lxz, int lr, int lc, const FwdV &in, bool mine,
                                             float4 &vxn, float4 &vzn, float *s4v, float *s5v)
{
    const FdK K = p.K;
    // windows: sxz rows j-2..j+1 (a0..a3), szz rows j-1..j+2 (b0..b3); the FRS-row planes start two rows earlier
    float4 a0 = lds4(Lxz + (lr + 0) * FSW + lc), a1 = lds4(Lxz + (lr + 1) * FSW + lc);
    const float4 a2 = lds4(Lxz + (lr + 2) * FSW + lc), a3 = lds4(Lxz + (lr + 3) * FSW + lc);
    float4 b0 = lds4(Lzz + (lr + 1) * FSW + lc);
    const float4 b1 = lds4(Lzz + (lr + 2) * FSW + lc), b2 = lds4(Lzz + (lr + 3) * FSW + lc);
    const float4 b3 = lds4(Lzz + (lr + 4) * FSW + lc);
    {
        // free surface, odd mirroring about row 0: sxz(-m) = -sxz(m-1), szz(-m) = -szz(m)
        const bool m0 = p.fsurf && j == 0, m1 = p.fsurf && j == 1;
#define MIFWI_SEL(dst, c, v) dst.x = (c) ? -(v).x : dst.x; dst.y = (c) ? -(v).y : dst.y; \
                             dst.z = (c) ? -(v).z : dst.z; dst.w = (c) ? -(v).w : dst.w
        MIFWI_SEL(a0, m1, a1);
        MIFWI_SEL(a0, m0, a3);
        MIFWI_SEL(a1, m0, a2);
        MIFWI_SEL(b0, m0, b2);
#undef MIFWI_SEL
    }
    const Row8 xx = row8(Lxx + lr * FSW, lc), xz = row8(Lxz + (lr + 2) * FSW, lc);
    float d1[4], d2[4], d3[4], d4[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        d1[c] = dfw(K, xx.v[c + 1], xx.v[c + 2], xx.v[c + 3], xx.v[c + 4]);
        d2[c] = dbw(K, comp(a0, c), comp(a1, c), comp(a2, c), comp(a3, c));
        d3[c] = dbw(K, xz.v[c], xz.v[c + 1], xz.v[c + 2], xz.v[c + 3]);
        d4[c] = dfw(K, comp(b0, c), comp(b1, c), comp(b2, c), comp(b3, c));
    }
    const int xs_off = xstrip(p, g);
    if (xs_off >= 0) {
        const float4 pxa = ld4(p.px + PA * p.gp + 4 * g), pxb = ld4(p.px + PB * p.gp + 4 * g);
        const float4 pxk = ld4(p.px + PK * p.gp + 4 * g), pxah = ld4(p.px + PAH * p.gp + 4 * g);
        const float4 pxbh = ld4(p.px + PBH * p.gp + 4 * g), pxkh = ld4(p.px + PKH * p.gp + 4 * g);
        const long long q1 = (long long)s * p.psix_shot + ((long long)0 * p.nz + j) * p.wx + xs_off;
        const long long q3 = (long long)s * p.psix_shot + ((long long)1 * p.nz + j) * p.wx + xs_off;
        const float4 s1 = ld4(p.psix + q1), s3 = ld4(p.psix + q3);
        float t1[4] = {s1.x, s1.y, s1.z, s1.w}, t3[4] = {s3.x, s3.y, s3.z, s3.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            d1[c] = pml(t1[c], comp(pxah, c), comp(pxbh, c), comp(pxkh, c), d1[c]);
            d3[c] = pml(t3[c], comp(pxa, c), comp(pxb, c), comp(pxk, c), d3[c]);
        }
        if (mine) {
            st4(p.psix_out + q1, make_float4(t1[0], t1[1], t1[2], t1[3]));
            st4(p.psix_out + q3, make_float4(t3[0], t3[1], t3[2], t3[3]));
        }
    }
    const int zs = zstrip(p, j);
    if (zs >= 0) {
        const float za = p.pz[PA * p.nz + j], zb = p.pz[PB * p.nz + j], zk = p.pz[PK * p.nz + j];
        const float zah = p.pz[PAH * p.nz + j], zbh = p.pz[PBH * p.nz + j], zkh = p.pz[PKH * p.nz + j];
        const long long q2 = (long long)s * p.psiz_shot + ((long long)0 * 2 * p.W + zs) * p.gp + 4 * g;
        const long long q4 = (long long)s * p.psiz_shot + ((long long)1 * 2 * p.W + zs) * p.gp + 4 * g;
        const float4 s2 = ld4(p.psiz + q2), s4 = ld4(p.psiz + q4);
        float t2[4] = {s2.x, s2.y, s2.z, s2.w}, t4[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            d2[c] = pml(t2[c], za, zb, zk, d2[c]);
            d4[c] = pml(t4[c], zah, zbh, zkh, d4[c]);
        }
        if (mine) {
            st4(p.psiz_out + q2, make_float4(t2[0], t2[1], t2[2], t2[3]));
            st4(p.psiz_out + q4, make_float4(t4[0], t4[1], t4[2], t4[3]));
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) { s4v[c] = d1[c] + d2[c]; s5v[c] = d3[c] + d4[c]; }
    vxn = make_float4(fmaf(in.bx.x, s4v[0], in.vx.x), fmaf(in.bx.y, s4v[1], in.vx.y),
                      fmaf(in.bx.z, s4v[2], in.vx.z), fmaf(in.bx.w, s4v[3], in.vx.w));
    vzn = make_float4(fmaf(in.bz.x, s5v[0], in.vz.x), fmaf(in.bz.y, s5v[1], in.vz.y),
                      fmaf(in.bz.z, s5v[2], in.vz.z), fmaf(in.bz.w, s5v[3], in.vz.w));
}

// Receivers sample the INPUT state, i.e. the velocities of the previous step (the driver shifts the output row by
// one and samples the last step with a launch of its own).
template <int SNAP>      // 0: no snapshots, 1: f32 planes, 2: bf16 planes
__global__ __launch_bounds__(kThreads, 4) void el_fwd_fused(const ElParams p)
{
    const FdK K = p.K;
    int bx, by, bz;
    xcd_tile(p, bx, by, bz);
    if (by >= p.tiles_z) {
        sample_points<0>(p, bx, by, bz);
        return;
    }
    __shared__ __attribute__((aligned(16))) float Lxz[FRS * FSW];
    __shared__ __attribute__((aligned(16))) float Lzz[FRS * FSW];
    __shared__ __attribute__((aligned(16))) float Lxx[FRV * FSW];
    __shared__ __attribute__((aligned(16))) float Vx[FRV * FSW];
    __shared__ __attribute__((aligned(16))) float Vz[FRV * FSW];
    __shared__ float inj[FTZ * 4 * FTG];
    const int tile_j = by * FTZ, tile_g = bx * FTG;
    const int s = p.s0 + bz;
    const unsigned fs = p.field_stride;
    const unsigned ncell = (unsigned)p.nz * p.gp;
    const int t = (int)threadIdx.x;
    const float *fin = p.fields + (long long)s * p.shot_stride;
    float *fout = p.fields_out + (long long)s * p.shot_stride;
    const float *gxx = fin + F_SXX * fs, *gzz = fin + F_SZZ * fs, *gxz = fin + F_SXZ * fs;
    const int orow = t / FTG, ogrp = t % FTG;
    const int oj = tile_j + orow, og = tile_g + ogrp;
    const bool own_ok = oj < p.nz && og < p.ng;
    const unsigned occ = (unsigned)oj * p.gp + 4 * og;
    const unsigned oo = (unsigned)(oj + 2) * p.pitch + 4 + 4 * og;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    // ---- every global operand of the tile is requested before the first use --------------------------------------
    // halo item of the velocity region (t < 104): row hvr, group hvg of the FRV x (FTG+2) region
    int hvr = 0, hvg = 0, hsr = 0, hsg = 0;
    if (t < kFHaloV) fhalo_item<2>(t, hvr, hvg);
    if (t < kFHaloS) fhalo_item<4>(t, hsr, hsg);
    const int hvj = tile_j - 2 + hvr, hvgg = tile_g - 1 + hvg;
    const bool hv_ok = t < kFHaloV && hvj >= 0 && hvj < p.nz && hvgg >= 0 && hvgg < p.ng;
    const int hsj = tile_j - 4 + hsr, hsgg = tile_g - 1 + hsg;
    const bool hs_ok = t < kFHaloS && hsj >= 0 && hsj < p.nz && hsgg >= 0 && hsgg < p.ng;
    const unsigned hvo = (unsigned)(hvj + 2) * p.pitch + 4 + 4 * hvgg;
    const unsigned hso = (unsigned)(hsj + 2) * p.pitch + 4 + 4 * hsgg;
    FwdV own, halo;
    own.vx = own.vz = own.bx = own.bz = halo.vx = halo.vz = halo.bx = halo.bz = zero4;
    float4 oxx = zero4, ozz = zero4, oxz = zero4, hxx = zero4, hzz = zero4, hxz = zero4;
    if (own_ok) { oxx = ld4(gxx + oo); ozz = ld4(gzz + oo); oxz = ld4(gxz + oo); }
    if (hs_ok) { hzz = ld4(gzz + hso); hxz = ld4(gxz + hso); }
    if (hv_ok) hxx = ld4(gxx + hvo);
    if (own_ok) {
        own.vx = ld4(fin + F_VX * fs + oo); own.vz = ld4(fin + F_VZ * fs + oo);
        own.bx = ld4(p.mat + M_BX * ncell + occ); own.bz = ld4(p.mat + M_BZ * ncell + occ);
    }
    if (hv_ok) {
        const unsigned hcc = (unsigned)hvj * p.gp + 4 * hvgg;
        halo.vx = ld4(fin + F_VX * fs + hvo); halo.vz = ld4(fin + F_VZ * fs + hvo);
        halo.bx = ld4(p.mat + M_BX * ncell + hcc); halo.bz = ld4(p.mat + M_BZ * ncell + hcc);
    }
    // the margin groups (never part of a result) hold zeros, not whatever the LDS held before
    if (t < FRS) {
        sts4(Lxz + t * FSW, zero4); sts4(Lxz + t * FSW + FSW - 4, zero4);
        sts4(Lzz + t * FSW, zero4); sts4(Lzz + t * FSW + FSW - 4, zero4);
        if (t < FRV) { sts4(Lxx + t * FSW, zero4); sts4(Lxx + t * FSW + FSW - 4, zero4); }
    }
    sts4(Lxx + (orow + 2) * FSW + 4 * (ogrp + 2), oxx);
    sts4(Lzz + (orow + 4) * FSW + 4 * (ogrp + 2), ozz);
    sts4(Lxz + (orow + 4) * FSW + 4 * (ogrp + 2), oxz);
    if (t < kFHaloS) {
        sts4(Lzz + hsr * FSW + 4 * (hsg + 1), hzz);
        sts4(Lxz + hsr * FSW + 4 * (hsg + 1), hxz);
    }
    if (t < kFHaloV) sts4(Lxx + hvr * FSW + 4 * (hvg + 1), hxx);
    const bool has_inj = stage_injection<FTZ, 4 * FTG, 1>(p, s, tile_j, 4 * tile_g, inj);
    __syncthreads();
    // ---- V: new velocities of the tile + halo into LDS ----------------------------------------------------------
    {
        float4 vxn = zero4, vzn = zero4;
        float s4v[4], s5v[4];
        if (own_ok) {
            fwd_v_update(p, s, oj, og, Lxx, Lzz, Lxz, orow + 2, 4 * (ogrp + 2), own, true, vxn, vzn, s4v, s5v);
            st4(fout + F_VX * fs + oo, vxn);
            st4(fout + F_VZ * fs + oo, vzn);
            if (SNAP == 1) {
                float *Sp = p.S + (long long)s * p.snap_shot + snap_cell(p, oj, og);
                mifwi::stnt4(Sp + 3 * (long long)p.splane, make_float4(s4v[0], s4v[1], s4v[2], s4v[3]));
                mifwi::stnt4(Sp + 4 * (long long)p.splane, make_float4(s5v[0], s5v[1], s5v[2], s5v[3]));
            } else if (SNAP == 2) {
                bf_store2(p.S + (long long)s * p.snap_shot + bf_reg_de(p.splane), snap_cell(p, oj, og) >> 2, s4v, s5v);
            }
        }
        sts4(Vx + (orow + 2) * FSW + 4 * (ogrp + 2), vxn);
        sts4(Vz + (orow + 2) * FSW + 4 * (ogrp + 2), vzn);
    }
    // the S phase's own operands fly while the halo velocities are computed and the barrier is crossed
    float4 Ls = zero4, Ms = zero4, mus = zero4;
    if (own_ok) {
        Ls = ld4(p.mat + M_L * ncell + occ); Ms = ld4(p.mat + M_M * ncell + occ);
        mus = ld4(p.mat + M_MU * ncell + occ);
    }
    if (t < kFHaloV) {
        float4 vxn = zero4, vzn = zero4;
        float s4v[4], s5v[4];
        if (hv_ok) fwd_v_update(p, s, hvj, hvgg, Lxx, Lzz, Lxz, hvr, 4 * (hvg + 1), halo, false, vxn, vzn, s4v, s5v);
        sts4(Vx + hvr * FSW + 4 * (hvg + 1), vxn);
        sts4(Vz + hvr * FSW + 4 * (hvg + 1), vzn);
    }
    __syncthreads();
    // ---- S: new stresses of the tile from LDS --------------------------------------------------------------------
    if (own_ok) {
        const int r = orow + 2, cb = 4 * (ogrp + 2);
        const Row8 xv = row8(Vx + r * FSW, cb), zv = row8(Vz + r * FSW, cb);
        const float4 a0 = lds4(Vz + (r - 2) * FSW + cb), a1 = lds4(Vz + (r - 1) * FSW + cb);
        const float4 a2 = lds4(Vz + r * FSW + cb), a3 = lds4(Vz + (r + 1) * FSW + cb);
        const float4 b0 = lds4(Vx + (r - 1) * FSW + cb), b1 = lds4(Vx + r * FSW + cb);
        const float4 b2 = lds4(Vx + (r + 1) * FSW + cb), b3 = lds4(Vx + (r + 2) * FSW + cb);
        oxx = lds4(Lxx + r * FSW + cb); ozz = lds4(Lzz + (r + 2) * FSW + cb); oxz = lds4(Lxz + (r + 2) * FSW + cb);
        float e1[4], e2[4], e3[4], e4[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            e1[c] = dbw(K, xv.v[c], xv.v[c + 1], xv.v[c + 2], xv.v[c + 3]);
            e2[c] = dbw(K, comp(a0, c), comp(a1, c), comp(a2, c), comp(a3, c));
            e3[c] = dfw(K, comp(b0, c), comp(b1, c), comp(b2, c), comp(b3, c));
            e4[c] = dfw(K, zv.v[c + 1], zv.v[c + 2], zv.v[c + 3], zv.v[c + 4]);
        }
        const int xs_off = xstrip(p, og);
        if (xs_off >= 0) {
            const float4 pxa = ld4(p.px + PA * p.gp + 4 * og), pxb = ld4(p.px + PB * p.gp + 4 * og);
            const float4 pxk = ld4(p.px + PK * p.gp + 4 * og), pxah = ld4(p.px + PAH * p.gp + 4 * og);
            const float4 pxbh = ld4(p.px + PBH * p.gp + 4 * og), pxkh = ld4(p.px + PKH * p.gp + 4 * og);
            const long long q5 = (long long)s * p.psix_shot + ((long long)2 * p.nz + oj) * p.wx + xs_off;
            const long long q8 = (long long)s * p.psix_shot + ((long long)3 * p.nz + oj) * p.wx + xs_off;
            const float4 s5 = ld4(p.psix + q5), s8 = ld4(p.psix + q8);
            float t5[4] = {s5.x, s5.y, s5.z, s5.w}, t8[4] = {s8.x, s8.y, s8.z, s8.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                e1[c] = pml(t5[c], comp(pxa, c), comp(pxb, c), comp(pxk, c), e1[c]);
                e4[c] = pml(t8[c], comp(pxah, c), comp(pxbh, c), comp(pxkh, c), e4[c]);
            }
            st4(p.psix_out + q5, make_float4(t5[0], t5[1], t5[2], t5[3]));
            st4(p.psix_out + q8, make_float4(t8[0], t8[1], t8[2], t8[3]));
        }
        const int zs = zstrip(p, oj);
        if (zs >= 0) {
            const float za = p.pz[PA * p.nz + oj], zb = p.pz[PB * p.nz + oj], zk = p.pz[PK * p.nz + oj];
            const float zah = p.pz[PAH * p.nz + oj], zbh = p.pz[PBH * p.nz + oj], zkh = p.pz[PKH * p.nz + oj];
            const long long q6 = (long long)s * p.psiz_shot + ((long long)2 * 2 * p.W + zs) * p.gp + 4 * og;
            const long long q7 = (long long)s * p.psiz_shot + ((long long)3 * 2 * p.W + zs) * p.gp + 4 * og;
            const float4 s6 = ld4(p.psiz + q6), s7 = ld4(p.psiz + q7);
            float t6[4] = {s6.x, s6.y, s6.z, s6.w}, t7[4] = {s7.x, s7.y, s7.z, s7.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                e2[c] = pml(t6[c], za, zb, zk, e2[c]);
                e3[c] = pml(t7[c], zah, zbh, zkh, e3[c]);
            }
            st4(p.psiz_out + q6, make_float4(t6[0], t6[1], t6[2], t6[3]));
            st4(p.psiz_out + q7, make_float4(t7[0], t7[1], t7[2], t7[3]));
        }
        float nxx[4], nzz[4], nxz[4], s3v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            s3v[c] = e3[c] + e4[c];
            nxx[c] = fmaf(comp(Ms, c), e1[c], fmaf(comp(Ls, c), e2[c], comp(oxx, c)));
            nzz[c] = fmaf(comp(Ls, c), e1[c], fmaf(comp(Ms, c), e2[c], comp(ozz, c)));
            nxz[c] = fmaf(comp(mus, c), s3v[c], comp(oxz, c));
            if (has_inj) {
                const float a = inj[orow * 4 * FTG + 4 * ogrp + c];
                nxx[c] += a;
                nzz[c] += a;
            }
            if (p.fsurf && oj == 0) nzz[c] = 0.f;
        }
        st4(fout + F_SXX * fs + oo, make_float4(nxx[0], nxx[1], nxx[2], nxx[3]));
        st4(fout + F_SZZ * fs + oo, make_float4(nzz[0], nzz[1], nzz[2], nzz[3]));
        st4(fout + F_SXZ * fs + oo, make_float4(nxz[0], nxz[1], nxz[2], nxz[3]));
        if (SNAP == 1) {
            float *Sp = p.S + (long long)s * p.snap_shot + snap_cell(p, oj, og);
            mifwi::stnt4(Sp, make_float4(e1[0], e1[1], e1[2], e1[3]));
            mifwi::stnt4(Sp + (long long)p.splane, make_float4(e2[0], e2[1], e2[2], e2[3]));
            mifwi::stnt4(Sp + 2 * (long long)p.splane, make_float4(s3v[0], s3v[1], s3v[2], s3v[3]));
        } else if (SNAP == 2) {
            float *Sp = p.S + (long long)s * p.snap_shot;
            bf_store2(Sp, snap_cell(p, oj, og) >> 2, e1, e2);
            bf_store1(Sp + bf_reg_c(p.splane), snap_cell(p, oj, og) >> 2, s3v);
        }
    }
}

// ================================================================================================================
// fused adjoint step: S^T and V^T of one step in ONE launch.
//   E = C^T sigma_bar (transposed C-PML) on the tile + 4 rows / 1 group  ->  LDS
//   v_bar' = v_bar - stencils(E) on the tile + 2 rows / 1 group (recomputing what the neighbours own; R^T g has
//            been added to the input state by el_inject_adjsrc);
//            the five material-gradient accumulators from the snapshot planes;  D = B^T v_bar' (transposed C-PML)
//   D -> LDS;  sigma_bar' = sigma_bar - stencils(D) on the tile
// Reads one copy of the adjoint state, writes the other (the memory variables of the adjoint already ping-pong in
// the two-launch form).  A workgroup walks the gs shots of its accumulator group with the accumulators in registers.
// ================================================================================================================
constexpr int kAFRows = 2 * FRS + 2 * FRV + 4 * FRV;       // E2, E3 (FRS rows), E1, E4 (FRV rows), D1..D4 (FRV rows)
constexpr int kAFElems = kAFRows * FSW;
// v_bar' of group (j, g) from the E planes: `lr` = row in the FRV-row planes, `lc` = first column
template <int SW = FSW>      // floats per LDS row of the planes
__device__ __forceinline__ void adj_v_update(const ElParams &p, int g, const float *E1, const float *E2, const float *E3,
                                             const float *E4, int lr, int lc, const float4 &vxb, const float4 &vzb,
                                             float *nvx, float *nvz)
{
    const FdK K = p.K;
    const Row8 x1 = row8(E1 + lr * SW, lc), x4 = row8(E4 + lr * SW, lc);
    const float4 z3a = lds4(E3 + (lr + 0) * SW + lc), z3b = lds4(E3 + (lr + 1) * SW + lc);
    const float4 z3c = lds4(E3 + (lr + 2) * SW + lc), z3d = lds4(E3 + (lr + 3) * SW + lc);
    const float4 z2a = lds4(E2 + (lr + 1) * SW + lc), z2b = lds4(E2 + (lr + 2) * SW + lc);
    const float4 z2c = lds4(E2 + (lr + 3) * SW + lc), z2d = lds4(E2 + (lr + 4) * SW + lc);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float dx1 = dfw(K, x1.v[c + 1], x1.v[c + 2], x1.v[c + 3], x1.v[c + 4]);
        const float dz3 = dbw(K, comp(z3a, c), comp(z3b, c), comp(z3c, c), comp(z3d, c));
        const float dz2 = dfw(K, comp(z2a, c), comp(z2b, c), comp(z2c, c), comp(z2d, c));
        const float dx4 = dbw(K, x4.v[c], x4.v[c + 1], x4.v[c + 2], x4.v[c + 3]);
        float ax = comp(vxb, c) - (dx1 + dz3);
        float az = comp(vzb, c) - (dz2 + dx4);
        if (4 * g + c >= p.nx) { ax = 0.f; az = 0.f; }
        nvx[c] = ax; nvz[c] = az;
    }
}

// index sets of a thread: its owned group, its halo item of the velocity region, its halo item of the stress region.
// Recomputed from an opaque thread index at the head of every phase: derived once, the compiler keeps a few dozen
// addresses alive across the whole step (and spills them).
struct FOwn { int orow, ogrp, oj, og; bool ok; unsigned occ, oo; };
struct FHalo { int r, g, j, gg; bool ok; unsigned cc, o; };
__device__ __forceinline__ FOwn f_own(const ElParams &p, int t, int tile_j, int tile_g)
{
    FOwn o;
    o.orow = t / FTG; o.ogrp = t % FTG;
    o.oj = tile_j + o.orow; o.og = tile_g + o.ogrp;
    o.ok = o.oj < p.nz && o.og < p.ng;
    o.occ = (unsigned)o.oj * p.gp + 4 * o.og;
    o.oo = (unsigned)(o.oj + 2) * p.pitch + 4 + 4 * o.og;
    return o;
}
template <int HR>
__device__ __forceinline__ FHalo f_halo(const ElParams &p, int t, int tile_j, int tile_g)
{
    FHalo h;
    h.r = 0; h.g = 0;
    constexpr int N = 2 * HR * (FTG + 2) + 2 * FTZ;
    if (t < N) fhalo_item<HR>(t, h.r, h.g);
    h.j = tile_j - HR + h.r; h.gg = tile_g - 1 + h.g;
    h.ok = t < N && h.j >= 0 && h.j < p.nz && h.gg >= 0 && h.gg < p.ng;
    h.cc = (unsigned)h.j * p.gp + 4 * h.gg;
    h.o = (unsigned)(h.j + 2) * p.pitch + 4 + 4 * h.gg;
    return h;
}

template <bool BF16>
__global__ __launch_bounds__(kThreads, 3) void el_adj_fused(const ElParams p)
{
    const FdK K = p.K;
    int bx, by, bz;
    xcd_tile(p, bx, by, bz);
    if (by >= p.tiles_z) {
        sample_points<1>(p, bx, by, bz);
        return;
    }
    __shared__ __attribute__((aligned(16))) float buf[kAFElems];      // 53,760 B: three workgroups per CU
    float *E2 = buf, *E3 = buf + FRS * FSW, *E1 = buf + 2 * FRS * FSW, *E4 = E1 + FRV * FSW;
    float *D1 = E4 + FRV * FSW, *D2 = D1 + FRV * FSW, *D3 = D2 + FRV * FSW, *D4 = D3 + FRV * FSW;
    const int tile_j = by * FTZ, tile_g = bx * FTG;
    const unsigned fs = p.field_stride;
    const unsigned ncell = (unsigned)p.nz * p.gp;
    const int t = (int)threadIdx.x;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 acc[5];
    {
        const FOwn o = f_own(p, t, tile_j, tile_g);
        const long long acc_base = (long long)(p.s0 / p.gs + bz) * 5 * p.splane + snap_cell(p, o.oj, o.og);
        if (o.ok) {
#pragma unroll
            for (int k = 0; k < 5; ++k) acc[k] = ld4(p.acc + acc_base + (long long)k * p.splane);
        }
    }
    // margins of the LDS rows: zeros (they only ever feed lanes whose results are discarded)
    for (int e = t; e < kAFRows; e += kThreads) {
        sts4(buf + e * FSW, zero4);
        sts4(buf + e * FSW + FSW - 4, zero4);
    }
    for (int si = 0; si < p.gs; ++si) {
        const int s = p.s0 + bz * p.gs + si;
        if (s >= p.nshot) break;
        const float *fin = p.fields + (long long)s * p.shot_stride;
        float *fout = p.fields_out + (long long)s * p.shot_stride;
        float4 ovx = zero4, ovz = zero4, hvx = zero4, hvz = zero4, obx = zero4, obz = zero4, hbx = zero4, hbz = zero4;
        float4 S1 = zero4, S2 = zero4, S3 = zero4, S4 = zero4, S5 = zero4;
        BfPlanes packed;
        packed.ab = packed.de = mifwi_u4{0u, 0u, 0u, 0u};
        packed.c = mifwi_u2{0u, 0u};
        float4 bxx = zero4, bzz = zero4, bxz = zero4;         // own adjoint stresses as S^T sees them
        // ---- E on the tile + halo ---------------------------------------------------------------------------------
        {
            const int tq = f_opaque(t);
            const FOwn o = f_own(p, tq, tile_j, tile_g);
            const FHalo h = f_halo<4>(p, tq, tile_j, tile_g);
            AdjIn own, halo;
            own.a = own.b = own.c = own.m0 = own.m1 = own.m2 = zero4;
            halo = own;
            if (o.ok) {
                own.a = ld4(fin + F_SXX * fs + o.oo); own.b = ld4(fin + F_SZZ * fs + o.oo); own.c = ld4(fin + F_SXZ * fs + o.oo);
                own.m0 = ld4(p.mat + M_L * ncell + o.occ); own.m1 = ld4(p.mat + M_M * ncell + o.occ);
                own.m2 = ld4(p.mat + M_MU * ncell + o.occ);
            }
            if (h.ok) {
                halo.a = ld4(fin + F_SXX * fs + h.o); halo.b = ld4(fin + F_SZZ * fs + h.o); halo.c = ld4(fin + F_SXZ * fs + h.o);
                halo.m0 = ld4(p.mat + M_L * ncell + h.cc); halo.m1 = ld4(p.mat + M_M * ncell + h.cc);
                halo.m2 = ld4(p.mat + M_MU * ncell + h.cc);
            }
            if (p.fsurf && o.oj == 0) own.b = zero4;             // S^T: the adjoint of szz(0,.) is discarded
            if (p.fsurf && h.j == 0) halo.b = zero4;
            bxx = own.a; bzz = own.b; bxz = own.c;
            {
                float4 e1 = zero4, e2 = zero4, e3 = zero4, e4 = zero4;
                if (o.ok) stage_E(p, s, o.oj, o.og, own, true, e1, e2, e3, e4);
                const int x = 4 * (o.ogrp + 2);
                sts4(E1 + (o.orow + 2) * FSW + x, e1); sts4(E4 + (o.orow + 2) * FSW + x, e4);
                sts4(E2 + (o.orow + 4) * FSW + x, e2); sts4(E3 + (o.orow + 4) * FSW + x, e3);
            }
            if (tq < kFHaloS) {
                float4 e1 = zero4, e2 = zero4, e3 = zero4, e4 = zero4;
                if (h.ok) stage_E(p, s, h.j, h.gg, halo, false, e1, e2, e3, e4);
                const int x = 4 * (h.g + 1);
                sts4(E2 + h.r * FSW + x, e2); sts4(E3 + h.r * FSW + x, e3);
                if (h.r >= 2 && h.r < FRS - 2) { sts4(E1 + (h.r - 2) * FSW + x, e1); sts4(E4 + (h.r - 2) * FSW + x, e4); }
            }
        }
        // operands of the second half: requested before the barrier, used after it
        {
            const int tq = f_opaque(t);
            const FOwn o = f_own(p, tq, tile_j, tile_g);
            const FHalo h = f_halo<2>(p, tq, tile_j, tile_g);
            if (o.ok) {
                ovx = ld4(fin + F_VX * fs + o.oo); ovz = ld4(fin + F_VZ * fs + o.oo);
                obx = ld4(p.mat + M_BX * ncell + o.occ); obz = ld4(p.mat + M_BZ * ncell + o.occ);
            }
            if (h.ok) {
                hvx = ld4(fin + F_VX * fs + h.o); hvz = ld4(fin + F_VZ * fs + h.o);
                hbx = ld4(p.mat + M_BX * ncell + h.cc); hbz = ld4(p.mat + M_BZ * ncell + h.cc);
            }
            if (o.ok && !BF16) {
                const float *Sp = p.S + (long long)s * p.snap_shot + snap_cell(p, o.oj, o.og);
                const long long sp = p.splane;
                S1 = mifwi::ldnt4(Sp); S2 = mifwi::ldnt4(Sp + sp); S3 = mifwi::ldnt4(Sp + 2 * sp);
                S4 = mifwi::ldnt4(Sp + 3 * sp); S5 = mifwi::ldnt4(Sp + 4 * sp);
            } else if (o.ok) {
                bf_request(p.S + (long long)s * p.snap_shot, p.splane, snap_cell(p, o.oj, o.og) >> 2, packed);
            }
        }
        __syncthreads();
        // ---- v_bar' (tile + halo), gradients, D ---------------------------------------------------------------------
        {
            const int tq = f_opaque(t);
            const FOwn o = f_own(p, tq, tile_j, tile_g);
            float4 oD1 = zero4, oD2 = zero4, oD3 = zero4, oD4 = zero4;
            if (o.ok) {
                float nvx[4], nvz[4];
                adj_v_update(p, o.og, E1, E2, E3, E4, o.orow + 2, 4 * (o.ogrp + 2), ovx, ovz, nvx, nvz);
                const float4 nx4 = make_float4(nvx[0], nvx[1], nvx[2], nvx[3]), nz4 = make_float4(nvz[0], nvz[1], nvz[2], nvz[3]);
                st4(fout + F_VX * fs + o.oo, nx4);
                st4(fout + F_VZ * fs + o.oo, nz4);
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
                stage_D(p, s, o.oj, o.og, nx4, nz4, obx, obz, true, oD1, oD2, oD3, oD4);
            }
            const int x = 4 * (o.ogrp + 2), r = o.orow + 2;
            sts4(D1 + r * FSW + x, oD1); sts4(D2 + r * FSW + x, oD2); sts4(D3 + r * FSW + x, oD3); sts4(D4 + r * FSW + x, oD4);
        }
        {
            const int tq = f_opaque(t);
            if (tq < kFHaloV) {
                const FHalo h = f_halo<2>(p, tq, tile_j, tile_g);
                float4 hD1 = zero4, hD2 = zero4, hD3 = zero4, hD4 = zero4;
                if (h.ok) {
                    float nvx[4], nvz[4];
                    adj_v_update(p, h.gg, E1, E2, E3, E4, h.r, 4 * (h.g + 1), hvx, hvz, nvx, nvz);
                    stage_D(p, s, h.j, h.gg, make_float4(nvx[0], nvx[1], nvx[2], nvx[3]),
                            make_float4(nvz[0], nvz[1], nvz[2], nvz[3]), hbx, hbz, false, hD1, hD2, hD3, hD4);
                }
                const int x = 4 * (h.g + 1);
                sts4(D1 + h.r * FSW + x, hD1); sts4(D2 + h.r * FSW + x, hD2); sts4(D3 + h.r * FSW + x, hD3); sts4(D4 + h.r * FSW + x, hD4);
            }
        }
        __syncthreads();
        // ---- sigma_bar' of the tile -----------------------------------------------------------------------------------
        {
            const int tq = f_opaque(t);
            const FOwn o = f_own(p, tq, tile_j, tile_g);
            if (o.ok) {
                const int r = o.orow + 2, cb = 4 * (o.ogrp + 2);
                float nxx[4], nzz[4], nxz[4];
                // the own adjoint stresses once more (cache hits): cheaper than registers held across the step;
                // V^T updates the stored szz_bar(0,.) as it is (S^T discarded it)
                const float4 sxx0 = ld4(fin + F_SXX * fs + o.oo), szz0 = ld4(fin + F_SZZ * fs + o.oo);
                const float4 sxz0 = ld4(fin + F_SXZ * fs + o.oo);
                const Row8 x1 = row8(D1 + r * FSW, cb), x3 = row8(D3 + r * FSW, cb);
                const float4 z2a = lds4(D2 + (r - 1) * FSW + cb), z2b = lds4(D2 + r * FSW + cb);
                const float4 z2c = lds4(D2 + (r + 1) * FSW + cb), z2d = lds4(D2 + (r + 2) * FSW + cb);
                const float4 z4a = lds4(D4 + (r - 2) * FSW + cb), z4b = lds4(D4 + (r - 1) * FSW + cb);
                const float4 z4c = lds4(D4 + r * FSW + cb), z4d = lds4(D4 + (r + 1) * FSW + cb);
                float4 m12 = zero4, m13 = zero4, m32 = zero4;
                if (p.fsurf && o.oj < 2) { m12 = lds4(D2 + 2 * FSW + cb); m13 = lds4(D2 + 3 * FSW + cb); m32 = lds4(D4 + 2 * FSW + cb); }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float dx1 = dbw(K, x1.v[c], x1.v[c + 1], x1.v[c + 2], x1.v[c + 3]);
                    const float dz2 = dfw(K, comp(z2a, c), comp(z2b, c), comp(z2c, c), comp(z2d, c));
                    const float dx3 = dfw(K, x3.v[c + 1], x3.v[c + 2], x3.v[c + 3], x3.v[c + 4]);
                    const float dz4 = dbw(K, comp(z4a, c), comp(z4b, c), comp(z4c, c), comp(z4d, c));
                    nxx[c] = comp(sxx0, c) - dx1;
                    nxz[c] = comp(sxz0, c) - (dz2 + dx3);
                    nzz[c] = comp(szz0, c) - dz4;
                    if (p.fsurf && o.oj < 2) {
                        // transposed odd mirroring (tile_j == 0 here: row 2 of the planes is grid row 0)
                        if (o.oj == 0) nxz[c] = nxz[c] + fmaf(K.c1, comp(m12, c), K.c2 * comp(m13, c));
                        else { nxz[c] = nxz[c] + K.c2 * comp(m12, c); nzz[c] = nzz[c] + K.c2 * comp(m32, c); }
                    }
                    if (4 * o.og + c >= p.nx) { nxx[c] = 0.f; nxz[c] = 0.f; nzz[c] = 0.f; }
                }
                st4(fout + F_SXX * fs + o.oo, make_float4(nxx[0], nxx[1], nxx[2], nxx[3]));
                st4(fout + F_SZZ * fs + o.oo, make_float4(nzz[0], nzz[1], nzz[2], nzz[3]));
                st4(fout + F_SXZ * fs + o.oo, make_float4(nxz[0], nxz[1], nxz[2], nxz[3]));
            }
        }
        __syncthreads();          // the planes are reused by the next shot
    }
    {
        const FOwn o = f_own(p, f_opaque(t), tile_j, tile_g);
        const long long acc_base = (long long)(p.s0 / p.gs + bz) * 5 * p.splane + snap_cell(p, o.oj, o.og);
        if (o.ok) {
#pragma unroll
            for (int k = 0; k < 5; ++k) st4(p.acc + acc_base + (long long)k * p.splane, acc[k]);
        }
    }
}
