// Second-order C-PML of the scalar scheme (one launch per step family): the memory variables of the layer live in
// strip-compact arrays, a few thin launches per step update them and leave the layer's contribution to the
// Laplacian ("e") in region-compact arrays that ac_step<..., PML = true> adds.  Same fmaf chains as
// oracle/acoustic_cpml.c (which states the scheme and its exact transposed adjoint), bit for bit.
//
//   strip d  = the W cells of the layer at either end of axis d;  region d = strip d and two cells beyond it
//   per shot: strip0 arrays [2 sides][W][gp], region0 [2][W+2][gp], strip1 [n0][2][W], region1 [n0][2][W+2]
//   forward state  Psi0, Z0, Psi1, Z1                (persistent: part of the checkpointed state)
//   adjoint state  Pb0,  Zb0, Pb1,  Zb1              (same slots)
//   scratch        P0, Q0, P1, Q1 (adjoint), e0, e1
// Included by mifwi_acoustic.hip inside its anonymous namespace (uses K0..K2, kThreads).
#pragma once

constexpr float CF1 = (float)(2.0 / 3.0);
constexpr float CF2 = (float)(-1.0 / 12.0);

struct AcPml {
    int W;                       // layer width in cells (0: no C-PML)
    int n0, n1, gp, pitch;
    long long shot_stride;       // floats per shot wavefield
    const float *ab0, *ab1;      // [2][n0], [2][gp]: a then b
    float *A0, *B0, *A1, *B1;    // persistent strip arrays (Psi / Z, or their adjoints), shot-major
    float *P0, *Q0, *P1, *Q1;    // adjoint scratch (strip-shaped)
    float *e0, *e1;              // region arrays
    long long s0, s1, r0, r1;    // per-shot sizes: 2 W gp, 2 W n0, 2 (W+2) gp, 2 (W+2) n0
    float c0, c1;
    int shot0;                   // first shot of the launch (blockIdx.y counts from it)
    int nshot;
    unsigned mg_ng, mg_w2;       // floor(2^32 / (gp / 4)), floor(2^32 / (2 (W + 2))): pml_div
    // single-launch kernels only: the arrays of `lds` (PML_* bits) live in the workgroup's LDS, where the pointer is the
    // slab's own part of the array and f0s/f0r/f1s/f1r the element of the shot's array that part starts at
    unsigned lds;
    long long f0s, f0r, f1s, f1r;
};

// Which of the layer's arrays of ONE slab live in LDS inside the single-launch kernels (the rest stay in global memory,
// i.e. the XCD's L2): as many as fit behind the slab's two field planes, the exchanged ones first (a neighbour's psi',
// P, Q and the layer's term e are read by other threads: through L2 they cost the CU's 64 B/clk L1 fill rate, which is
// what bounds the layer's phases), the pointwise memory variables last.  Same function on the host (LDS size of the
// launch) and on the device (carving): bit k set = array k of kPmlOrder is in LDS.
enum { PML_A0 = 1, PML_B0 = 2, PML_P0 = 4, PML_Q0 = 8, PML_E0 = 16, PML_A1 = 32, PML_B1 = 64, PML_P1 = 128, PML_Q1 = 256, PML_E1 = 512 };
// Element offset of shot s inside array `bit`: s x the per-shot size in global memory; minus the first element the slab
// holds when the array lives in LDS.  Always added to the cell's index BEFORE the pointer (the sum is a valid index of
// the buffer; a pointer biased below an LDS buffer leaves the LDS aperture).
__device__ __forceinline__ long long pml_off(const AcPml &p, const unsigned bit, const int s)
{
    const bool ax0 = bit & (PML_A0 | PML_B0 | PML_P0 | PML_Q0 | PML_E0), reg = bit & (PML_E0 | PML_E1);
    const long long stride = ax0 ? (reg ? p.r0 : p.s0) : (reg ? p.r1 : p.s1);
    const long long first = ax0 ? (reg ? p.f0r : p.f0s) : (reg ? p.f1r : p.f1s);
    return (p.lds & bit) ? -first : (long long)s * stride;
}
__host__ __device__ inline long long pml_lds_floats(int bit, int rows, int W, int gp)
{
    const long long W2 = W + 2;
    long long n = 0;
    switch (bit) {
        case PML_A0: case PML_B0: case PML_P0: case PML_Q0: n = (long long)W * gp; break;
        case PML_E0: n = W2 * gp; break;
        case PML_A1: case PML_B1: case PML_P1: case PML_Q1: n = 2LL * W * rows; break;
        case PML_E1: n = 2LL * W2 * rows; break;
    }
    return (n + 3) & ~3LL;
}
// own = the edge slab runs the layer of axis 0 in the own-group form below (pml_own_*): its arrays are not placed here
__host__ __device__ inline unsigned pml_place(bool adjoint, bool edge, int rows, int W, int gp, long long free_floats, long long *used,
                                              bool own = false)
{
    const int fwd[6] = {PML_A1, PML_E1, PML_A0, PML_E0, PML_B1, PML_B0};
    const int adj[10] = {PML_P1, PML_Q1, PML_E1, PML_P0, PML_Q0, PML_E0, PML_A1, PML_B1, PML_A0, PML_B0};
    unsigned mask = 0;
    long long u = 0;
    for (int k = 0; k < (adjoint ? 10 : 6); ++k) {
        const int bit = adjoint ? adj[k] : fwd[k];
        const bool axis0 = bit & (PML_A0 | PML_B0 | PML_P0 | PML_Q0 | PML_E0);
        if (axis0 && (!edge || own)) continue;
        const long long n = pml_lds_floats(bit, rows, W, gp);
        if (u + n <= free_floats) { mask |= (unsigned)bit; u += n; }
    }
    if (used) *used = u;
    return mask;
}

__host__ __device__ inline long long pml_persist_per_shot(int W, int n0, int gp) { return 2LL * (2LL * W * gp) + 2LL * (2LL * W * n0); }
__host__ __device__ inline long long pml_scratch_per_shot(int W, int n0, int gp)
{
    return pml_persist_per_shot(W, n0, gp) + 2LL * (W + 2) * gp + 2LL * (W + 2) * n0;
}

// x / d for d < 2^31 with mg = floor(2^32 / d): a multiply-high and one correction instead of a division sequence (the
// thread maps below divide once per cell, phase and step)
__host__ __device__ inline unsigned pml_magic(unsigned d) { return d <= 1 ? 0xffffffffu : (unsigned)(0x100000000ULL / d); }
__device__ __forceinline__ unsigned pml_div(unsigned x, unsigned d, unsigned mg)
{
    unsigned q = __umulhi(x, mg);
    if (x - q * d >= d) ++q;
    return q;
}

// strip-shaped array of axis 0 / axis 1, zero outside the strip (and outside the grid)
// strip-shaped arrays read at a cell that may lie outside the strip (zero there).  Branch-free: the load goes to a
// clamped, always valid index and the result is selected afterwards, so that the four or five neighbour reads of a
// cell are independent loads in flight together (as separate branches they were serialised round trips: 3 000 clocks
// per cell in the single-launch kernel's layer phases).
__device__ __forceinline__ float pml_get1(const AcPml &p, const float *a, long long off, int i0, int i1)
{
    // (unsigned compares: one each for "inside the low strip" / "inside the high strip")
    const bool lo = (unsigned)i1 < (unsigned)p.W, hi = (unsigned)(i1 - (p.n1 - p.W)) < (unsigned)p.W;
    const int l = (lo || hi) ? (lo ? i1 : p.W + i1 - (p.n1 - p.W)) : 0;         // outside: any valid element of the row
    const float v = a[off + (i0 * 2 * p.W + l)];
    return (lo || hi) ? v : 0.f;          // (the load above is unconditional)
}
__device__ __forceinline__ float pml_d1(float m2, float m1, float p1, float p2) { return fmaf(CF1, p1 - m1, CF2 * (p2 - m2)); }
__device__ __forceinline__ float pml_d2(float m2, float m1, float c, float p1, float p2)
{
    return fmaf(K1, m1 + p1, fmaf(K2, m2 + p2, K0 * c));
}

// Thread maps.  blockIdx.z = axis, blockIdx.y = shot.
//   axis 0 (the layer's rows, whole rows of the grid): one thread per GROUP of four cells, id = (side (W+2) + l) ng + g
//           - 16-byte accesses along i1, four cells of arithmetic per thread;
//   axis 1 (the layer's columns, two short runs per row): one thread per cell, id = i0 2 (W+2) + side (W+2) + l.
struct PmlCell {
    int i0, i1;                  // axis 0: i1 = first cell of the group
    bool ok, strip;
    int sidx, ridx;              // index inside the strip / region array of the axis
};
__device__ __forceinline__ PmlCell pml_cell(const AcPml &p, unsigned id, int axis)
{
    PmlCell c;
    c.ok = false; c.i0 = c.i1 = 0; c.strip = false; c.sidx = c.ridx = 0;
    const unsigned W2 = (unsigned)p.W + 2u;
    if (axis == 0) {
        const unsigned ng = (unsigned)p.gp / 4u;
        if (id >= 2u * W2 * ng) return c;
        const unsigned row = pml_div(id, ng, p.mg_ng);               // side * W2 + l
        const unsigned side = row >= W2 ? 1u : 0u, l = row - side * W2;
        c.ok = true; c.i1 = 4 * (int)(id - row * ng);
        c.i0 = side == 0 ? (int)l : p.n0 - (int)W2 + (int)l;
        c.strip = side == 0 ? l < (unsigned)p.W : l >= 2u;
        const int ls = side == 0 ? (int)l : (int)l - 2;
        c.sidx = ((int)side * p.W + ls) * p.gp + c.i1;
        c.ridx = (int)row * p.gp + c.i1;
    } else {
        if (id >= 2u * W2 * (unsigned)p.n0) return c;
        const unsigned i0 = pml_div(id, 2u * W2, p.mg_w2), rem = id - i0 * 2u * W2;
        const unsigned side = rem >= W2 ? 1u : 0u, l = rem - side * W2;
        c.ok = true; c.i0 = (int)i0;
        c.i1 = side == 0 ? (int)l : p.n1 - (int)W2 + (int)l;
        c.strip = side == 0 ? l < (unsigned)p.W : l >= 2u;
        const int ls = side == 0 ? (int)l : (int)l - 2;
        c.sidx = ((int)i0 * 2 + (int)side) * p.W + ls;
        c.ridx = (int)id;
    }
    return c;
}
__device__ __forceinline__ float4 pml_ld4(const float *q) { return *reinterpret_cast<const float4 *>(q); }
__device__ __forceinline__ void pml_st4(float *q, const float (&v)[4]) { *reinterpret_cast<float4 *>(q) = make_float4(v[0], v[1], v[2], v[3]); }
// four cells of a strip-shaped axis-0 array at row i0 (zero outside the strip / the grid); branch-free as pml_get1
__device__ __forceinline__ float4 pml_get0v(const AcPml &p, const float *a, long long off, int i0, int i1)
{
    const bool lo = (unsigned)i0 < (unsigned)p.W, hi = (unsigned)(i0 - (p.n0 - p.W)) < (unsigned)p.W;
    // outside the strips: the first row of the strip on the row's own side (the part a slab of a single-launch kernel
    // holds in LDS) - any valid element, the value is discarded
    const int l = (lo || hi) ? (lo ? i0 : p.W + i0 - (p.n0 - p.W)) : (2 * i0 >= p.n0 ? p.W : 0);
    const float4 v = pml_ld4(a + (off + (l * p.gp + i1)));
    const bool ok = lo || hi;
    return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
}

// forward 1: Psi_d = fma(b, Psi_d, a * D1_d u) on the strips
__device__ __forceinline__ void ac_pml_fwd_psi_cell(const AcPml &p, const int s, const int axis, const PmlCell &c, const float *u, const int pt)
{
    const int k = c.i0 * pt + c.i1;
    (void)k;
    if (!c.strip) return;
    if (axis == 0) {
        float *A = p.A0 + (pml_off(p, PML_A0, s) + c.sidx);
        const float4 m2 = pml_ld4(u + k - 2 * pt), m1 = pml_ld4(u + k - pt), p1 = pml_ld4(u + k + pt), p2 = pml_ld4(u + k + 2 * pt);
        const float4 a4 = pml_ld4(A);
        const float a = p.ab0[c.i0], b = p.ab0[p.n0 + c.i0];
        float o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = fmaf(b, comp(a4, q), a * pml_d1(comp(m2, q), comp(m1, q), comp(p1, q), comp(p2, q)));
        pml_st4(A, o);
    } else {
        float *A = p.A1 + (pml_off(p, PML_A1, s) + c.sidx);
        const float d = pml_d1(u[k - 2], u[k - 1], u[k + 1], u[k + 2]);
        *A = fmaf(p.ab1[p.gp + c.i1], *A, p.ab1[c.i1] * d);
    }
}

// forward 2: Z_d = fma(b, Z_d, a * (D2_d u + D1_d Psi_d)) on the strips; e_d = D1_d Psi_d + Z_d on the regions
__device__ __forceinline__ void ac_pml_fwd_zeta_cell(const AcPml &p, const int s, const int axis, const PmlCell &c, const float *u, const int pt)
{
    const int k = c.i0 * pt + c.i1;
    // every load is issued before the first use (and unconditionally, at a clamped index): one memory round trip per
    // cell, not one per operand group
    if (axis == 0) {
        const long long oA = pml_off(p, PML_A0, s);
        float *B = p.B0 + (pml_off(p, PML_B0, s) + (c.strip ? (long long)c.sidx : p.f0s));
        const float4 am2 = pml_get0v(p, p.A0, oA, c.i0 - 2, c.i1), am1 = pml_get0v(p, p.A0, oA, c.i0 - 1, c.i1);
        const float4 ap1 = pml_get0v(p, p.A0, oA, c.i0 + 1, c.i1), ap2 = pml_get0v(p, p.A0, oA, c.i0 + 2, c.i1);
        const float4 b4 = pml_ld4(B);
        const float a = p.ab0[c.i0], b = p.ab0[p.n0 + c.i0];
        const float4 m2 = pml_ld4(u + k - 2 * pt), m1 = pml_ld4(u + k - pt), uc = pml_ld4(u + k);
        const float4 p1 = pml_ld4(u + k + pt), p2 = pml_ld4(u + k + 2 * pt);
        float dp[4], z[4], e[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            dp[q] = pml_d1(comp(am2, q), comp(am1, q), comp(ap1, q), comp(ap2, q));
            z[q] = fmaf(b, comp(b4, q), a * (pml_d2(comp(m2, q), comp(m1, q), comp(uc, q), comp(p1, q), comp(p2, q)) + dp[q]));
            if (!c.strip) z[q] = 0.f;
            e[q] = dp[q] + z[q];
        }
        if (c.strip) pml_st4(B, z);
        pml_st4(p.e0 + (pml_off(p, PML_E0, s) + c.ridx), e);
    } else {
        const long long oA = pml_off(p, PML_A1, s);
        float *B = p.B1 + (pml_off(p, PML_B1, s) + (c.strip ? c.sidx : c.i0 * 2 * p.W));
        const float a0 = pml_get1(p, p.A1, oA, c.i0, c.i1 - 2), a1 = pml_get1(p, p.A1, oA, c.i0, c.i1 - 1);
        const float a2 = pml_get1(p, p.A1, oA, c.i0, c.i1 + 1), a3 = pml_get1(p, p.A1, oA, c.i0, c.i1 + 2);
        const float bz = *B;
        const float a = p.ab1[c.i1], b = p.ab1[p.gp + c.i1];
        const float dp = pml_d1(a0, a1, a2, a3);
        const float d2 = pml_d2(u[k - 2], u[k - 1], u[k], u[k + 1], u[k + 2]);
        float z = fmaf(b, bz, a * (d2 + dp));
        if (!c.strip) z = 0.f;
        if (c.strip) *B = z;
        p.e1[pml_off(p, PML_E1, s) + c.ridx] = dp + z;
    }
}

// adjoint 1 (w = z^{k+1} = cur): A = fma(c_d, w, Zb);  P = a A;  Zb = b A   on the strips
__device__ __forceinline__ void ac_pml_adj_a_cell(const AcPml &p, const int s, const int axis, const PmlCell &c, const float *u, const int pt)
{
    const int k = c.i0 * pt + c.i1;
    (void)k;
    if (!c.strip) return;
    if (axis == 0) {
        float *B = p.B0 + (pml_off(p, PML_B0, s) + c.sidx);
        const float4 w4 = pml_ld4(u + k), b4 = pml_ld4(B);
        const float a = p.ab0[c.i0], b = p.ab0[p.n0 + c.i0];
        float P[4], Z[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float t = fmaf(p.c0, comp(w4, q), comp(b4, q));
            P[q] = a * t; Z[q] = b * t;
        }
        pml_st4(p.P0 + (pml_off(p, PML_P0, s) + c.sidx), P);
        pml_st4(B, Z);
    } else {
        float *B = p.B1 + (pml_off(p, PML_B1, s) + c.sidx);
        const float a = fmaf(p.c1, u[k], *B);
        p.P1[pml_off(p, PML_P1, s) + c.sidx] = p.ab1[c.i1] * a;
        *B = p.ab1[p.gp + c.i1] * a;
    }
}

// adjoint 2: T = Pb - D1_d(fma(c_d, w, P));  Q = a T;  Pb = b T   on the strips
__device__ __forceinline__ void ac_pml_adj_b_cell(const AcPml &p, const int s, const int axis, const PmlCell &c, const float *u, const int pt)
{
    const int k = c.i0 * pt + c.i1;
    (void)k;
    if (!c.strip) return;
    if (axis == 0) {
        const long long oP = pml_off(p, PML_P0, s);
        float *A = p.A0 + (pml_off(p, PML_A0, s) + c.sidx);
        float4 w[4], pp[4];
        const int off[4] = {-2, -1, 1, 2};
#pragma unroll
        for (int o = 0; o < 4; ++o) { w[o] = pml_ld4(u + k + off[o] * pt); pp[o] = pml_get0v(p, p.P0, oP, c.i0 + off[o], c.i1); }
        const float4 a4 = pml_ld4(A);
        const float a = p.ab0[c.i0], b = p.ab0[p.n0 + c.i0];
        float Q[4], T[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float v0 = fmaf(p.c0, comp(w[0], q), comp(pp[0], q)), v1 = fmaf(p.c0, comp(w[1], q), comp(pp[1], q));
            const float v2 = fmaf(p.c0, comp(w[2], q), comp(pp[2], q)), v3 = fmaf(p.c0, comp(w[3], q), comp(pp[3], q));
            const float t = comp(a4, q) - pml_d1(v0, v1, v2, v3);
            Q[q] = a * t; T[q] = b * t;
        }
        pml_st4(p.Q0 + (pml_off(p, PML_Q0, s) + c.sidx), Q);
        pml_st4(A, T);
    } else {
        const long long oP = pml_off(p, PML_P1, s);
        float *A = p.A1 + (pml_off(p, PML_A1, s) + c.sidx);
        auto V = [&](int o) { return fmaf(p.c1, u[k + o], pml_get1(p, p.P1, oP, c.i0, c.i1 + o)); };
        const float t = *A - pml_d1(V(-2), V(-1), V(1), V(2));
        p.Q1[pml_off(p, PML_Q1, s) + c.sidx] = p.ab1[c.i1] * t;
        *A = p.ab1[p.gp + c.i1] * t;
    }
}

// adjoint 3: e_d = D2_d P - D1_d Q on the regions
__device__ __forceinline__ void ac_pml_adj_c_cell(const AcPml &p, const int s, const int axis, const PmlCell &c, const float *u, const int pt)
{
    const int k = c.i0 * pt + c.i1;
    (void)k;
    if (axis == 0) {
        const long long oP = pml_off(p, PML_P0, s), oQ = pml_off(p, PML_Q0, s);
        float4 pv[5], qv[5];
#pragma unroll
        for (int o = 0; o < 5; ++o) { pv[o] = pml_get0v(p, p.P0, oP, c.i0 + o - 2, c.i1); qv[o] = pml_get0v(p, p.Q0, oQ, c.i0 + o - 2, c.i1); }
        float e[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            e[q] = pml_d2(comp(pv[0], q), comp(pv[1], q), comp(pv[2], q), comp(pv[3], q), comp(pv[4], q)) -
                   pml_d1(comp(qv[0], q), comp(qv[1], q), comp(qv[3], q), comp(qv[4], q));
        pml_st4(p.e0 + (pml_off(p, PML_E0, s) + c.ridx), e);
    } else {
        const long long oP = pml_off(p, PML_P1, s), oQ = pml_off(p, PML_Q1, s);
        auto gp_ = [&](int o) { return pml_get1(p, p.P1, oP, c.i0, c.i1 + o); };
        auto gq_ = [&](int o) { return pml_get1(p, p.Q1, oQ, c.i0, c.i1 + o); };
        p.e1[pml_off(p, PML_E1, s) + c.ridx] =
            pml_d2(gp_(-2), gp_(-1), gp_(0), gp_(1), gp_(2)) - pml_d1(gq_(-2), gq_(-1), gq_(1), gq_(2));
    }
}

// does group g of row j touch the layer's regions at all?  (the single-launch kernels skip pml_term for the others)
__device__ __forceinline__ bool pml_layer_group(const AcPml &m, int j, int g)
{
    const int W2 = m.W + 2;
    return j < W2 || j >= m.n0 - W2 || 4 * g < W2 || 4 * g + 3 >= m.n1 - W2;
}
// The layer's term of the four cells of group g in row j, read from the region arrays of shot s (what ac_step<PML> and
// ac_cluster<PML> add to their Laplacian): forward fma(c0, e0, c1 e1), adjoint e0 + e1.  Branch-free loads.
// in1 = false: the caller knows that none of the group's cells lies in the region of axis 1 (only the two short runs at
// the ends of a row do): e1 is an exact zero there and is not read
// (e0own != nullptr: the caller holds e0 of the four cells - the own-group form of the edge slabs, where every row lies
// in the region of axis 0 - and the region array of axis 0 is not read)
__device__ __forceinline__ float4 pml_term(const AcPml &m, int s, int j, int g, bool adjoint, bool in1 = true, const float4 *e0own = nullptr)
{
    const int W2 = m.W + 2;
    const bool rlo = j < W2, rhi = j >= m.n0 - W2;
    int l0 = rlo ? j : W2 + j - (m.n0 - W2);
    l0 = l0 < 0 ? 0 : (l0 > 2 * W2 - 1 ? 2 * W2 - 1 : l0);
    const float4 e0v = e0own ? *e0own : pml_ld4(m.e0 + (pml_off(m, PML_E0, s) + (l0 * m.gp + 4 * g)));
    const bool ok0 = e0own ? true : (rlo || rhi);
    float e1v[4] = {0.f, 0.f, 0.f, 0.f};
    if (in1) {
        const float *pe1 = m.e1 + (pml_off(m, PML_E1, s) + j * 2 * W2);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int i1 = 4 * g + c;
            const bool clo = i1 < W2, chi = i1 >= m.n1 - W2 && i1 < m.n1;
            int l1 = clo ? i1 : W2 + i1 - (m.n1 - W2);
            l1 = l1 < 0 ? 0 : (l1 > 2 * W2 - 1 ? 2 * W2 - 1 : l1);
            const float v1 = pe1[l1];
            e1v[c] = (clo || chi) ? v1 : 0.f;
        }
    }
    float ev[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float e0 = ok0 ? comp(e0v, c) : 0.f;
        ev[c] = adjoint ? e0 + e1v[c] : fmaf(m.c0, e0, m.c1 * e1v[c]);
    }
    return make_float4(ev[0], ev[1], ev[2], ev[3]);
}

// ---- the own-group form of the edge slabs (single-launch kernels) -----------------------------------------------------
// An edge slab holds exactly the W + 2 rows of its end of the grid: every group a thread updates there is a region cell of
// axis 0, so the thread that owns a group runs the layer's recursions of axis 0 on it - no thread map, no index
// arithmetic.  The exchanged variables (Psi forward; P, Q adjoint) live in LDS planes shaped like the slab's field planes
// ([R + 4][PL], own row l at plane row l + 2, zero outside the strip: halo and beyond-the-strip reads need no clamp and no
// select), so a neighbour's value is the own address +- k PL.  The pointwise memory variables (Z forward; Zb, Pb adjoint)
// stay in the global state arrays.  e0 is computed where it is used, inside the field update of the same thread.  Same
// operations in the same order as the cell functions above: the same bits.
//   top = the slab at row 0 (strip = own rows l < W), else the slab at the last row (strip = l >= 2);  j = the grid row,
//   u / P / Q / Psi = the plane's element of the group's first cell
__host__ __device__ inline long long pml_own_floats(bool adjoint, int rows, int PL) { return (adjoint ? 2LL : 1LL) * (rows + 4) * PL; }
__device__ __forceinline__ bool pml_own_strip(const AcPml &m, bool top, int l) { return top ? l < m.W : l >= 2; }
__device__ __forceinline__ long long pml_own_idx(const AcPml &m, int s, bool top, int l, int g)
{
    return (long long)s * m.s0 + ((top ? l : m.W + l - 2) * m.gp + 4 * g);
}
__device__ __forceinline__ void pml_own_fwd_psi(const AcPml &m, bool top, int l, int j, const float *u, float *Psi, int PL)
{
    if (!pml_own_strip(m, top, l)) return;
    const float4 m2 = pml_ld4(u - 2 * PL), m1 = pml_ld4(u - PL), p1 = pml_ld4(u + PL), p2 = pml_ld4(u + 2 * PL);
    const float4 a4 = pml_ld4(Psi);
    const float a = m.ab0[j], b = m.ab0[m.n0 + j];
    float o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = fmaf(b, comp(a4, q), a * pml_d1(comp(m2, q), comp(m1, q), comp(p1, q), comp(p2, q)));
    pml_st4(Psi, o);
}
// forward, inside the update: Z on the strip, e0 = D1 Psi + Z on the region
__device__ __forceinline__ float4 pml_own_fwd_e0(const AcPml &m, int s, bool top, int l, int j, int g, const float *u, const float *Psi, int PL)
{
    const bool strip = pml_own_strip(m, top, l);
    const float4 am2 = pml_ld4(Psi - 2 * PL), am1 = pml_ld4(Psi - PL), ap1 = pml_ld4(Psi + PL), ap2 = pml_ld4(Psi + 2 * PL);
    float dp[4], z[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) dp[q] = pml_d1(comp(am2, q), comp(am1, q), comp(ap1, q), comp(ap2, q));
    if (strip) {
        float *B = m.B0 + pml_own_idx(m, s, top, l, g);
        const float4 b4 = pml_ld4(B);
        const float a = m.ab0[j], b = m.ab0[m.n0 + j];
        const float4 m2 = pml_ld4(u - 2 * PL), m1 = pml_ld4(u - PL), uc = pml_ld4(u), p1 = pml_ld4(u + PL), p2 = pml_ld4(u + 2 * PL);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            z[q] = fmaf(b, comp(b4, q), a * (pml_d2(comp(m2, q), comp(m1, q), comp(uc, q), comp(p1, q), comp(p2, q)) + dp[q]));
        pml_st4(B, z);
    }
    return make_float4(dp[0] + z[0], dp[1] + z[1], dp[2] + z[2], dp[3] + z[3]);
}
__device__ __forceinline__ void pml_own_adj_a(const AcPml &m, int s, bool top, int l, int j, int g, const float *u, float *P)
{
    if (!pml_own_strip(m, top, l)) return;
    float *B = m.B0 + pml_own_idx(m, s, top, l, g);
    const float4 w4 = pml_ld4(u), b4 = pml_ld4(B);
    const float a = m.ab0[j], b = m.ab0[m.n0 + j];
    float Pv[4], Z[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float t = fmaf(m.c0, comp(w4, q), comp(b4, q));
        Pv[q] = a * t; Z[q] = b * t;
    }
    pml_st4(P, Pv);
    pml_st4(B, Z);
}
__device__ __forceinline__ void pml_own_adj_b(const AcPml &m, int s, bool top, int l, int j, int g, const float *u, const float *P, float *Q, int PL)
{
    if (!pml_own_strip(m, top, l)) return;
    float *A = m.A0 + pml_own_idx(m, s, top, l, g);
    float4 w[4], pp[4];
    const int off[4] = {-2, -1, 1, 2};
#pragma unroll
    for (int o = 0; o < 4; ++o) { w[o] = pml_ld4(u + off[o] * PL); pp[o] = pml_ld4(P + off[o] * PL); }
    const float4 a4 = pml_ld4(A);
    const float a = m.ab0[j], b = m.ab0[m.n0 + j];
    float Qv[4], T[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float v0 = fmaf(m.c0, comp(w[0], q), comp(pp[0], q)), v1 = fmaf(m.c0, comp(w[1], q), comp(pp[1], q));
        const float v2 = fmaf(m.c0, comp(w[2], q), comp(pp[2], q)), v3 = fmaf(m.c0, comp(w[3], q), comp(pp[3], q));
        const float t = comp(a4, q) - pml_d1(v0, v1, v2, v3);
        Qv[q] = a * t; T[q] = b * t;
    }
    pml_st4(Q, Qv);
    pml_st4(A, T);
}
// adjoint, inside the update: e0 = D2 P - D1 Q on the region
__device__ __forceinline__ float4 pml_own_adj_e0(const float *P, const float *Q, int PL)
{
    float4 pv[5], qv[5];
#pragma unroll
    for (int o = 0; o < 5; ++o) { pv[o] = pml_ld4(P + (o - 2) * PL); if (o != 2) qv[o] = pml_ld4(Q + (o - 2) * PL); }
    float e[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
        e[q] = pml_d2(comp(pv[0], q), comp(pv[1], q), comp(pv[2], q), comp(pv[3], q), comp(pv[4], q)) -
               pml_d1(comp(qv[0], q), comp(qv[1], q), comp(qv[3], q), comp(qv[4], q));
    return make_float4(e[0], e[1], e[2], e[3]);
}

// ---- the thin launches of the one-launch-per-step family: one thread per cell (group), u = the global wavefield -------
#define PML_KERNEL(NAME)                                                                                      \
    __global__ __launch_bounds__(kThreads) void NAME(const AcPml p, const float *cur)                          \
    {                                                                                                          \
        const int s = p.shot0 + (int)blockIdx.y;                                                               \
        if (s >= p.nshot) return;                                                                              \
        const int axis = (int)blockIdx.z;                                                                      \
        const PmlCell c = pml_cell(p, blockIdx.x * blockDim.x + threadIdx.x, axis);                            \
        if (!c.ok) return;                                                                                     \
        NAME##_cell(p, s, axis, c, cur + (long long)s * p.shot_stride + 2LL * p.pitch + 4, p.pitch);           \
    }
PML_KERNEL(ac_pml_fwd_psi)
PML_KERNEL(ac_pml_fwd_zeta)
PML_KERNEL(ac_pml_adj_a)
PML_KERNEL(ac_pml_adj_b)
PML_KERNEL(ac_pml_adj_c)
#undef PML_KERNEL
