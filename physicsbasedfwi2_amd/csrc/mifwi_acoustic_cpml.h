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
};

__host__ __device__ inline long long pml_persist_per_shot(int W, int n0, int gp) { return 2LL * (2LL * W * gp) + 2LL * (2LL * W * n0); }
__host__ __device__ inline long long pml_scratch_per_shot(int W, int n0, int gp)
{
    return pml_persist_per_shot(W, n0, gp) + 2LL * (W + 2) * gp + 2LL * (W + 2) * n0;
}

// strip-shaped array of axis 0 / axis 1, zero outside the strip (and outside the grid)
__device__ __forceinline__ float pml_get0(const AcPml &p, const float *a, int i0, int i1)
{
    if (i0 < 0 || i0 >= p.n0) return 0.f;
    if (i0 < p.W) return a[(long long)i0 * p.gp + i1];
    if (i0 >= p.n0 - p.W) return a[(long long)(p.W + i0 - (p.n0 - p.W)) * p.gp + i1];
    return 0.f;
}
__device__ __forceinline__ float pml_get1(const AcPml &p, const float *a, int i0, int i1)
{
    if (i1 < 0 || i1 >= p.n1) return 0.f;
    if (i1 < p.W) return a[((long long)i0 * 2) * p.W + i1];
    if (i1 >= p.n1 - p.W) return a[((long long)i0 * 2 + 1) * p.W + i1 - (p.n1 - p.W)];
    return 0.f;
}
__device__ __forceinline__ float pml_d1(float m2, float m1, float p1, float p2) { return fmaf(CF1, p1 - m1, CF2 * (p2 - m2)); }
__device__ __forceinline__ float pml_d2(float m2, float m1, float c, float p1, float p2)
{
    return fmaf(K1, m1 + p1, fmaf(K2, m2 + p2, K0 * c));
}

// One thread per region cell: ids [0, 2 (W+2) n1) walk region 0 (side, l, i1), the rest region 1 (i0, side, l).
struct PmlCell {
    int axis, i0, i1;            // axis < 0: no cell
    bool strip;
    long long sidx, ridx;        // index inside the strip / region array of the axis
};
__device__ __forceinline__ PmlCell pml_cell(const AcPml &p, long long id)
{
    PmlCell c;
    c.axis = -1; c.i0 = c.i1 = 0; c.strip = false; c.sidx = c.ridx = 0;
    const int W2 = p.W + 2;
    const long long n0c = 2LL * W2 * p.n1, n1c = 2LL * W2 * p.n0;
    if (id < n0c) {
        const int side = (int)(id / ((long long)W2 * p.n1));
        const long long rem = id - (long long)side * W2 * p.n1;
        const int l = (int)(rem / p.n1);
        c.axis = 0; c.i1 = (int)(rem - (long long)l * p.n1);
        c.i0 = side == 0 ? l : p.n0 - W2 + l;
        c.strip = side == 0 ? l < p.W : l >= 2;
        const int ls = side == 0 ? l : l - 2;
        c.sidx = ((long long)side * p.W + ls) * p.gp + c.i1;
        c.ridx = ((long long)side * W2 + l) * p.gp + c.i1;
    } else if (id < n0c + n1c) {
        const long long q = id - n0c;
        c.i0 = (int)(q / (2 * W2));
        const int rem = (int)(q - (long long)c.i0 * 2 * W2);
        const int side = rem / W2, l = rem - side * W2;
        c.axis = 1;
        c.i1 = side == 0 ? l : p.n1 - W2 + l;
        c.strip = side == 0 ? l < p.W : l >= 2;
        const int ls = side == 0 ? l : l - 2;
        c.sidx = ((long long)c.i0 * 2 + side) * p.W + ls;
        c.ridx = ((long long)c.i0 * 2 + side) * W2 + l;
    }
    return c;
}

#define PML_PROLOGUE                                                                              \
    const int s = p.shot0 + (int)blockIdx.y;                                                      \
    if (s >= p.nshot) return;                                                                     \
    const PmlCell c = pml_cell(p, (long long)blockIdx.x * blockDim.x + threadIdx.x);              \
    if (c.axis < 0) return;                                                                       \
    const float *u = cur + (long long)s * p.shot_stride + 2LL * p.pitch + 4;                      \
    const long long pt = p.pitch;                                                                 \
    const long long k = (long long)c.i0 * pt + c.i1;                                              \
    (void)u; (void)k; (void)pt

// forward 1: Psi_d = fma(b, Psi_d, a * D1_d u) on the strips
__global__ __launch_bounds__(kThreads) void ac_pml_fwd_psi(const AcPml p, const float *cur)
{
    PML_PROLOGUE;
    if (!c.strip) return;
    if (c.axis == 0) {
        float *A = p.A0 + (long long)s * p.s0;
        const float d = pml_d1(u[k - 2 * pt], u[k - pt], u[k + pt], u[k + 2 * pt]);
        A[c.sidx] = fmaf(p.ab0[p.n0 + c.i0], A[c.sidx], p.ab0[c.i0] * d);
    } else {
        float *A = p.A1 + (long long)s * p.s1;
        const float d = pml_d1(u[k - 2], u[k - 1], u[k + 1], u[k + 2]);
        A[c.sidx] = fmaf(p.ab1[p.gp + c.i1], A[c.sidx], p.ab1[c.i1] * d);
    }
}

// forward 2: Z_d = fma(b, Z_d, a * (D2_d u + D1_d Psi_d)) on the strips; e_d = D1_d Psi_d + Z_d on the regions
__global__ __launch_bounds__(kThreads) void ac_pml_fwd_zeta(const AcPml p, const float *cur)
{
    PML_PROLOGUE;
    if (c.axis == 0) {
        const float *A = p.A0 + (long long)s * p.s0;
        float *B = p.B0 + (long long)s * p.s0;
        const float dp = pml_d1(pml_get0(p, A, c.i0 - 2, c.i1), pml_get0(p, A, c.i0 - 1, c.i1),
                                pml_get0(p, A, c.i0 + 1, c.i1), pml_get0(p, A, c.i0 + 2, c.i1));
        float z = 0.f;
        if (c.strip) {
            const float d2 = pml_d2(u[k - 2 * pt], u[k - pt], u[k], u[k + pt], u[k + 2 * pt]);
            z = fmaf(p.ab0[p.n0 + c.i0], B[c.sidx], p.ab0[c.i0] * (d2 + dp));
            B[c.sidx] = z;
        }
        (p.e0 + (long long)s * p.r0)[c.ridx] = dp + z;
    } else {
        const float *A = p.A1 + (long long)s * p.s1;
        float *B = p.B1 + (long long)s * p.s1;
        const float dp = pml_d1(pml_get1(p, A, c.i0, c.i1 - 2), pml_get1(p, A, c.i0, c.i1 - 1),
                                pml_get1(p, A, c.i0, c.i1 + 1), pml_get1(p, A, c.i0, c.i1 + 2));
        float z = 0.f;
        if (c.strip) {
            const float d2 = pml_d2(u[k - 2], u[k - 1], u[k], u[k + 1], u[k + 2]);
            z = fmaf(p.ab1[p.gp + c.i1], B[c.sidx], p.ab1[c.i1] * (d2 + dp));
            B[c.sidx] = z;
        }
        (p.e1 + (long long)s * p.r1)[c.ridx] = dp + z;
    }
}

// adjoint 1 (w = z^{k+1} = cur): A = fma(c_d, w, Zb);  P = a A;  Zb = b A   on the strips
__global__ __launch_bounds__(kThreads) void ac_pml_adj_a(const AcPml p, const float *cur)
{
    PML_PROLOGUE;
    if (!c.strip) return;
    if (c.axis == 0) {
        float *B = p.B0 + (long long)s * p.s0;
        const float a = fmaf(p.c0, u[k], B[c.sidx]);
        (p.P0 + (long long)s * p.s0)[c.sidx] = p.ab0[c.i0] * a;
        B[c.sidx] = p.ab0[p.n0 + c.i0] * a;
    } else {
        float *B = p.B1 + (long long)s * p.s1;
        const float a = fmaf(p.c1, u[k], B[c.sidx]);
        (p.P1 + (long long)s * p.s1)[c.sidx] = p.ab1[c.i1] * a;
        B[c.sidx] = p.ab1[p.gp + c.i1] * a;
    }
}

// adjoint 2: T = Pb - D1_d(fma(c_d, w, P));  Q = a T;  Pb = b T   on the strips
__global__ __launch_bounds__(kThreads) void ac_pml_adj_b(const AcPml p, const float *cur)
{
    PML_PROLOGUE;
    if (!c.strip) return;
    if (c.axis == 0) {
        const float *P = p.P0 + (long long)s * p.s0;
        float *A = p.A0 + (long long)s * p.s0;
        auto V = [&](int o) { return fmaf(p.c0, u[k + o * pt], pml_get0(p, P, c.i0 + o, c.i1)); };
        const float t = A[c.sidx] - pml_d1(V(-2), V(-1), V(1), V(2));
        (p.Q0 + (long long)s * p.s0)[c.sidx] = p.ab0[c.i0] * t;
        A[c.sidx] = p.ab0[p.n0 + c.i0] * t;
    } else {
        const float *P = p.P1 + (long long)s * p.s1;
        float *A = p.A1 + (long long)s * p.s1;
        auto V = [&](int o) { return fmaf(p.c1, u[k + o], pml_get1(p, P, c.i0, c.i1 + o)); };
        const float t = A[c.sidx] - pml_d1(V(-2), V(-1), V(1), V(2));
        (p.Q1 + (long long)s * p.s1)[c.sidx] = p.ab1[c.i1] * t;
        A[c.sidx] = p.ab1[p.gp + c.i1] * t;
    }
}

// adjoint 3: e_d = D2_d P - D1_d Q on the regions
__global__ __launch_bounds__(kThreads) void ac_pml_adj_c(const AcPml p, const float *cur)
{
    PML_PROLOGUE;
    if (c.axis == 0) {
        const float *P = p.P0 + (long long)s * p.s0, *Q = p.Q0 + (long long)s * p.s0;
        auto gp_ = [&](int o) { return pml_get0(p, P, c.i0 + o, c.i1); };
        auto gq_ = [&](int o) { return pml_get0(p, Q, c.i0 + o, c.i1); };
        (p.e0 + (long long)s * p.r0)[c.ridx] =
            pml_d2(gp_(-2), gp_(-1), gp_(0), gp_(1), gp_(2)) - pml_d1(gq_(-2), gq_(-1), gq_(1), gq_(2));
    } else {
        const float *P = p.P1 + (long long)s * p.s1, *Q = p.Q1 + (long long)s * p.s1;
        auto gp_ = [&](int o) { return pml_get1(p, P, c.i0, c.i1 + o); };
        auto gq_ = [&](int o) { return pml_get1(p, Q, c.i0, c.i1 + o); };
        (p.e1 + (long long)s * p.r1)[c.ridx] =
            pml_d2(gp_(-2), gp_(-1), gp_(0), gp_(1), gp_(2)) - pml_d1(gq_(-2), gq_(-1), gq_(1), gq_(2));
    }
}
#undef PML_PROLOGUE
