// Model parameterisations on the device (libmifwi, gfx950): (Vp, Vs, rho) -> the five staggered material planes of the
// elastic kernels, vp -> the scalar scheme's coefficient r on the padded grid, and their chain rules back.
// What DENISE does inside `set_model` / its model-averaging step before a forward run (the reference hands it Vp, Vs, rho:
// models/networks.py:7698-7712, 9790-9800) and undoes on the way back when `get_fwi_gradients` returns gradients with
// respect to Vp, Vs, rho (7802-7806): here one launch each way instead of the ~60 + ~100 elementwise launches of the torch
// expression (physicsbasedfwi2_amd/elastic.py:staggered_materials, which stays the definition: same operations, same
// order, same roundings - `-ffp-contract=off`, IEEE division).
//   mu = rho vs^2, lambda = rho vp^2 - 2 mu, s = dt / h
//   out0 = lambda s                       out1 = (lambda + 2 mu) s
//   out2 = s * harmonic mean of mu over (i,j), (i,j+1), (i+1,j), (i+1,j+1)   (0 where any of the four is 0: water)
//   out3 = s / (rho(i,j) + rho(i,j+1)) / 2   (vx node)        out4 = s / mean(rho(i,j), rho(i+1,j))   (vz node)
//   neighbours beyond the last row / column replicate the edge; free surface: row 0 of out0 -> 0 and of out1 ->
//   out1 - out0^2 / out1 (the effective moduli of the stress-imaging condition)
#include "mifwi_common.h"

namespace {

constexpr int kMT = 256;

__device__ __forceinline__ float mat_mu(const float *vs, const float *rho, long long k) { return rho[k] * vs[k] * vs[k]; }

// harmonic-mean node (a, b): the four mu, whether any is zero, and inv = sum 1 / mu (zeros replaced by 1, as the torch
// expression does before it discards the value)
struct MuNode { float m[4]; bool anyzero; float inv; };
__device__ __forceinline__ MuNode mat_node(const float *vs, const float *rho, int nz, int nx, int a, int b)
{
    const int a2 = min(a + 1, nz - 1), b2 = min(b + 1, nx - 1);
    MuNode n;
    n.m[0] = mat_mu(vs, rho, (long long)a * nx + b);
    n.m[1] = mat_mu(vs, rho, (long long)a * nx + b2);
    n.m[2] = mat_mu(vs, rho, (long long)a2 * nx + b);
    n.m[3] = mat_mu(vs, rho, (long long)a2 * nx + b2);
    n.anyzero = n.m[0] == 0.f || n.m[1] == 0.f || n.m[2] == 0.f || n.m[3] == 0.f;
    float inv = 1.0f / (n.m[0] == 0.f ? 1.0f : n.m[0]);
    inv = inv + 1.0f / (n.m[1] == 0.f ? 1.0f : n.m[1]);
    inv = inv + 1.0f / (n.m[2] == 0.f ? 1.0f : n.m[2]);
    inv = inv + 1.0f / (n.m[3] == 0.f ? 1.0f : n.m[3]);
    n.inv = inv;
    return n;
}

__global__ __launch_bounds__(kMT) void materials_fwd(const float *vp, const float *vs, const float *rho, float *out, int nz, int nx,
                                                     float s, int fsurf)
{
    const long long k = (long long)blockIdx.x * kMT + threadIdx.x, n = (long long)nz * nx;
    if (k >= n) return;
    const int i = (int)(k / nx), j = (int)(k - (long long)i * nx);
    const int i2 = min(i + 1, nz - 1), j2 = min(j + 1, nx - 1);
    const float r = rho[k];
    const float mu = r * vs[k] * vs[k];
    const float lam = r * vp[k] * vp[k] - 2.0f * mu;
    const float rx = 0.5f * (r + rho[(long long)i * nx + j2]);
    const float rz = 0.5f * (r + rho[(long long)i2 * nx + j]);
    const MuNode nd = mat_node(vs, rho, nz, nx, i, j);
    const float muxz = nd.anyzero ? 0.f : (1.0f / nd.inv) * 4.0f;
    float Ls = lam * s, Ms = (lam + 2.0f * mu) * s;
    if (fsurf) {
        const float top = i == 0 ? 1.0f : 0.f;
        Ms = Ms - top * (Ls * Ls / Ms);
        Ls = Ls * (1.0f - top);
    }
    out[k] = Ls;
    out[n + k] = Ms;
    out[2 * n + k] = muxz * s;
    out[3 * n + k] = (1.0f / rx) * s;                 // torch evaluates `s / tensor` as reciprocal(tensor) * s
    out[4 * n + k] = (1.0f / rz) * s;
}

// chain rule, one thread per model cell c = (i, j): gathers from the (at most four) output cells whose stencil touches c,
// in a fixed order (no atomics: the result is the same bits on every run)
__global__ __launch_bounds__(kMT) void materials_vjp(const float *vp, const float *vs, const float *rho, const float *g, float *gvp,
                                                     float *gvs, float *grho, int nz, int nx, float s, int fsurf)
{
    const long long k = (long long)blockIdx.x * kMT + threadIdx.x, n = (long long)nz * nx;
    if (k >= n) return;
    const int i = (int)(k / nx), j = (int)(k - (long long)i * nx);
    const float r = rho[k], p = vp[k], q = vs[k];
    // out0, out1 of the own cell -> lambda, mu
    float gL = g[k], gM = g[n + k];
    if (fsurf && i == 0) {
        const float mu = r * q * q, lam = r * p * p - 2.0f * mu;
        const float Ls = lam * s, Ms = (lam + 2.0f * mu) * s;
        const float ratio = Ls / Ms;
        gL = gM * (-2.0f * ratio);                    // out0 = 0 there; out1 = Ms - Ls^2 / Ms
        gM = gM * (1.0f + ratio * ratio);
    }
    const float dlam = (gL + gM) * s;
    float dmu = 2.0f * s * gM - 2.0f * dlam;          // through out1 directly and through lambda = rho vp^2 - 2 mu
    float drho = 0.f;
    // out2: every node (a, b) in {i-1, i} x {j-1, j} whose k-th corner is c (edge replication can make it several corners)
    for (int da = 1; da >= 0; --da)
        for (int db = 1; db >= 0; --db) {
            const int a = i - da, b = j - db;
            if (a < 0 || b < 0) continue;
            const int a2 = min(a + 1, nz - 1), b2 = min(b + 1, nx - 1);
            const bool c0 = a == i && b == j, c1 = a == i && b2 == j, c2 = a2 == i && b == j, c3 = a2 == i && b2 == j;
            if (!(c0 || c1 || c2 || c3)) continue;
            const MuNode nd = mat_node(vs, rho, nz, nx, a, b);
            if (nd.anyzero) continue;
            const float w = g[2 * n + (long long)a * nx + b] * s * (4.0f / (nd.inv * nd.inv));       // d(4 / inv) / d(1 / m)
            if (c0) dmu += w / (nd.m[0] * nd.m[0]);
            if (c1) dmu += w / (nd.m[1] * nd.m[1]);
            if (c2) dmu += w / (nd.m[2] * nd.m[2]);
            if (c3) dmu += w / (nd.m[3] * nd.m[3]);
        }
    // out3 = s / rx at (i, b), rx = (rho(i, b) + rho(i, b2)) / 2: both operands may be c
    for (int db = 1; db >= 0; --db) {
        const int b = j - db;
        if (b < 0) continue;
        const int b2 = min(b + 1, nx - 1);
        const float rx = 0.5f * (rho[(long long)i * nx + b] + rho[(long long)i * nx + b2]);
        const float w = g[3 * n + (long long)i * nx + b] * (-s / (rx * rx)) * 0.5f;
        if (b == j) drho += w;
        if (b2 == j) drho += w;
    }
    for (int da = 1; da >= 0; --da) {
        const int a = i - da;
        if (a < 0) continue;
        const int a2 = min(a + 1, nz - 1);
        const float rz = 0.5f * (rho[(long long)a * nx + j] + rho[(long long)a2 * nx + j]);
        const float w = g[4 * n + (long long)a * nx + j] * (-s / (rz * rz)) * 0.5f;
        if (a == i) drho += w;
        if (a2 == i) drho += w;
    }
    // mu = rho vs^2, lambda = rho vp^2 - 2 mu
    gvp[k] = dlam * 2.0f * r * p;
    gvs[k] = dmu * 2.0f * r * q;
    grho[k] = drho + dlam * p * p + dmu * q * q;
}

// DENISE's INVMAT1: gradients with respect to (Vp, Vs, rho) -> (Zp, Zs, rho) or (lambda, mu, rho); include/mifwi.h
template <int MODE>
__global__ __launch_bounds__(kMT) void reparam_vjp(const float *vp, const float *vs, const float *rho, const float *gvp,
                                                   const float *gvs, const float *grho, float *oa, float *ob, float *orho, long long n)
{
    const long long k = (long long)blockIdx.x * kMT + threadIdx.x;
    if (k >= n) return;
    const float p = vp[k], q = vs[k], r = rho[k], a = gvp[k], b = gvs[k], c = grho[k];
    if (MODE == 2) {
        oa[k] = a / r;
        ob[k] = b / r;
        orho[k] = c - (p * a + q * b) / r;
    } else {
        oa[k] = a / (2.0f * r * p);
        ob[k] = a / (r * p) + (q == 0.f ? 0.f : b / (2.0f * r * q));
        orho[k] = c - (p * a + q * b) / (2.0f * r);
    }
}

// ---- scalar scheme: vp [nz][nx] -> r [nz + 2 pad][nx + 2 pad] = (vp dt / h)^2, the model replicated into the absorbing
// layer (what the deepwave-shaped shim computes per call: compat/deepwave/scalar.py) ------------------------------------
__global__ __launch_bounds__(kMT) void coef_fwd(const float *vp, float *r, int nz, int nx, int pad, float c)
{
    const int n1 = nx + 2 * pad;
    const long long k = (long long)blockIdx.x * kMT + threadIdx.x, n = (long long)(nz + 2 * pad) * n1;
    if (k >= n) return;
    const int a = (int)(k / n1), b = (int)(k - (long long)a * n1);
    const int i = min(max(a - pad, 0), nz - 1), j = min(max(b - pad, 0), nx - 1);
    const float x = vp[(long long)i * nx + j] * c;
    r[k] = x * x;
}
// one thread per model cell: the cells of the layer that replicate it are folded in here, in a fixed order
__global__ __launch_bounds__(kMT) void coef_vjp(const float *vp, const float *gr, float *gvp, int nz, int nx, int pad, float c)
{
    const long long k = (long long)blockIdx.x * kMT + threadIdx.x;
    if (k >= (long long)nz * nx) return;
    const int i = (int)(k / nx), j = (int)(k - (long long)i * nx);
    const int n0 = nz + 2 * pad, n1 = nx + 2 * pad;
    const int a0 = i == 0 ? 0 : i + pad, a1 = i == nz - 1 ? n0 - 1 : i + pad;
    const int b0 = j == 0 ? 0 : j + pad, b1 = j == nx - 1 ? n1 - 1 : j + pad;
    float sum = 0.f;
    for (int a = a0; a <= a1; ++a)
        for (int b = b0; b <= b1; ++b) sum += gr[(long long)a * n1 + b];
    const float x = vp[k] * c;
    gvp[k] = sum * (2.0f * x) * c;
}

}  // namespace

extern "C" {

int mifwi_acoustic_coefficients(int device, const float *vp, float *r, int32_t nz, int32_t nx, int32_t pad, float dt_over_h,
                                void *stream)
{
    if (!vp || !r || nz < 1 || nx < 1 || pad < 0) return mifwi::fail(MIFWI_EINVAL, "mifwi_acoustic_coefficients: bad argument");
    MIFWI_HIP_TRY(hipSetDevice(device));
    const long long n = (long long)(nz + 2 * pad) * (nx + 2 * pad);
    hipLaunchKernelGGL(coef_fwd, dim3((unsigned)((n + kMT - 1) / kMT)), dim3(kMT), 0, (hipStream_t)stream, vp, r, nz, nx, pad,
                       dt_over_h);
    MIFWI_HIP_TRY(hipGetLastError());
    return MIFWI_OK;
}

int mifwi_acoustic_coefficients_vjp(int device, const float *vp, const float *grad_r, float *grad_vp, int32_t nz, int32_t nx,
                                    int32_t pad, float dt_over_h, void *stream)
{
    if (!vp || !grad_r || !grad_vp || nz < 1 || nx < 1 || pad < 0)
        return mifwi::fail(MIFWI_EINVAL, "mifwi_acoustic_coefficients_vjp: bad argument");
    MIFWI_HIP_TRY(hipSetDevice(device));
    const long long n = (long long)nz * nx;
    hipLaunchKernelGGL(coef_vjp, dim3((unsigned)((n + kMT - 1) / kMT)), dim3(kMT), 0, (hipStream_t)stream, vp, grad_r, grad_vp, nz,
                       nx, pad, dt_over_h);
    MIFWI_HIP_TRY(hipGetLastError());
    return MIFWI_OK;
}


int mifwi_elastic_gradient_parametrization(int device, int32_t parametrization, const float *vp, const float *vs, const float *rho,
                                           const float *grad_vp, const float *grad_vs, const float *grad_rho, float *out_a,
                                           float *out_b, float *out_rho, int64_t n, void *stream)
{
    if (!vp || !vs || !rho || !grad_vp || !grad_vs || !grad_rho || !out_a || !out_b || !out_rho || n < 1)
        return mifwi::fail(MIFWI_EINVAL, "mifwi_elastic_gradient_parametrization: bad argument");
    if (parametrization < MIFWI_PARAM_VELOCITY || parametrization > MIFWI_PARAM_LAME)
        return mifwi::fail(MIFWI_EINVAL, "parametrization %d: 1 = Vp/Vs/rho, 2 = Zp/Zs/rho, 3 = lambda/mu/rho", parametrization);
    MIFWI_HIP_TRY(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((n + kMT - 1) / kMT)), block(kMT);
    if (parametrization == MIFWI_PARAM_VELOCITY) {
        if (out_a != grad_vp) MIFWI_HIP_TRY(hipMemcpyAsync(out_a, grad_vp, sizeof(float) * n, hipMemcpyDeviceToDevice, st));
        if (out_b != grad_vs) MIFWI_HIP_TRY(hipMemcpyAsync(out_b, grad_vs, sizeof(float) * n, hipMemcpyDeviceToDevice, st));
        if (out_rho != grad_rho) MIFWI_HIP_TRY(hipMemcpyAsync(out_rho, grad_rho, sizeof(float) * n, hipMemcpyDeviceToDevice, st));
    } else if (parametrization == MIFWI_PARAM_IMPEDANCE) {
        hipLaunchKernelGGL(reparam_vjp<2>, grid, block, 0, st, vp, vs, rho, grad_vp, grad_vs, grad_rho, out_a, out_b, out_rho, (long long)n);
    } else {
        hipLaunchKernelGGL(reparam_vjp<3>, grid, block, 0, st, vp, vs, rho, grad_vp, grad_vs, grad_rho, out_a, out_b, out_rho, (long long)n);
    }
    MIFWI_HIP_TRY(hipGetLastError());
    return MIFWI_OK;
}

int mifwi_elastic_materials(int device, const float *vp, const float *vs, const float *rho, float *out, int32_t nz, int32_t nx,
                            float dt_over_h, int32_t free_surface, void *stream)
{
    if (!vp || !vs || !rho || !out || nz < 1 || nx < 1) return mifwi::fail(MIFWI_EINVAL, "mifwi_elastic_materials: bad argument");
    MIFWI_HIP_TRY(hipSetDevice(device));
    const long long n = (long long)nz * nx;
    hipLaunchKernelGGL(materials_fwd, dim3((unsigned)((n + kMT - 1) / kMT)), dim3(kMT), 0, (hipStream_t)stream, vp, vs, rho, out,
                       nz, nx, dt_over_h, free_surface);
    MIFWI_HIP_TRY(hipGetLastError());
    return MIFWI_OK;
}

int mifwi_elastic_materials_vjp(int device, const float *vp, const float *vs, const float *rho, const float *grad_out,
                                float *grad_vp, float *grad_vs, float *grad_rho, int32_t nz, int32_t nx, float dt_over_h,
                                int32_t free_surface, void *stream)
{
    if (!vp || !vs || !rho || !grad_out || !grad_vp || !grad_vs || !grad_rho || nz < 1 || nx < 1)
        return mifwi::fail(MIFWI_EINVAL, "mifwi_elastic_materials_vjp: bad argument");
    MIFWI_HIP_TRY(hipSetDevice(device));
    const long long n = (long long)nz * nx;
    hipLaunchKernelGGL(materials_vjp, dim3((unsigned)((n + kMT - 1) / kMT)), dim3(kMT), 0, (hipStream_t)stream, vp, vs, rho,
                       grad_out, grad_vp, grad_vs, grad_rho, nz, nx, dt_over_h, free_surface);
    MIFWI_HIP_TRY(hipGetLastError());
    return MIFWI_OK;
}

}  // extern "C"
