// Gradient conditioning on the device (libmifwi, gfx950): what the reference does on the host, in numpy / scipy,
// between `d.get_fwi_gradients(...)` and `fake_Vp.backward(vp_grad)`:
//   models/networks.py:7808-7862 (9884-9919)  flipud, zero rows 0:25, scale by max(model) / max(gradient), rho x 0.1
//   models/networks.py:10522-10540            flipud, scipy.ndimage.gaussian_filter(sigma = 3), zero rows 0:5, same scaling
//   models/networks.py:7731, 9832-9833        DENISE's SWS_TAPER_GRAD_HOR / EXP_TAPER_GRAD_HOR depth window (a weight per row)
// One call, the gradient never leaves the GPU:
//   t_k   = mute( smooth_sigma( w[row] * g_k[row or nz-1-row] ) )          (tile + halo in LDS, separable Gaussian)
//   out_k = t_k * factor_k * max(model_k) / max(t_k)                        (models given; otherwise t_k * factor_k)
// The Gaussian is scipy's: radius int(4 sigma + 0.5), weights exp(-x^2 / 2 sigma^2) normalised, 'reflect' boundary
// (edge sample repeated), axis 0 then axis 1.  max() is the plain maximum (np.max), not the largest magnitude.
#include "mifwi_common.h"

#include <algorithm>
#include <cmath>

namespace {

constexpr int kGT = 256;
constexpr int kGZ = 32, kGX = 64;             // output tile
constexpr int kGRmax = 32;                    // largest Gaussian radius served (sigma <= 7.8)

__device__ __forceinline__ unsigned f_key(float x)            // order-preserving float -> unsigned
{
    const unsigned b = __float_as_uint(x);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ __forceinline__ float key_f(unsigned k)
{
    const unsigned b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
#ifdef __HIP_DEVICE_COMPILE__
    return __uint_as_float(b);
#else
    float f;
    memcpy(&f, &b, 4);
    return f;
#endif
}
__device__ __forceinline__ int reflect(int i, int n)          // scipy 'reflect': d c b a | a b c d | d c b a
{
    const int p = 2 * n;
    i %= p;
    if (i < 0) i += p;
    return i < n ? i : p - 1 - i;
}

// block maximum of v (lanes without a value pass -inf's key 0) into keys[slot]
__device__ __forceinline__ void block_max(unsigned key, unsigned *keys, int slot)
{
    for (int off = 32; off > 0; off >>= 1) key = max(key, (unsigned)__shfl_down((int)key, off, 64));
    __shared__ unsigned s[kGT / 64];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = key;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned m = s[0];
        for (int k = 1; k < kGT / 64; ++k) m = max(m, s[k]);
        atomicMax(keys + slot, m);
    }
}

__global__ __launch_bounds__(kGT) void grad_model_max(const float *models, long long n, unsigned *keys)
{
    const float *m = models + (long long)blockIdx.y * n;
    unsigned key = 0;
    for (long long i = (long long)blockIdx.x * kGT + threadIdx.x; i < n; i += (long long)gridDim.x * kGT)
        key = max(key, f_key(m[i]));
    block_max(key, keys, (int)blockIdx.y);
}

// weights [2R+1] in LDS; tile + halo staged once, z pass into a second plane, x pass to the output
__global__ __launch_bounds__(kGT) void grad_condition(const float *grad, float *out, int nz, int nx, const float *row_w,
                                                      float sigma, int R, int flip, int mute_rows, unsigned *keys)
{
    extern __shared__ float lds[];
    const int SW = kGX + 2 * R, SH = kGZ + 2 * R;
    float *w = lds;                               // [2R+1]
    float *A = lds + 2 * kGRmax + 4;              // staged input [SH][SW]
    float *B = A + SH * SW;                       // after the z pass [kGZ][SW]
    const int k = (int)blockIdx.z;
    const float *g = grad + (long long)k * nz * nx;
    float *o = out + (long long)k * nz * nx;
    const int j0 = (int)blockIdx.y * kGZ, i0 = (int)blockIdx.x * kGX;
    const int t = (int)threadIdx.x;
    if (R > 0) {
        // scipy _gaussian_kernel1d: exp(-0.5 x^2 / sigma^2) / sum, in double, handed over as float
        if (t <= 2 * R) {
            double sum = 0.0;
            for (int x = -R; x <= R; ++x) sum += exp(-0.5 * (double)x * x / ((double)sigma * sigma));
            const int x = t - R;
            w[t] = (float)(exp(-0.5 * (double)x * x / ((double)sigma * sigma)) / sum);
        }
    }
    for (int e = t; e < SH * SW; e += kGT) {
        const int lr = e / SW, lc = e - lr * SW;
        const int j = reflect(j0 - R + lr, nz), i = reflect(i0 - R + lc, nx);
        const int js = flip ? nz - 1 - j : j;                          // row of the stored gradient
        float v = g[(long long)js * nx + i];
        if (row_w) v *= row_w[js];
        A[e] = v;
    }
    __syncthreads();
    if (R > 0) {
        for (int e = t; e < kGZ * SW; e += kGT) {
            const int lr = e / SW, lc = e - lr * SW;
            float a = 0.f;
            for (int q = 0; q <= 2 * R; ++q) a = fmaf(w[q], A[(lr + q) * SW + lc], a);
            B[e] = a;
        }
        __syncthreads();
    }
    unsigned key = 0;
    for (int e = t; e < kGZ * kGX; e += kGT) {
        const int lr = e / kGX, lc = e - lr * kGX;
        const int j = j0 + lr, i = i0 + lc;
        if (j >= nz || i >= nx) continue;
        float a;
        if (R > 0) {
            a = 0.f;
            for (int q = 0; q <= 2 * R; ++q) a = fmaf(w[q], B[lr * SW + lc + q], a);
        } else {
            a = A[lr * SW + lc];
        }
        if (j < mute_rows) a = 0.f;
        o[(long long)j * nx + i] = a;
        key = max(key, f_key(a));
    }
    if (keys) block_max(key, keys, k);
}

__global__ __launch_bounds__(kGT) void grad_scale(float *out, long long n, const unsigned *keys, int nplane, int have_models,
                                                  float f0, float f1, float f2, float f3)
{
    const int k = (int)blockIdx.y;
    float s = k == 0 ? f0 : k == 1 ? f1 : k == 2 ? f2 : f3;
    if (have_models) s = s * (key_f(keys[nplane + k]) / key_f(keys[k]));       // np.max(model) / np.max(grad)
    float *o = out + (long long)k * n;
    for (long long i = (long long)blockIdx.x * kGT + threadIdx.x; i < n; i += (long long)gridDim.x * kGT) o[i] *= s;
}

}  // namespace

extern "C" {

int64_t mifwi_gradient_condition_work_elems(int32_t nplane) { return mifwi::round_up64(2LL * std::max(nplane, 1), 64); }

int mifwi_gradient_condition(int device, const float *grad, const float *models, float *out, int32_t nplane, int32_t nz,
                             int32_t nx, const float *row_weight, float sigma, int32_t flip, int32_t mute_rows,
                             const float *factors, float *work, void *stream)
{
    if (!grad || !out || !work) return mifwi::fail(MIFWI_EINVAL, "null argument");
    if (nplane < 1 || nplane > 4 || nz < 1 || nx < 1) return mifwi::fail(MIFWI_EINVAL, "bad sizes: %d planes of %d x %d", nplane, nz, nx);
    if (grad == out) return mifwi::fail(MIFWI_EINVAL, "conditioning is not in place (the smoothing reads neighbours)");
    if (!(sigma >= 0.f)) return mifwi::fail(MIFWI_EINVAL, "sigma must be >= 0");
    const int R = sigma > 0.f ? (int)(4.0f * sigma + 0.5f) : 0;
    if (R > kGRmax) return mifwi::fail(MIFWI_EINVAL, "sigma %g needs a radius of %d cells, the kernel serves up to %d", sigma, R, kGRmax);
    int rc = mifwi::check_device(device);
    if (rc) return rc;
    MIFWI_HIP_TRY(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    unsigned *keys = reinterpret_cast<unsigned *>(work);
    MIFWI_HIP_TRY(hipMemsetAsync(keys, 0, sizeof(unsigned) * 2 * nplane, st));
    const long long n = (long long)nz * nx;
    if (models)
        hipLaunchKernelGGL(grad_model_max, dim3((unsigned)std::min<long long>(256, (n + kGT - 1) / kGT), nplane), dim3(kGT), 0, st,
                           models, n, keys + nplane);
    const int SW = kGX + 2 * R, SH = kGZ + 2 * R;
    const size_t lds = sizeof(float) * (2 * kGRmax + 4 + (size_t)SH * SW + (size_t)kGZ * SW);
    if (lds > 48 * 1024)                              // 96 x 128 + 32 x 128 floats at the largest radius: 66 KB
        MIFWI_HIP_TRY(hipFuncSetAttribute((const void *)grad_condition, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
    hipLaunchKernelGGL(grad_condition, dim3(mifwi::ceil_div(nx, kGX), mifwi::ceil_div(nz, kGZ), nplane), dim3(kGT), lds, st, grad,
                       out, nz, nx, row_weight, sigma, R, flip ? 1 : 0, mute_rows, models ? keys : nullptr);
    float f[4] = {1.f, 1.f, 1.f, 1.f};
    bool any = models != nullptr;
    for (int k = 0; k < nplane && factors; ++k) { f[k] = factors[k]; any = any || f[k] != 1.f; }
    if (any)
        hipLaunchKernelGGL(grad_scale, dim3((unsigned)std::min<long long>(256, (n + kGT - 1) / kGT), nplane), dim3(kGT), 0, st, out,
                           n, keys, nplane, models ? 1 : 0, f[0], f[1], f[2], f[3]);
    MIFWI_HIP_TRY(hipGetLastError());
    return MIFWI_OK;
}

}  // extern "C"
