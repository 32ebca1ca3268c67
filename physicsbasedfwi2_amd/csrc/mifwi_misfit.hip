// Fused data misfit + adjoint source (libmifwi, gfx950).
//
// Replaces the O(data) torch expressions between the propagator call and .backward() in the
// reference's prop() methods:
//   kind L1_TRACE_NORM  models/networks.py:5418-5419 (observed side), 5467-5476 (predicted side):
//       d = pred - direct;  m = max_t |d|;  dn = d / (m + 1e-10);  loss = mean |dn - obs|
//       adj = dloss/dpred (what autograd's backward of those lines delivers to the propagator)
//   kind L2             seisgan/fwi/layers.py:176-178 and DENISE lnorm=2 (networks.py:7758):
//       loss = 1/2 sum (pred - obs)^2;  adj = pred - obs
//   kind GLOBAL_CORRELATION  (DENISE's global-correlation norm, the `lnorm` argument of add_fwi_stage at
//       models/networks.py:9863, 10503; Choi & Alkhalifah 2012): per trace s = pred, o = obs
//       loss = - sum_traces <s, o> / (|s| |o|);  adj = -(o/|o| - <s,o>/(|s||o|) s/|s|) / |s|
//       (a trace with |s| = 0 or |o| = 0 contributes nothing)
// Layout [nt][ntrace] (trace = shot*nrec + receiver fastest), i.e. the propagators' own output.
//
// One workgroup = 64 neighbouring traces x 16 interleaved time slices: every wave reads whole
// 256-byte rows, traces never leave their lane, the 16 slices of a trace meet twice in LDS
// (max/argmax, then the sum the max's gradient needs).  HBM-bound: pred/direct are read twice,
// obs once, adj written once = 28 B per sample (20 without a direct wave).
#include "mifwi_common.h"

namespace {

constexpr int kTr = 64, kSl = 16;          // traces x time slices per workgroup (1024 threads)
constexpr float kEps = 1e-10f;

__device__ __forceinline__ float sgn(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }

template <bool DIRECT>
__global__ __launch_bounds__(kTr *kSl) void misfit_l1_trace_norm(const float *pred, const float *obs,
                                                                 const float *direct, int nt, long long ntrace,
                                                                 float inv_n, float *adj, double *partial)
{
    __shared__ float s_max[kSl][kTr];
    __shared__ int s_arg[kSl][kTr];
    __shared__ float s_c[kSl][kTr];
    __shared__ double s_loss[kSl];
    const int lane = (int)threadIdx.x % kTr, sl = (int)threadIdx.x / kTr;
    const long long tr = (long long)blockIdx.x * kTr + lane;
    const bool live = tr < ntrace;
    // ---- pass 1: max |d| and the first time it is reached --------------------------------------
    float m = -1.f;
    int arg = 0;
    if (live)
        for (int t = sl; t < nt; t += kSl) {
            const long long o = (long long)t * ntrace + tr;
            const float d = DIRECT ? pred[o] - direct[o] : pred[o];
            const float a = fabsf(d);
            if (a > m) { m = a; arg = t; }
        }
    s_max[sl][lane] = m; s_arg[sl][lane] = arg;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kSl; ++k) {
        const float mk = s_max[k][lane];
        const int ak = s_arg[k][lane];
        if (mk > m || (mk == m && ak < arg)) { m = mk; arg = ak; }
    }
    const float inv = 1.0f / (m + kEps);
    // ---- pass 2: residual sign, loss, adj without the max's own gradient ----------------------
    float c = 0.f;
    double loss = 0.0;
    if (live)
        for (int t = sl; t < nt; t += kSl) {
            const long long o = (long long)t * ntrace + tr;
            const float d = DIRECT ? pred[o] - direct[o] : pred[o];
            const float r = d * inv - obs[o];
            const float a = sgn(r) * inv_n;
            loss += (double)fabsf(r);
            c = fmaf(a, d, c);
            if (adj) adj[o] = a * inv;
        }
    s_c[sl][lane] = c;
    // workgroup loss: wave reduction, then the 16 waves in order
    for (int off = 32; off > 0; off >>= 1) loss += __shfl_down(loss, off, 64);
    if (lane == 0) s_loss[sl] = loss;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int k = 0; k < kSl; ++k) tot += s_loss[k];
        partial[blockIdx.x] = tot;
    }
    // ---- the max feeds every sample of its trace: d(dn_t)/d(d_t*) = -d_t sign(d_t*) / (m+eps)^2 --
    if (live && adj && sl == arg % kSl) {
        float ctot = 0.f;
#pragma unroll
        for (int k = 0; k < kSl; ++k) ctot += s_c[k][lane];
        const long long o = (long long)arg * ntrace + tr;
        const float d = DIRECT ? pred[o] - direct[o] : pred[o];
        adj[o] = adj[o] - sgn(d) * ctot * inv * inv;
    }
}

// global correlation: pass 1 the three inner products of a trace (fp64, fixed order), pass 2 the adjoint source
__global__ __launch_bounds__(kTr *kSl) void misfit_global_correlation(const float *pred, const float *obs, int nt,
                                                                      long long ntrace, float *adj, double *partial)
{
    __shared__ double s_ss[kSl][kTr], s_oo[kSl][kTr], s_so[kSl][kTr];
    __shared__ double s_loss[kSl];
    const int lane = (int)threadIdx.x % kTr, sl = (int)threadIdx.x / kTr;
    const long long tr = (long long)blockIdx.x * kTr + lane;
    const bool live = tr < ntrace;
    double ss = 0.0, oo = 0.0, so = 0.0;
    if (live)
        for (int t = sl; t < nt; t += kSl) {
            const long long o = (long long)t * ntrace + tr;
            const double a = (double)pred[o], b = (double)obs[o];
            ss += a * a; oo += b * b; so += a * b;
        }
    s_ss[sl][lane] = ss; s_oo[sl][lane] = oo; s_so[sl][lane] = so;
    __syncthreads();
    ss = oo = so = 0.0;
#pragma unroll
    for (int k = 0; k < kSl; ++k) { ss += s_ss[k][lane]; oo += s_oo[k][lane]; so += s_so[k][lane]; }
    const bool ok = live && ss > 0.0 && oo > 0.0;
    const double ns = sqrt(ss), no = sqrt(oo);
    const double c = ok ? so / (ns * no) : 0.0;
    if (adj && live) {
        // in double: on a weak trace the two coefficients overflow a float long before their combination does
        const double ka = ok ? -1.0 / (ns * no) : 0.0;               // coefficient of o
        const double kb = ok ? c / ss : 0.0;                         // coefficient of s
        for (int t = sl; t < nt; t += kSl) {
            const long long o = (long long)t * ntrace + tr;
            adj[o] = (float)(ka * (double)obs[o] + kb * (double)pred[o]);
        }
    }
    double loss = sl == 0 ? -c : 0.0;                                   // one slice speaks for the trace
    for (int off = 32; off > 0; off >>= 1) loss += __shfl_down(loss, off, 64);
    if (lane == 0) s_loss[sl] = loss;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = s_loss[0];
}

__global__ __launch_bounds__(256) void misfit_l2(const float *pred, const float *obs, long long n, float *adj,
                                                 double *partial)
{
    __shared__ double s_loss[4];
    const long long stride = (long long)gridDim.x * blockDim.x * 4;
    double loss = 0.0;
    for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 3 < n) {
            const float4 p = *reinterpret_cast<const float4 *>(pred + i);
            const float4 q = *reinterpret_cast<const float4 *>(obs + i);
            const float4 r = make_float4(p.x - q.x, p.y - q.y, p.z - q.z, p.w - q.w);
            loss += (double)(r.x * r.x) + (double)(r.y * r.y) + (double)(r.z * r.z) + (double)(r.w * r.w);
            if (adj) *reinterpret_cast<float4 *>(adj + i) = r;
        } else {
            for (long long k = i; k < n; ++k) {
                const float r = pred[k] - obs[k];
                loss += (double)(r * r);
                if (adj) adj[k] = r;
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) loss += __shfl_down(loss, off, 64);
    if ((threadIdx.x & 63) == 0) s_loss[threadIdx.x >> 6] = loss;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = s_loss[0] + s_loss[1] + s_loss[2] + s_loss[3];
}

// fixed-order sum of the workgroup partials (run-to-run deterministic loss)
__global__ __launch_bounds__(256) void misfit_finish(const double *partial, int n, double scale, float *loss_out)
{
    __shared__ double s[256];
    double a = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) a += partial[i];
    s[threadIdx.x] = a;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) s[threadIdx.x] += s[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss_out = (float)(s[0] * scale);
}

int l2_blocks(long long n) { return (int)std::min<long long>(2048, (n / 4 + 255) / 256 + 1); }

}  // namespace

extern "C" {

int64_t mifwi_misfit_work_elems(int32_t kind, int64_t nt, int64_t ntrace)
{
    if (nt < 1 || ntrace < 1) return 0;
    const long long blocks = kind == MIFWI_MISFIT_L2 ? l2_blocks(nt * ntrace) : (ntrace + kTr - 1) / kTr;
    return mifwi::round_up64(2 * blocks + 2, 64);          // doubles, counted in floats
}

int mifwi_misfit(int device, int32_t kind, const float *pred, const float *obs, const float *direct,
                 int64_t nt, int64_t ntrace, float *loss_out, float *adj_out, float *work, void *stream)
{
    if (!pred || !obs || !loss_out || !work) return mifwi::fail(MIFWI_EINVAL, "null argument");
    if (nt < 1 || ntrace < 1 || nt > 0x7fffffff)
        return mifwi::fail(MIFWI_EINVAL, "bad sizes nt=%lld ntrace=%lld", (long long)nt, (long long)ntrace);
    if (kind != MIFWI_MISFIT_L1_TRACE_NORM && kind != MIFWI_MISFIT_L2 && kind != MIFWI_MISFIT_GLOBAL_CORRELATION)
        return mifwi::fail(MIFWI_EINVAL, "unknown misfit kind %d", kind);
    if (kind != MIFWI_MISFIT_L1_TRACE_NORM && direct)
        return mifwi::fail(MIFWI_EINVAL, "only the trace-normalised L1 misfit takes a direct wave");
    if ((reinterpret_cast<uintptr_t>(work) & 7) != 0) return mifwi::fail(MIFWI_EINVAL, "work must be 8-byte aligned");
    int rc = mifwi::check_device(device);
    if (rc) return rc;
    MIFWI_HIP_TRY(hipSetDevice(device));
    hipStream_t st = (hipStream_t)stream;
    double *partial = reinterpret_cast<double *>(work);
    const long long n = nt * ntrace;
    if (kind == MIFWI_MISFIT_L1_TRACE_NORM) {
        const long long blocks = (ntrace + kTr - 1) / kTr;
        if (blocks > 0x7fffffff) return mifwi::fail(MIFWI_EINVAL, "too many traces");
        const float inv_n = (float)(1.0 / (double)n);
        if (direct)
            hipLaunchKernelGGL(misfit_l1_trace_norm<true>, dim3((unsigned)blocks), dim3(kTr * kSl), 0, st, pred, obs,
                               direct, (int)nt, (long long)ntrace, inv_n, adj_out, partial);
        else
            hipLaunchKernelGGL(misfit_l1_trace_norm<false>, dim3((unsigned)blocks), dim3(kTr * kSl), 0, st, pred, obs,
                               direct, (int)nt, (long long)ntrace, inv_n, adj_out, partial);
        hipLaunchKernelGGL(misfit_finish, dim3(1), dim3(256), 0, st, partial, (int)blocks, 1.0 / (double)n, loss_out);
    } else if (kind == MIFWI_MISFIT_GLOBAL_CORRELATION) {
        const long long blocks = (ntrace + kTr - 1) / kTr;
        if (blocks > 0x7fffffff) return mifwi::fail(MIFWI_EINVAL, "too many traces");
        hipLaunchKernelGGL(misfit_global_correlation, dim3((unsigned)blocks), dim3(kTr * kSl), 0, st, pred, obs, (int)nt,
                           (long long)ntrace, adj_out, partial);
        hipLaunchKernelGGL(misfit_finish, dim3(1), dim3(256), 0, st, partial, (int)blocks, 1.0, loss_out);
    } else {
        const int blocks = l2_blocks(n);
        hipLaunchKernelGGL(misfit_l2, dim3(blocks), dim3(256), 0, st, pred, obs, n, adj_out, partial);
        hipLaunchKernelGGL(misfit_finish, dim3(1), dim3(256), 0, st, partial, blocks, 0.5, loss_out);
    }
    MIFWI_HIP_TRY(hipGetLastError());
    return MIFWI_OK;
}

}  // extern "C"
