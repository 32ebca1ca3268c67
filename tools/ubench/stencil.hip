// Micro-benchmark: what does a V-update-shaped kernel cost on MI355X as features are added?
// Layout mimics libmifwi: fields [ns][5][nz+4][pitch], materials [5][nz][gp].
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

struct P { int nz, ng, gp, pitch; unsigned fs; long long ss; const float *mat; float *f; };

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ float2 ld2(const float *p) { return *reinterpret_cast<const float2 *>(p); }

// MODE 0: centre loads only (7 in, 2 out) ; 1: + z neighbours ; 2: + x halos (full V pattern)
template <int MODE, int LX, int OCC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void k_v(const P p)
{
    constexpr int LZ = 256 / LX;
    const int lx = threadIdx.x % LX, lz = threadIdx.x / LX;
    const int g = blockIdx.x * LX + lx, j = blockIdx.y * LZ + lz;
    if (g >= p.ng || j >= p.nz) return;
    float *fl = p.f + (long long)blockIdx.z * p.ss;
    const unsigned o = (unsigned)(j + 2) * p.pitch + 4 + 4 * g;
    const unsigned cc = (unsigned)j * p.gp + 4 * g;
    const unsigned ncell = (unsigned)p.nz * p.gp;
    float4 acc = ld4(fl + 2 * p.fs + o);
    float4 t = ld4(fl + 3 * p.fs + o); acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
    t = ld4(fl + 4 * p.fs + o); acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
    if (MODE >= 1) {
#pragma unroll
        for (int k = -2; k <= 2; ++k) {
            if (k == 0) continue;
            t = ld4(fl + 4 * p.fs + o + k * p.pitch); acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
            if (k != -2) { t = ld4(fl + 3 * p.fs + o + k * p.pitch); acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w; }
        }
    }
    if (MODE >= 2) {
        float2 h = ld2(fl + 2 * p.fs + o - 2); acc.x += h.x + h.y;
        h = ld2(fl + 2 * p.fs + o + 4); acc.y += h.x + h.y;
        h = ld2(fl + 4 * p.fs + o - 2); acc.z += h.x + h.y;
        h = ld2(fl + 4 * p.fs + o + 4); acc.w += h.x + h.y;
    }
    const float4 bx = ld4(p.mat + 3 * ncell + cc), bz = ld4(p.mat + 4 * ncell + cc);
    float4 vx = ld4(fl + o), vz = ld4(fl + p.fs + o);
    vx.x += bx.x * acc.x; vx.y += bx.y * acc.y; vx.z += bx.z * acc.z; vx.w += bx.w * acc.w;
    vz.x += bz.x * acc.x; vz.y += bz.y * acc.y; vz.z += bz.z * acc.z; vz.w += bz.w * acc.w;
    *reinterpret_cast<float4 *>(fl + o) = vx;
    *reinterpret_cast<float4 *>(fl + p.fs + o) = vz;
}

template <int MODE, int LX, int OCC = 8>
float run(const P &p, int ns, int iters, hipStream_t st)
{
    dim3 grid((p.ng + LX - 1) / LX, (p.nz + 256 / LX - 1) / (256 / LX), ns), block(256);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k_v<MODE, LX, OCC>), grid, block, 0, st, p);
    (void)hipStreamSynchronize(st);
    (void)hipEventRecord(e0, st);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k_v<MODE, LX, OCC>), grid, block, 0, st, p);
    (void)hipEventRecord(e1, st);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / iters;
}

int main(int argc, char **argv)
{
    const int nz = argc > 1 ? atoi(argv[1]) : 100, nx = argc > 2 ? atoi(argv[2]) : 300;
    const int ns = argc > 3 ? atoi(argv[3]) : 32;
    const int padrows = argc > 4 ? atoi(argv[4]) : 0;    // extra rows per field to de-alias strides
    P p;
    p.nz = nz; p.ng = (nx + 3) / 4; p.gp = 4 * p.ng;
    p.pitch = ((4 * (p.ng + 3) + 31) / 32) * 32;
    p.fs = (unsigned)(nz + 4 + padrows) * p.pitch;
    p.ss = 5LL * p.fs;
    float *f, *mat;
    const size_t nf = (size_t)p.ss * ns, nm = (size_t)5 * nz * p.gp;
    if (hipMalloc(&f, nf * 4) != hipSuccess || hipMalloc(&mat, nm * 4) != hipSuccess) return 1;
    (void)hipMemset(f, 0, nf * 4); (void)hipMemset(mat, 0, nm * 4);
    p.f = f; p.mat = mat;
    hipStream_t st;
    (void)hipStreamCreate(&st);
    const double mb = (double)nz * p.gp * ns * 4 / 1e6;
    printf("grid %dx%d ns=%d pitch=%d fs=%u (%.1f KB) field-set %.1f MB; V moves ~%.1f MB\n", nz, nx, ns,
           p.pitch, p.fs, p.fs * 4 / 1024.0, 5 * mb, 9 * mb);
    printf("centre only   LX16 %.2f  LX32 %.2f  LX64 %.2f us\n", run<0, 16>(p, ns, 300, st), run<0, 32>(p, ns, 300, st), run<0, 64>(p, ns, 300, st));
    printf("+z neighbours LX16 %.2f  LX32 %.2f  LX64 %.2f us\n", run<1, 16>(p, ns, 300, st), run<1, 32>(p, ns, 300, st), run<1, 64>(p, ns, 300, st));
    printf("+x halos      LX16 %.2f  LX32 %.2f  LX64 %.2f us\n", run<2, 16>(p, ns, 300, st), run<2, 32>(p, ns, 300, st), run<2, 64>(p, ns, 300, st));
    printf("full V pattern LX32 at occupancy 1..8 waves/SIMD: %.2f %.2f %.2f %.2f %.2f %.2f us\n", run<2, 32, 1>(p, ns, 300, st), run<2, 32, 2>(p, ns, 300, st), run<2, 32, 3>(p, ns, 300, st), run<2, 32, 4>(p, ns, 300, st), run<2, 32, 6>(p, ns, 300, st), run<2, 32, 8>(p, ns, 300, st));
    return 0;
}
