// Micro-benchmark: cost of a chain of dependent small kernels on MI355X (launch boundary, small
// copies from Infinity Cache, halo-style re-reads).  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_empty() {}

__global__ void k_copy(const float4 *__restrict__ a, float4 *__restrict__ b, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) b[i] = a[i];
}

// each thread reads NR float4 (strided by n/NR) and writes one: emulates "k inputs -> 1 output"
template <int NR>
__global__ void k_multi(const float4 *__restrict__ a, float4 *__restrict__ b, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const float4 v = a[(size_t)r * n + i];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    b[i] = acc;
}

static float time_chain(void (*launch)(int, hipStream_t), int iters, hipStream_t st)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) launch(i, st);
    hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    for (int i = 0; i < iters; ++i) launch(i, st);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / iters;
}

static float4 *A, *B;
static int N, BS;
static void l_empty(int, hipStream_t st) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st); }
static void l_empty_big(int, hipStream_t st) { hipLaunchKernelGGL(k_empty, dim3(2048), dim3(256), 0, st); }
static void l_copy(int i, hipStream_t st)
{
    float4 *a = (i & 1) ? B : A, *b = (i & 1) ? A : B;
    hipLaunchKernelGGL(k_copy, dim3((N + BS - 1) / BS), dim3(BS), 0, st, a, b, N);
}
template <int NR>
static void l_multi(int i, hipStream_t st)
{
    float4 *a = (i & 1) ? B : A, *b = (i & 1) ? A : B;
    hipLaunchKernelGGL((k_multi<NR>), dim3((N + BS - 1) / BS), dim3(BS), 0, st, a, b, N);
}

int main()
{
    hipStream_t st;
    hipStreamCreate(&st);
    const size_t maxn = (size_t)64 << 20;   // float4 elements (1 GiB)
    if (hipMalloc(&A, maxn * sizeof(float4)) != hipSuccess || hipMalloc(&B, maxn * sizeof(float4)) != hipSuccess) {
        printf("alloc failed\n");
        return 1;
    }
    hipMemset(A, 0, maxn * sizeof(float4));
    hipMemset(B, 0, maxn * sizeof(float4));
    printf("empty<<<1,64>>>      : %.2f us/launch\n", time_chain(l_empty, 2000, st));
    printf("empty<<<2048,256>>>  : %.2f us/launch\n", time_chain(l_empty_big, 2000, st));
    BS = 256;
    for (size_t mb : {1, 4, 16, 19, 54, 128, 512, 1024}) {
        if (mb * (1 << 20) / 16 > maxn) { printf("skip %zu\n", mb); continue; }
        N = (int)(mb * (1 << 20) / 16);
        const float t = time_chain(l_copy, 500, st);
        printf("copy %5zu MB (R+W %zu MB): %.2f us  -> %.2f TB/s\n", mb, 2 * mb, t, 2.0 * mb * 1.048576 / t);
    }
    // "5 fields in, 5 out" shaped like the elastic step: 5 reads + 1 write per thread
    for (size_t mb : {4, 19}) {
        N = (int)(mb * (1 << 20) / 16 / 5);
        float t = time_chain(l_multi<5>, 500, st);
        printf("multi5 total in %zu MB: %.2f us\n", mb, t);
        N = (int)(mb * (1 << 20) / 16 / 10);
        t = time_chain(l_multi<10>, 500, st);
        printf("multi10 total in %zu MB: %.2f us\n", mb, t);
    }
    return 0;
}
