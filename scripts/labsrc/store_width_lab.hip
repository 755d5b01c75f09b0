/*
 * store_width_lab.hip -- LABORATORY (round 6): does the store pattern's floor
 * depend on how wide a wavefront's store into ONE row stream is?  The trace
 * writes 84 row streams, 512 B per wavefront and store (8 B per lane); here
 * the same 84 streams over 10 GiB are written with 8, 16 and 32 B per lane
 * (512 B, 1 KiB, 2 KiB contiguous per wavefront and stream), non-temporal,
 * same total bytes, the same memory (hipMalloc: one allocation for all
 * variants, so only the RATIOS mean something).
 *   hipcc --offload-arch=gfx950 -O3 -o store_width_lab store_width_lab.hip
 */
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__,                  \
                    hipGetErrorString(e_));                                    \
            exit(2);                                                           \
        }                                                                      \
    } while (0)

template <int R> /* R doubles per lane and stream */
__global__ __launch_bounds__(256) void rows(double *base, long long n,
                                            long long pitch, int streams)
{
    const long long j = ((long long)blockIdx.x * 256 + threadIdx.x) * R;
    if (j >= n)
        return;
    const double v = 1e-9 * (double)j;
    for (int s = 0; s < streams; ++s) {
        double *p = base + (long long)s * pitch + j;
#pragma unroll
        for (int r = 0; r < R; ++r)
            __builtin_nontemporal_store(v + r + s, p + r);
    }
}

template <int R>
static double run(double *d, long long n, long long pitch, int streams,
                  hipStream_t st)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    const unsigned grid = (unsigned)((n / R + 255) / 256);
    for (int w = 0; w < 3; ++w)
        hipLaunchKernelGGL(rows<R>, dim3(grid), dim3(256), 32768, st, d, n,
                           pitch, streams);
    CK(hipEventRecord(a, st));
    const int reps = 20;
    for (int w = 0; w < reps; ++w)
        hipLaunchKernelGGL(rows<R>, dim3(grid), dim3(256), 32768, st, d, n,
                           pitch, streams);
    CK(hipEventRecord(b, st));
    CK(hipEventSynchronize(b));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, a, b));
    return 8. * n * streams * reps / (ms * 1e-3) / 1e9;
}

int main()
{
    const long long n = 10000128, pitch = n; /* rays; streams one pitch apart */
    const int streams = 84;
    double *d = NULL;
    CK(hipMalloc((void **)&d, (size_t)pitch * streams * 8 + 4096));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (int round = 0; round < 3; ++round)
        printf("round %d: 8 B/lane %.0f GB/s, 16 B/lane %.0f, 32 B/lane %.0f\n",
               round, run<1>(d, n, pitch, streams, st),
               run<2>(d, n, pitch, streams, st),
               run<4>(d, n, pitch, streams, st));
    return 0;
}
