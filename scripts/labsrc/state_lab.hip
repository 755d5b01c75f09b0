/*
 * state_lab.hip -- LABORATORY (round 5): the store pattern of a trace runs
 * at one of two speeds on the SAME pieces of memory in the SAME order
 * (arrange_lab: 1.19 ms for the first forty arrangements of a process, 0.98
 * from then on, the same ten pieces in both halves) -- a state of the
 * device, not a property of the placement.  This program holds everything
 * else still and watches the state: one array built from 1 GiB pieces and
 * one plain hipMalloc of the same size, the C3 pattern (two blocks) and the
 * C2 pattern timed on both in turn for `seconds`, with idle gaps at given
 * times, every measurement stamped with the wall clock so that a sampler of
 * the SMU's metrics (scripts/state_watch.py) can be laid beside it.
 *   hipcc --offload-arch=gfx950 -O3 -o state_lab state_lab.hip
 *   ./state_lab seconds [gap_at gap_seconds]...
 */
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x,          \
                    hipGetErrorString(e_));                                    \
            exit(2);                                                           \
        }                                                                      \
    } while (0)

struct lay {
    double *Y, *U, *T;
    long long bs, ts;
};

__global__ __launch_bounds__(256) void pattern(lay a, int L, long long n)
{
    const long long j = (long long)blockIdx.x * 256 + threadIdx.x;
    if (j >= n)
        return;
    long long r = j;
    if (a.ts) {
        const unsigned long long b = (unsigned long long)j / (unsigned long long)a.bs;
        r = (long long)b * a.ts + (j - (long long)b * a.bs);
    }
    const double v = 1e-9 * (double)j;
    for (int s = 1; s < L; ++s) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            __builtin_nontemporal_store(v + c, &a.Y[(long long)s * 3 * a.bs + c * a.bs + r]);
            __builtin_nontemporal_store(v - c, &a.U[(long long)s * 3 * a.bs + c * a.bs + r]);
        }
        __builtin_nontemporal_store(v, &a.T[(long long)s * a.bs + r]);
    }
}

static double now()
{
    return std::chrono::duration<double>(
               std::chrono::system_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv)
{
    const double seconds = argc > 1 ? atof(argv[1]) : 30.;
    std::vector<std::pair<double, double>> gaps;
    for (int k = 2; k + 1 < argc; k += 2)
        gaps.push_back({atof(argv[k]), atof(argv[k + 1])});
    CK(hipSetDevice(0));
    hipStream_t st;
    hipEvent_t e0, e1;
    CK(hipStreamCreate(&st));
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const size_t piece = (size_t)1 << 30;
    const int need = 10;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = 0;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    void *A = NULL, *B = NULL;
    CK(hipMemAddressReserve(&A, need * piece, piece, NULL, 0));
    for (int k = 0; k < need; ++k) {
        hipMemGenericAllocationHandle_t h;
        CK(hipMemCreate(&h, piece, &prop, 0));
        CK(hipMemMap((char *)A + k * piece, piece, 0, h, 0));
    }
    CK(hipMemSetAccess(A, need * piece, &acc, 1));
    CK(hipMalloc(&B, need * piece));
    auto c3 = [&](void *base) {
        lay a;
        a.bs = 5000192;
        a.ts = 10LL * 13 * a.bs;
        a.Y = (double *)base;
        a.U = a.Y + 3LL * 13 * a.bs;
        a.T = a.Y + 9LL * 13 * a.bs;
        return a;
    };
    auto c2 = [&](void *base) {
        lay a;
        a.bs = 3000000;
        a.ts = 0;
        a.Y = (double *)base;
        a.U = a.Y + 3LL * 9 * a.bs;
        a.T = a.Y + 9LL * 9 * a.bs;
        return a;
    };
    auto run = [&](const lay &a, int L, long long n, int reps) {
        const unsigned grid = (unsigned)((n + 255) / 256);
        CK(hipEventRecord(e0, st));
        for (int k = 0; k < reps; ++k)
            hipLaunchKernelGGL(pattern, dim3(grid), dim3(256), 32768, st, a, L, n);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        return ms / reps;
    };
    const double t0 = now();
    printf("{\"t0\": %.6f}\n", t0);
    size_t gi = 0;
    while (now() - t0 < seconds) {
        if (gi < gaps.size() && now() - t0 >= gaps[gi].first) {
            printf("{\"gap_at\": %.4f, \"seconds\": %.3f}\n", now() - t0, gaps[gi].second);
            fflush(stdout);
            std::this_thread::sleep_for(std::chrono::duration<double>(gaps[gi].second));
            ++gi;
        }
        const double t = now();
        const double a3 = run(c3(A), 13, 2 * 5000192LL, 4);
        const double b3 = run(c3(B), 13, 2 * 5000192LL, 4);
        const double a2 = run(c2(A), 9, 3000000LL, 16);
        const double b2 = run(c2(B), 9, 3000000LL, 16);
        printf("{\"t\": %.4f, \"c3_pieces\": %.4f, \"c3_malloc\": %.4f, \"c2_pieces\": %.4f, "
               "\"c2_malloc\": %.4f}\n", t - t0, a3, b3, a2, b2);
    }
    fflush(stdout);
    return 0;
}
