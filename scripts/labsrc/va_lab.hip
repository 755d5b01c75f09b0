/*
 * va_lab.hip -- LABORATORY (round 5): the same ten 1 GiB pieces of physical
 * memory behind DIFFERENT virtual addresses.  map_lab: a set of pieces runs
 * the trace's store pattern at 1.157 ms behind one address range and at
 * 1.008 ms behind another, whatever the order of the pieces -- the speed
 * follows the virtual address.  Here one big reservation is scanned: the
 * pieces are mapped at offset after offset inside it (and behind separate
 * reservations, and next to plain hipMallocs), both patterns timed at each.
 *   hipcc --offload-arch=gfx950 -O3 -o va_lab va_lab.hip
 */
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
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

static hipStream_t st;
static hipEvent_t e0, e1;

static double run(const lay &a, int L, long long n, int reps)
{
    const unsigned grid = (unsigned)((n + 255) / 256);
    std::vector<float> v;
    for (int b = 0; b < 5; ++b) {
        CK(hipEventRecord(e0, st));
        for (int k = 0; k < reps; ++k)
            hipLaunchKernelGGL(pattern, dim3(grid), dim3(256), 32768, st, a, L, n);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        v.push_back(ms / reps);
    }
    std::sort(v.begin(), v.end());
    return v[2];
}

static void both(const char *what, int k, void *base)
{
    lay a;
    a.bs = 5000192;
    a.ts = 10LL * 13 * a.bs;
    a.Y = (double *)base;
    a.U = a.Y + 3LL * 13 * a.bs;
    a.T = a.Y + 9LL * 13 * a.bs;
    const double c3 = run(a, 13, 2 * a.bs, 4);
    /* the same rays as ONE block (rows 80 MB apart) */
    a.bs = 10000000;
    a.ts = 0;
    a.U = a.Y + 3LL * 13 * a.bs;
    a.T = a.Y + 9LL * 13 * a.bs;
    const double c3one = run(a, 13, a.bs, 4);
    a.bs = 3000000;
    a.ts = 0;
    a.U = a.Y + 3LL * 9 * a.bs;
    a.T = a.Y + 9LL * 9 * a.bs;
    const double c2 = run(a, 9, a.bs, 16);
    printf("{\"what\": \"%s\", \"k\": %d, \"va\": \"0x%llx\", \"va_gib\": %.3f, "
           "\"c3_ms\": %.4f, \"c3_one_block_ms\": %.4f, \"c2_ms\": %.4f}\n",
           what, k, (unsigned long long)base, (double)(unsigned long long)base / (1 << 30),
           c3, c3one, c2);
    fflush(stdout);
}

int main()
{
    CK(hipSetDevice(0));
    CK(hipStreamCreate(&st));
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = 0;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    const size_t G = (size_t)1 << 30;
    const int need = 10;
    hipMemGenericAllocationHandle_t h[need];
    for (int k = 0; k < need; ++k)
        CK(hipMemCreate(&h[k], G, &prop, 0));
    auto map_at = [&](void *base) {
        for (int k = 0; k < need; ++k)
            CK(hipMemMap((char *)base + k * G, G, 0, h[k], 0));
        CK(hipMemSetAccess(base, need * G, &acc, 1));
    };
    auto unmap_at = [&](void *base) {
        CK(hipStreamSynchronize(st));
        CK(hipMemUnmap(base, need * G));
    };
    /* warm the clocks on a first mapping */
    {
        void *w = NULL;
        CK(hipMemAddressReserve(&w, need * G, G, NULL, 0));
        map_at(w);
        lay a;
        a.bs = 10000000;
        a.ts = 0;
        a.Y = (double *)w;
        a.U = a.Y + 3LL * 13 * a.bs;
        a.T = a.Y + 9LL * 13 * a.bs;
        for (int k = 0; k < 20; ++k)
            (void)run(a, 13, a.bs, 4);
        unmap_at(w);
        CK(hipMemAddressFree(w, need * G));
    }
    /* (1) one big reservation, scanned */
    size_t span = (size_t)1100 * G;
    void *big = NULL;
    while (span >= 64 * G && hipMemAddressReserve(&big, span, G, NULL, 0) != hipSuccess) {
        (void)hipGetLastError();
        big = NULL;
        span /= 2;
    }
    printf("{\"reserved_gib\": %zu, \"at\": \"0x%llx\"}\n", span / G, (unsigned long long)big);
    std::vector<size_t> offs;
    for (size_t o = 0; o <= 40; ++o)
        offs.push_back(o);
    for (size_t o : {48, 64, 96, 128, 160, 192, 256, 320, 384, 512, 640, 768, 1024})
        offs.push_back(o);
    for (int pass = 0; pass < 2; ++pass)
        for (size_t o : offs) {
            if ((o + need) * G > span)
                continue;
            void *b = (char *)big + o * G;
            map_at(b);
            both(pass ? "scan2" : "scan1", (int)o, b);
            unmap_at(b);
        }
    /* (2) separate reservations, as an allocation gets them */
    std::vector<void *> keep;
    for (int k = 0; k < 12; ++k) {
        void *b = NULL;
        CK(hipMemAddressReserve(&b, need * G, G, NULL, 0));
        map_at(b);
        both("own_reservation", k, b);
        unmap_at(b);
        keep.push_back(b); /* held: the next one lands elsewhere */
    }
    for (void *b : keep)
        CK(hipMemAddressFree(b, need * G));
    /* (3) hipMalloc, several held at once */
    std::vector<void *> ms;
    for (int k = 0; k < 5; ++k) {
        void *p = NULL;
        if (hipMalloc(&p, need * G) != hipSuccess)
            break;
        both("hipMalloc", k, p);
        ms.push_back(p);
    }
    CK(hipStreamSynchronize(st));
    for (void *p : ms)
        CK(hipFree(p));
    return 0;
}
