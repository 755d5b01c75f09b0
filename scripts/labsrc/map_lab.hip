/*
 * map_lab.hip -- LABORATORY (round 5): HOW the arrays are mapped, everything
 * else held still.  state_lab showed the C3 / C2 store patterns at 0.984 /
 * 0.197 ms on ten fresh 1 GiB pieces mapped once and at 1.205 / 0.2405 ms on
 * a plain hipMalloc of the same size, in the same process, stable over 45 s;
 * arrange_lab, which maps every piece into a scratch range first (as
 * rt_place_alloc does to classify it) and re-maps it afterwards, ran at the
 * hipMalloc level on the same kind of pieces.  Variants, each timed with both
 * patterns, the whole list three times over:
 *   fresh1g      ten 1 GiB pieces, created, mapped once
 *   fresh512     twenty 512 MiB pieces, mapped once
 *   fresh2m      1 GiB pieces behind a range that is only 2 MiB aligned and
 *                starts 514 MiB into a 1 GiB frame (virtual and physical
 *                offsets inside a piece differ)
 *   remap_new    pieces mapped into a scratch range, written, unmapped,
 *                mapped into a NEW range (what rt_place_alloc does)
 *   remap_same   ... unmapped and mapped again into the SAME range
 *   one10g       one hipMemCreate of 10 GiB
 *   malloc       hipMalloc
 *   malloc_2nd   a second hipMalloc while the first is still held
 * Output: JSON lines.  hipcc --offload-arch=gfx950 -O3 -o map_lab map_lab.hip
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
static hipMemAllocationProp prop;
static hipMemAccessDesc acc;

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

static void both(const char *name, int round, void *base)
{
    lay a;
    a.bs = 5000192;
    a.ts = 10LL * 13 * a.bs;
    a.Y = (double *)base;
    a.U = a.Y + 3LL * 13 * a.bs;
    a.T = a.Y + 9LL * 13 * a.bs;
    const double c3 = run(a, 13, 2 * a.bs, 4);
    a.bs = 3000000;
    a.ts = 0;
    a.U = a.Y + 3LL * 9 * a.bs;
    a.T = a.Y + 9LL * 9 * a.bs;
    const double c2 = run(a, 9, a.bs, 16);
    printf("{\"variant\": \"%s\", \"round\": %d, \"c3_ms\": %.4f, \"c2_ms\": %.4f}\n",
           name, round, c3, c2);
    fflush(stdout);
}

struct mapped {
    void *base = NULL, *reserve = NULL;
    size_t bytes = 0, reserved = 0;
    std::vector<hipMemGenericAllocationHandle_t> h;
    size_t piece = 0;
};

static mapped make(size_t piece, int n, size_t align, size_t offset)
{
    mapped m;
    m.piece = piece;
    m.bytes = piece * n;
    m.reserved = m.bytes + offset;
    CK(hipMemAddressReserve(&m.reserve, m.reserved, align, NULL, 0));
    m.base = (char *)m.reserve + offset;
    m.h.resize(n);
    for (int k = 0; k < n; ++k) {
        CK(hipMemCreate(&m.h[k], piece, &prop, 0));
        CK(hipMemMap((char *)m.base + k * piece, piece, 0, m.h[k], 0));
    }
    CK(hipMemSetAccess(m.base, m.bytes, &acc, 1));
    return m;
}

static void drop(mapped &m)
{
    CK(hipStreamSynchronize(st));
    CK(hipMemUnmap(m.base, m.bytes));
    for (auto h : m.h)
        CK(hipMemRelease(h));
    CK(hipMemAddressFree(m.reserve, m.reserved));
}

int main()
{
    CK(hipSetDevice(0));
    CK(hipStreamCreate(&st));
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = 0;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    const size_t G = (size_t)1 << 30;
    {
        size_t gran = 0;
        CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
        printf("{\"granularity_recommended\": %zu}\n", gran);
    }
    for (int round = 0; round < 3; ++round) {
        {
            mapped m = make(G, 10, G, 0);
            both("fresh1g", round, m.base);
            drop(m);
        }
        {
            mapped m = make(G / 2, 20, G / 2, 0);
            both("fresh512", round, m.base);
            drop(m);
        }
        {
            mapped m = make(G, 10, G, ((size_t)514) << 20);
            both("fresh2m_offset514MiB", round, m.base);
            drop(m);
        }
        {
            /* scratch first, kernels write there, then a new range */
            mapped m = make(G, 10, G, 0);
            both("scratch_before_remap", round, m.base);
            CK(hipStreamSynchronize(st));
            CK(hipMemUnmap(m.base, m.bytes));
            void *nb = NULL;
            CK(hipMemAddressReserve(&nb, m.bytes, G, NULL, 0));
            for (int k = 0; k < 10; ++k)
                CK(hipMemMap((char *)nb + k * G, G, 0, m.h[k], 0));
            CK(hipMemSetAccess(nb, m.bytes, &acc, 1));
            both("remap_new", round, nb);
            CK(hipStreamSynchronize(st));
            /* ... and once more into the same range, other order */
            CK(hipMemUnmap(nb, m.bytes));
            for (int k = 0; k < 10; ++k)
                CK(hipMemMap((char *)nb + k * G, G, 0, m.h[9 - k], 0));
            CK(hipMemSetAccess(nb, m.bytes, &acc, 1));
            both("remap_same_range_reversed", round, nb);
            CK(hipStreamSynchronize(st));
            /* piece by piece: unmap one, map one (arrange_lab's loop) */
            for (int k = 0; k < 10; ++k) {
                CK(hipMemUnmap((char *)nb + k * G, G));
                CK(hipMemMap((char *)m.base + k * G, G, 0, m.h[9 - k], 0));
            }
            CK(hipMemSetAccess(m.base, m.bytes, &acc, 1));
            both("remap_back_piecewise", round, m.base);
            CK(hipStreamSynchronize(st));
            CK(hipMemUnmap(m.base, m.bytes));
            for (auto h : m.h)
                CK(hipMemRelease(h));
            CK(hipMemAddressFree(m.reserve, m.reserved));
            CK(hipMemAddressFree(nb, m.bytes));
        }
        {
            mapped m = make(10 * G, 1, G, 0);
            both("one10g", round, m.base);
            drop(m);
        }
        {
            void *p = NULL, *q = NULL;
            CK(hipMalloc(&p, 10 * G));
            both("malloc", round, p);
            CK(hipMalloc(&q, 10 * G));
            both("malloc_2nd", round, q);
            CK(hipStreamSynchronize(st));
            CK(hipFree(p));
            CK(hipFree(q));
        }
        {
            /* 512 MiB pieces for the C2-sized array alone (5 pieces) */
            mapped m = make(G / 2, 5, G / 2, 0);
            lay a;
            a.bs = 3000000;
            a.ts = 0;
            a.Y = (double *)m.base;
            a.U = a.Y + 3LL * 9 * a.bs;
            a.T = a.Y + 9LL * 9 * a.bs;
            printf("{\"variant\": \"c2_only_5x512\", \"round\": %d, \"c2_ms\": %.4f}\n", round,
                   run(a, 9, a.bs, 16));
            drop(m);
            mapped m2 = make(G, 3, G, 0);
            a.Y = (double *)m2.base;
            a.U = a.Y + 3LL * 9 * a.bs;
            a.T = a.Y + 9LL * 9 * a.bs;
            printf("{\"variant\": \"c2_only_3x1g\", \"round\": %d, \"c2_ms\": %.4f}\n", round,
                   run(a, 9, a.bs, 16));
            drop(m2);
            void *p = NULL;
            CK(hipMalloc(&p, (size_t)2160000000));
            a.Y = (double *)p;
            a.U = a.Y + 3LL * 9 * a.bs;
            a.T = a.Y + 9LL * 9 * a.bs;
            printf("{\"variant\": \"c2_only_malloc\", \"round\": %d, \"c2_ms\": %.4f}\n", round,
                   run(a, 9, a.bs, 16));
            CK(hipStreamSynchronize(st));
            CK(hipFree(p));
        }
    }
    return 0;
}
