/*
 * chunk_lab.hip -- LABORATORY (round 4): is the "good / bad allocation" of
 * profiles/r03_probes/README.md ("Placement") a property of the PHYSICAL
 * memory chunks behind the arrays?
 *
 * A synthetic kernel with the trace kernel's store pattern (C3: 12 elements,
 * per element y[3] u[3] t = 7 rows of 8-byte stores, one ray per lane,
 * 256-thread workgroups, arrays Y U I T laid out like rt_reserve does) and an
 * optional FP64 FMA filler, run
 *   part 1  on NA plain hipMalloc allocations of the engine's size,
 *   part 2  on virtual ranges backed by 1 GiB hipMemCreate chunks, whole
 *           pattern per allocation and a scaled-down pattern per CHUNK,
 *   part 3  on ranges assembled from the fastest / the slowest chunks,
 * each at two and at four resident workgroups per CU (unused dynamic LDS).
 * Output: JSON lines.   hipcc --offload-arch=gfx950 -O3 -o chunk_lab chunk_lab.hip
 */
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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

static const int L = 13; /* C3: object + 12 propagated elements */

extern __shared__ double lab_lds[];

/* arrays like rt_reserve: Y,U,I [L][3][ld], T [L][ld] in one range */
__global__ __launch_bounds__(256) void pattern(double *base, long long ld,
                                               long long n, int fl)
{
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= n)
        return;
    double *Y = base, *U = base + (long long)L * 3 * ld,
           *T = base + (long long)L * 9 * ld;
    double a = 1e-9 * (double)r, b = a + 1., c = a + 2., d = a + 3.;
    for (int s = 1; s < L; ++s) {
        for (int k = 0; k < fl; k += 4) {
            a = __builtin_fma(a, 1.0000001, 1e-9);
            b = __builtin_fma(b, 0.9999999, 1e-9);
            c = __builtin_fma(c, 1.0000002, -1e-9);
            d = __builtin_fma(d, 0.9999998, 1e-9);
        }
        for (int j = 0; j < 3; ++j) {
            Y[((long long)s * 3 + j) * ld + r] = a + j;
            U[((long long)s * 3 + j) * ld + r] = b + c * j;
        }
        T[(long long)s * ld + r] = d;
    }
}

static hipStream_t st;
static hipEvent_t e0, e1;

static double run_ms(double *base, long long ld, long long n, int fl,
                     size_t lds, int reps)
{
    const unsigned grid = (unsigned)((n + 255) / 256);
    for (int w = 0; w < 3; ++w)
        hipLaunchKernelGGL(pattern, dim3(grid), dim3(256), lds, st, base, ld,
                           n, fl);
    std::vector<float> ms;
    for (int b = 0; b < 5; ++b) {
        CK(hipEventRecord(e0, st));
        for (int k = 0; k < reps; ++k)
            hipLaunchKernelGGL(pattern, dim3(grid), dim3(256), lds, st, base,
                               ld, n, fl);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float t;
        CK(hipEventElapsedTime(&t, e0, e1));
        ms.push_back(t / reps);
    }
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
}

struct vrange {
    void *base;
    size_t bytes;
};

static size_t gran_, chunk_;
static hipMemAllocationProp prop_;

static vrange map_chunks(const std::vector<hipMemGenericAllocationHandle_t> &h,
                         size_t align)
{
    vrange v;
    v.bytes = h.size() * chunk_;
    CK(hipMemAddressReserve(&v.base, v.bytes, align, NULL, 0));
    for (size_t k = 0; k < h.size(); ++k)
        CK(hipMemMap((char *)v.base + k * chunk_, chunk_, 0, h[k], 0));
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = 0;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(v.base, v.bytes, &acc, 1));
    return v;
}

static void unmap(vrange v)
{
    CK(hipMemUnmap(v.base, v.bytes));
    CK(hipMemAddressFree(v.base, v.bytes));
}

int main(int argc, char **argv)
{
    const int NA = argc > 1 ? atoi(argv[1]) : 6;      /* hipMalloc allocations */
    int NCv = argc > 2 ? atoi(argv[2]) : 44;     /* 1 GiB chunks */
    const int &NC = NCv;
    const int chunk_mb = argc > 3 ? atoi(argv[3]) : 1024;
    CK(hipSetDevice(0));
    CK(hipStreamCreate(&st));
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const long long n = 10000000, ld = n; /* 10^7 is a multiple of 64 */
    const size_t bytes = (size_t)L * 10 * ld * sizeof(double);
    const int FL[2] = {0, 128};
    const size_t LDS[2] = {65536, 32768};

    /* part 1: plain hipMalloc, all kept alive */
    std::vector<void *> plain;
    for (int a = 0; a < NA; ++a) {
        void *p;
        CK(hipMalloc(&p, bytes));
        plain.push_back(p);
        for (int f = 0; f < 2; ++f) {
            double ms[2];
            for (int q = 0; q < 2; ++q)
                ms[q] = run_ms((double *)p, ld, n, FL[f], LDS[q], 10);
            const double s4 = run_ms((double *)p, ld, n, FL[f], LDS[1], 1);
            printf("{\"part\": 1, \"alloc\": %d, \"address\": \"%p\", \"fma_per_op\": %d, "
                   "\"two_per_cu_ms\": %.4f, \"four_per_cu_ms\": %.4f, \"four_single_launch_ms\": %.4f}\n",
                   a, p, FL[f], ms[0], ms[1], s4);
            fflush(stdout);
        }
    }
    for (void *p : plain)
        CK(hipFree(p));

    /* part 2: 1 GiB chunks */
    memset(&prop_, 0, sizeof prop_);
    prop_.type = hipMemAllocationTypePinned;
    prop_.location.type = hipMemLocationTypeDevice;
    prop_.location.id = 0;
    CK(hipMemGetAllocationGranularity(&gran_, &prop_,
                                      hipMemAllocationGranularityRecommended));
    chunk_ = ((size_t)chunk_mb << 20);
    chunk_ = (chunk_ + gran_ - 1) / gran_ * gran_;
    const size_t per = (bytes + chunk_ - 1) / chunk_; /* chunks per allocation */
    std::vector<hipMemGenericAllocationHandle_t> H(NC);
    {
        int got = 0;
        for (; got < NC; ++got)
            if (hipMemCreate(&H[got], chunk_, &prop_, 0) != hipSuccess) {
                (void)hipGetLastError();
                break;
            }
        const_cast<int &>(NC) = got;
        H.resize(got);
    }
    printf("{\"part\": 2, \"granularity\": %zu, \"chunk_bytes\": %zu, "
           "\"chunks\": %d, \"chunks_per_allocation\": %zu}\n",
           gran_, chunk_, NC, per);

    /* one address range for all chunks, chunk k at k * chunk_ */
    void *big = NULL;
    CK(hipMemAddressReserve(&big, (size_t)NC * chunk_, chunk_, NULL, 0));
    for (int k = 0; k < NC; ++k)
        CK(hipMemMap((char *)big + (size_t)k * chunk_, chunk_, 0, H[k], 0));
    {
        hipMemAccessDesc acc = {};
        acc.location.type = hipMemLocationTypeDevice;
        acc.location.id = 0;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        CK(hipMemSetAccess(big, (size_t)NC * chunk_, &acc, 1));
    }
    /* 2a: every chunk alone: the same 40-row structure scaled into the chunk;
     * back to back and one launch at a time (serialised, like under --pmc) */
    const long long ldc = (long long)(chunk_ / sizeof(double) / (L * 10)) / 64 * 64;
    std::vector<double> c4(NC), c2(NC);
    for (int k = 0; k < NC; ++k) {
        double *base = (double *)((char *)big + (size_t)k * chunk_);
        double m[2][2];
        for (int f = 0; f < 2; ++f)
            for (int q = 0; q < 2; ++q)
                m[f][q] = run_ms(base, ldc, ldc, FL[f], LDS[q], 20);
        const double ser = run_ms(base, ldc, ldc, FL[1], LDS[1], 1);
        c2[k] = m[1][0];
        c4[k] = m[1][1];
        printf("{\"part\": \"2a\", \"chunk\": %d, \"rays\": %lld, "
               "\"store_only_two_four_ms\": [%.5f, %.5f], "
               "\"with_fma_two_four_ms\": [%.5f, %.5f], \"with_fma_four_single_launch_ms\": %.5f}\n",
               k, ldc, m[0][0], m[0][1], m[1][0], m[1][1], ser);
        fflush(stdout);
    }
    /* 2b: allocations of `per` consecutive chunks, whole pattern */
    for (size_t a = 0; (a + 1) * per <= (size_t)NC && a < 12; ++a) {
        double *base = (double *)((char *)big + a * per * chunk_);
        for (int f = 0; f < 2; ++f) {
            const double m2 = run_ms(base, ld, n, FL[f], LDS[0], 10);
            const double m4 = run_ms(base, ld, n, FL[f], LDS[1], 10);
            const double s4 = run_ms(base, ld, n, FL[f], LDS[1], 1);
            double mean4 = 0;
            for (size_t k = a * per; k < (a + 1) * per; ++k)
                mean4 += c4[k] / per;
            printf("{\"part\": \"2b\", \"alloc\": %zu, \"first_chunk\": %zu, \"fma_per_op\": %d, "
                   "\"two_per_cu_ms\": %.4f, \"four_per_cu_ms\": %.4f, \"four_single_launch_ms\": %.4f, "
                   "\"mean_chunk_four_ms\": %.5f}\n",
                   a, a * per, FL[f], m2, m4, s4, mean4);
        }
        fflush(stdout);
    }
    /* part 3: the fastest and the slowest chunks (by their own four-per-CU
     * time with the FMA filler) mapped a second time into one range each */
    std::vector<int> order(NC);
    for (int k = 0; k < NC; ++k)
        order[k] = k;
    std::sort(order.begin(), order.end(),
              [&](int x, int y) { return c4[x] < c4[y]; });
    for (int which = 0; which < 3; ++which) {
        void *rg = NULL;
        if (hipMemAddressReserve(&rg, per * chunk_, chunk_, NULL, 0) != hipSuccess)
            break;
        bool ok = true;
        for (size_t k = 0; k < per && ok; ++k) {
            const size_t idx = which == 0 ? k
                               : which == 1 ? NC - 1 - k
                                            : (NC / 2 - per / 2 + k);
            ok = hipMemMap((char *)rg + k * chunk_, chunk_, 0, H[order[idx]], 0) == hipSuccess;
        }
        hipMemAccessDesc acc = {};
        acc.location.type = hipMemLocationTypeDevice;
        acc.location.id = 0;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        ok = ok && hipMemSetAccess(rg, per * chunk_, &acc, 1) == hipSuccess;
        if (!ok) {
            printf("{\"part\": 3, \"error\": \"a chunk cannot be mapped twice: %s\"}\n",
                   hipGetErrorString(hipGetLastError()));
            break;
        }
        for (int f = 0; f < 2; ++f) {
            const double m2 = run_ms((double *)rg, ld, n, FL[f], LDS[0], 10);
            const double m4 = run_ms((double *)rg, ld, n, FL[f], LDS[1], 10);
            const double s4 = run_ms((double *)rg, ld, n, FL[f], LDS[1], 1);
            printf("{\"part\": 3, \"chunks\": \"%s\", \"fma_per_op\": %d, "
                   "\"two_per_cu_ms\": %.4f, \"four_per_cu_ms\": %.4f, \"four_single_launch_ms\": %.4f}\n",
                   which == 0 ? "fastest" : which == 1 ? "slowest" : "median",
                   FL[f], m2, m4, s4);
        }
        fflush(stdout);
    }
    for (int k = 0; k < NC; ++k)
        CK(hipMemRelease(H[k]));
    return 0;
}
