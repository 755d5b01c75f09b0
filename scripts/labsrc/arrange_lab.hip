/*
 * arrange_lab.hip -- LABORATORY (round 5): what decides the speed of a
 * batch's store pattern once the arrays are built from measured pieces?
 *
 * Round 4 left C2 (Cooke triplet, 3 x 10^6 rays, 2.2 GB in five 512 MiB
 * pieces) bimodal -- 0.207 or 0.259 ms per trace with the SAME class mix
 * [3, 2, 0] -- and the bundles with per-ray launch directions at 1.07 or
 * 1.146 ms.  This program takes a pool of pieces, measures the pair matrix
 * the engine's classification rests on, and then times the batch's own
 * pattern (the engine's layout: blocks of Y | U | I | T planes; writes only,
 * and writes with the 48 B per ray of launch rows read first) over many
 * ARRANGEMENTS of pieces behind one address range:
 *   consecutive   need pieces as hipMemCreate handed them out
 *   engine        rt_place_alloc's choice from a window of the pool
 *   random        a random subset in random order
 *   permute       one fixed subset, random orders
 * Output: JSON lines.
 *   hipcc --offload-arch=gfx950 -O3 -o arrange_lab arrange_lab.hip
 *   ./arrange_lab L n nblk piece_mib pool trials [seed]
 */
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
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

struct lay {
    double *Y, *U, *T;
    long long bs, ts; /* rays per block, doubles between blocks (0: one) */
};

__device__ __forceinline__ long long col_of(const lay &a, long long j)
{
    if (!a.ts)
        return j;
    const unsigned long long b = (unsigned long long)j / (unsigned long long)a.bs;
    return (long long)b * a.ts + (j - (long long)b * a.bs);
}

/* the trace's rows: y0 y1 y2 u0 u1 u2 t of elements 1 .. L-1 */
template <int NT> __device__ __forceinline__ void put(double v, double *p)
{
    if (NT)
        __builtin_nontemporal_store(v, p);
    else
        *p = v;
}

template <int READ, int NT>
__global__ __launch_bounds__(256) void pattern(lay a, int L, long long n)
{
    const long long j = (long long)blockIdx.x * 256 + threadIdx.x;
    if (j >= n)
        return;
    const long long r = col_of(a, j);
    double v = 1e-9 * (double)j;
    if (READ) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
            v += a.Y[c * a.bs + r] + a.U[c * a.bs + r];
        v *= 1e-30;
    }
    for (int s = 1; s < L; ++s) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            put<NT>(v + c, &a.Y[(long long)s * 3 * a.bs + c * a.bs + r]);
            put<NT>(v - c, &a.U[(long long)s * 3 * a.bs + c * a.bs + r]);
        }
        put<NT>(v, &a.T[(long long)s * a.bs + r]);
    }
}

/* the engine's classification kernel: 84 short rows through pointers */
struct rows84 {
    double *row[84];
};
__global__ __launch_bounds__(256) void pair_kernel(rows84 tb, long long n)
{
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= n)
        return;
    const double a = 1e-9 * (double)r;
    for (int s = 0; s < 12; ++s) {
#pragma unroll
        for (int j = 0; j < 7; ++j)
            tb.row[s * 7 + j][r] = a + j;
    }
}

static hipStream_t st;
static hipEvent_t e0, e1;

template <typename F> static double median_ms(F launch, int reps, int blocks)
{
    for (int w = 0; w < 2; ++w)
        launch();
    std::vector<float> ms;
    for (int b = 0; b < blocks; ++b) {
        CK(hipEventRecord(e0, st));
        for (int k = 0; k < reps; ++k)
            launch();
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float t;
        CK(hipEventElapsedTime(&t, e0, e1));
        ms.push_back(t / reps);
    }
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
}

static unsigned long long rng_state = 0x9E3779B97F4A7C15ull;
static unsigned rnd(unsigned m)
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (unsigned)(rng_state % m);
}

int main(int argc, char **argv)
{
    if (argc < 7) {
        fprintf(stderr, "usage: %s L n nblk piece_mib pool trials [seed]\n", argv[0]);
        return 1;
    }
    const int L = atoi(argv[1]);
    const long long nrays = atoll(argv[2]);
    const int nblk = atoi(argv[3]);
    const size_t piece = (size_t)atoi(argv[4]) << 20;
    int pool = atoi(argv[5]);
    const int trials = atoi(argv[6]);
    if (argc > 7)
        rng_state ^= (unsigned long long)atoll(argv[7]) * 0x2545F4914F6CDD1Dull;
    /* "regions": the pool is (nearly) the whole device memory; no pair
     * matrix, every piece against three far-apart references instead, and
     * arrangements drawn from one window / two / three far-apart windows /
     * spread evenly over the pool */
    const bool regions = argc > 8 && !strcmp(argv[8], "regions");
    CK(hipSetDevice(0));
    CK(hipStreamCreate(&st));
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));

    /* the engine's layout */
    const long long bs = nblk > 1 ? ((nrays + nblk - 1) / nblk + 255) / 256 * 256
                                  : (nrays + 63) / 64 * 64;
    const long long ld = bs * nblk;
    const size_t total = (size_t)10 * L * ld * 8;
    const int need = (int)((total + piece - 1) / piece);
    const double alg = 56. * (L - 1) * (double)ld;
    printf("{\"L\": %d, \"rays\": %lld, \"nblk\": %d, \"bs\": %lld, \"piece_mib\": %zu, "
           "\"need\": %d, \"bytes\": %zu}\n", L, nrays, nblk, bs, piece >> 20, need, total);

    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = 0;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    std::vector<hipMemGenericAllocationHandle_t> H(pool);
    int got = 0;
    for (; got < pool; ++got)
        if (hipMemCreate(&H[got], piece, &prop, 0) != hipSuccess) {
            (void)hipGetLastError();
            break;
        }
    pool = got;
    void *scratch = NULL, *range = NULL;
    CK(hipMemAddressReserve(&scratch, (size_t)pool * piece, piece, NULL, 0));
    CK(hipMemAddressReserve(&range, (size_t)need * piece, piece, NULL, 0));
    for (int k = 0; k < pool; ++k)
        CK(hipMemMap((char *)scratch + (size_t)k * piece, piece, 0, H[k], 0));
    CK(hipMemSetAccess(scratch, (size_t)pool * piece, &acc, 1));
    auto base = [&](int c) { return (double *)((char *)scratch + (size_t)c * piece); };

    /* settle the clocks */
    const long long nprobe = (long long)(piece / 8 / 84) / 256 * 256;
    auto pair_ms = [&](int i, int j, int reps) {
        rows84 tb;
        for (int s = 0; s < 84; ++s)
            tb.row[s] = i == j ? base(i) + (long long)s * nprobe
                               : base(s < 42 ? i : j) + (long long)(s % 42) * nprobe;
        const unsigned grid = (unsigned)((nprobe + 255) / 256);
        return median_ms([&] { hipLaunchKernelGGL(pair_kernel, dim3(grid), dim3(256), 32768, st, tb, nprobe); },
                         reps, 3);
    };
    {
        const auto t0 = std::chrono::steady_clock::now();
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < .4)
            (void)pair_ms(0, 0, 20);
    }

    /* pair matrix and the engine's greedy classes */
    std::vector<float> self(pool);
    std::vector<std::vector<float>> M(pool, std::vector<float>(pool, 0.f));
    const int refs[3] = {0, pool / 3, 2 * pool / 3};
    for (int i = 0; i < pool; ++i) {
        self[i] = (float)pair_ms(i, i, regions ? 3 : 6);
        if (regions) {
            for (int q = 0; q < 3; ++q)
                if (refs[q] != i)
                    M[i][refs[q]] = M[refs[q]][i] = (float)pair_ms(i, refs[q], 3);
            printf("{\"part\": \"R\", \"i\": %d, \"self_ms\": %.4f, \"vs_refs\": [%.4f, %.4f, %.4f]}\n",
                   i, self[i], M[i][refs[0]], M[i][refs[1]], M[i][refs[2]]);
            continue;
        }
        for (int j = i + 1; j < pool; ++j)
            M[i][j] = M[j][i] = (float)pair_ms(i, j, 4);
        printf("{\"part\": \"P\", \"i\": %d, \"self_ms\": %.4f, \"pair_ms\": [", i, self[i]);
        for (int j = 0; j < pool; ++j)
            printf("%s%.4f", j ? ", " : "", j == i ? self[i] : M[i][j]);
        printf("]}\n");
        fflush(stdout);
    }
    std::vector<int> cls(pool, 0), rep;
    rep.push_back(0);
    for (int k = 1; k < pool && !regions; ++k) {
        int found = -1;
        for (size_t q = 0; q < rep.size() && found < 0; ++q)
            if (M[k][rep[q]] > .91f * self[0])
                found = (int)q;
        if (found < 0) {
            found = (int)rep.size();
            rep.push_back(k);
        }
        cls[k] = found;
    }
    printf("{\"part\": \"classes\", \"cls\": [");
    for (int k = 0; k < pool; ++k)
        printf("%s%d", k ? ", " : "", cls[k]);
    printf("]}\n");
    fflush(stdout);

    /* arrangements */
    lay a;
    a.bs = bs;
    a.ts = nblk > 1 ? (long long)10 * L * bs : 0;
    a.Y = (double *)range;
    a.U = a.Y + (size_t)3 * L * bs;
    a.T = a.Y + (size_t)9 * L * bs;
    const unsigned grid = (unsigned)((ld + 255) / 256);
    std::vector<int> fixed; /* the subset of the "permute" trials */
    for (int t = 0; t < trials; ++t) {
        const int kind = t % 4;
        std::vector<int> set;
        const int k0 = (int)rnd((unsigned)(pool - need + 1));
        if (regions) {
            /* kind 0: one window; 1: two windows a third of the pool apart,
             * interleaved; 2: three windows; 3: spread over the whole pool */
            const int nw = kind == 0 ? 1 : kind == 1 ? 2 : kind == 2 ? 3 : need;
            const int per = (need + nw - 1) / nw;
            const int span = pool - per;
            const int w0 = (int)rnd((unsigned)(span > 0 ? span : 1));
            for (int i = 0; i < need; ++i) {
                const int w = i % nw, k = i / nw;
                int c = kind == 3 ? (int)(((long long)i * pool) / need + w0 % (pool / need))
                                  : (w0 + w * (pool / nw) + k) % pool;
                while (std::find(set.begin(), set.end(), c) != set.end())
                    c = (c + 1) % pool;
                set.push_back(c);
            }
        } else if (kind == 0) {
            for (int i = 0; i < need; ++i)
                set.push_back(k0 + i);
        } else if (kind == 1) {
            /* rt_place_alloc without hops: pieces from k0 on until `need` can
             * be picked with no class above half, then round-robin */
            std::vector<int> cnt(rep.size(), 0);
            int made = 0;
            bool enough = false;
            while (k0 + made < pool && !enough) {
                ++cnt[cls[k0 + made]];
                ++made;
                if (made >= need) {
                    int can = 0, seen = 0;
                    for (size_t q = 0; q < cnt.size(); ++q) {
                        can += std::min(cnt[q], (need + 1) / 2);
                        seen += cnt[q] > 0;
                    }
                    enough = can >= need && seen >= 2;
                }
            }
            std::vector<int> next(rep.size(), k0);
            size_t q = 0;
            int idle = 0;
            while ((int)set.size() < need && idle < (int)rep.size()) {
                int k = next[q];
                while (k < k0 + made && cls[k] != (int)q)
                    ++k;
                if (k < k0 + made) {
                    set.push_back(k);
                    next[q] = k + 1;
                    idle = 0;
                } else {
                    next[q] = k0 + made;
                    ++idle;
                }
                q = (q + 1) % rep.size();
            }
            if ((int)set.size() < need) {
                set.clear();
                for (int i = 0; i < need; ++i)
                    set.push_back(k0 + i);
            }
            if (fixed.empty())
                fixed = set;
        } else if (kind == 2 || fixed.empty()) {
            while ((int)set.size() < need) {
                const int c = (int)rnd((unsigned)pool);
                if (std::find(set.begin(), set.end(), c) == set.end())
                    set.push_back(c);
            }
        } else {
            set = fixed;
            for (int i = need - 1; i > 0; --i)
                std::swap(set[i], set[rnd((unsigned)(i + 1))]);
        }
        for (int i = 0; i < need; ++i) {
            CK(hipMemUnmap((char *)scratch + (size_t)set[i] * piece, piece));
            CK(hipMemMap((char *)range + (size_t)i * piece, piece, 0, H[set[i]], 0));
        }
        CK(hipMemSetAccess(range, (size_t)need * piece, &acc, 1));
        const int reps = alg > 3e9 ? 4 : 12;
        const double w = median_ms([&] { hipLaunchKernelGGL((pattern<0, 0>), dim3(grid), dim3(256), 32768, st, a, L, ld); }, reps, 5);
        const double rw = median_ms([&] { hipLaunchKernelGGL((pattern<1, 0>), dim3(grid), dim3(256), 32768, st, a, L, ld); }, reps, 5);
        const double w2 = median_ms([&] { hipLaunchKernelGGL((pattern<0, 0>), dim3(grid), dim3(256), 65536, st, a, L, ld); }, reps, 3);
        const double wn = median_ms([&] { hipLaunchKernelGGL((pattern<0, 1>), dim3(grid), dim3(256), 32768, st, a, L, ld); }, reps, 5);
        const double rwn = median_ms([&] { hipLaunchKernelGGL((pattern<1, 1>), dim3(grid), dim3(256), 32768, st, a, L, ld); }, reps, 5);
        CK(hipStreamSynchronize(st));
        for (int i = 0; i < need; ++i) {
            CK(hipMemUnmap((char *)range + (size_t)i * piece, piece));
            CK(hipMemMap((char *)scratch + (size_t)set[i] * piece, piece, 0, H[set[i]], 0));
            CK(hipMemSetAccess((char *)scratch + (size_t)set[i] * piece, piece, &acc, 1));
        }
        printf("{\"part\": \"A\", \"kind\": \"%s\", \"pieces\": [",
               regions ? (kind == 0 ? "one_window" : kind == 1 ? "two_windows" : kind == 2 ? "three_windows" : "spread")
               : kind == 0 ? "consecutive" : kind == 1 ? "engine" : kind == 2 ? "random" : "permute");
        for (int i = 0; i < need; ++i)
            printf("%s%d", i ? ", " : "", set[i]);
        printf("], \"cls\": [");
        for (int i = 0; i < need; ++i)
            printf("%s%d", i ? ", " : "", cls[set[i]]);
        printf("], \"w_ms\": %.4f, \"rw_ms\": %.4f, \"w_two_per_cu_ms\": %.4f, "
               "\"w_nt_ms\": %.4f, \"rw_nt_ms\": %.4f, "
               "\"w_gbps\": %.0f, \"rw_gbps\": %.0f}\n",
               w, rw, w2, wn, rwn, alg / (w * 1e-3) / 1e9, (alg + 48. * ld) / (rw * 1e-3) / 1e9);
        fflush(stdout);
    }
    return 0;
}
