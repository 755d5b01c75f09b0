/*
 * stream_lab.hip -- LABORATORY (round 4, after chunk_lab): the store pattern
 * of the trace kernel with every one of its 84 row streams placed through a
 * TABLE of base pointers, so that streams can be dealt onto 1 GiB physical
 * chunks (hipMemCreate) in any way without re-mapping:
 *   k = 1   all 84 streams in one chunk
 *   k = 2,4 halves / quarters of the streams per chunk
 *   k = 7   12 streams per chunk (the engine's own density: 80 MB rows)
 *   k = 84  every stream in a chunk of its own
 * with n rays short enough (12 MB rows) that every variant fits, the same n
 * for all.  What decides 0.96 / 1.07 / 1.19 ms per 10^7 rays
 * (profiles/r04_probes: the store pattern per allocation)?
 * Output: JSON lines.  hipcc --offload-arch=gfx950 -O3 -o stream_lab stream_lab.hip
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

#define NSTREAM 84 /* 12 elements x (y0 y1 y2 u0 u1 u2 t) */

struct table {
    double *row[NSTREAM];
};

extern __shared__ double lab_lds[];

__global__ __launch_bounds__(256) void pattern(table tb, long long n)
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

static double run_ms(const table &tb, long long n, size_t lds, int reps)
{
    const unsigned grid = (unsigned)((n + 255) / 256);
    for (int w = 0; w < 2; ++w)
        hipLaunchKernelGGL(pattern, dim3(grid), dim3(256), lds, st, tb, n);
    std::vector<float> ms;
    for (int b = 0; b < 5; ++b) {
        CK(hipEventRecord(e0, st));
        for (int k = 0; k < reps; ++k)
            hipLaunchKernelGGL(pattern, dim3(grid), dim3(256), lds, st, tb, n);
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
    int NC = argc > 1 ? atoi(argv[1]) : 200;
    CK(hipSetDevice(0));
    CK(hipStreamCreate(&st));
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    const size_t chunk = (size_t)(argc > 4 ? atoi(argv[4]) : 1024) << 20;
    std::vector<hipMemGenericAllocationHandle_t> H(NC);
    int got = 0;
    for (; got < NC; ++got)
        if (hipMemCreate(&H[got], chunk, &prop, 0) != hipSuccess) {
            (void)hipGetLastError();
            break;
        }
    NC = got;
    void *big = NULL;
    CK(hipMemAddressReserve(&big, (size_t)NC * chunk, chunk, NULL, 0));
    for (int k = 0; k < NC; ++k)
        CK(hipMemMap((char *)big + (size_t)k * chunk, chunk, 0, H[k], 0));
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = 0;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(big, (size_t)NC * chunk, &acc, 1));
    printf("{\"chunks\": %d}\n", NC);
    auto base = [&](int c) { return (double *)((char *)big + (size_t)c * chunk); };

    const size_t LDS = 32768;
    /* short rows: 84 of them fit one chunk */
    /* 12 MB rows in a 1 GiB chunk: 84 of them + slack */
    const long long ns = (long long)(1500160.0 * ((double)chunk / (double)((size_t)1 << 30))) / 256 * 256;
    const double per1e7 = 1e7 / (double)ns;
    /* engine-sized rows (80 MB): 12 per chunk */
    const long long nl = 10000000;

    if (argc > 2 && !strcmp(argv[2], "pairs")) {
        /* every pair of the first NP chunks: 42 short rows in each */
        const int NP = argc > 3 ? atoi(argv[3]) : 48;
        for (int i = 0; i < NP && i < NC; ++i) {
            printf("{\"part\": \"P\", \"i\": %d, \"ms_per_1e7_vs_j\": [", i);
            for (int j = 0; j < NP && j < NC; ++j) {
                table tb;
                for (int s = 0; s < NSTREAM; ++s)
                    tb.row[s] = base(s < 42 ? i : j) + (long long)(s % 42 + (i == j && s >= 42 ? 42 : 0)) * ns;
                const double ms = j < i ? 0. : run_ms(tb, ns, LDS, 10);
                printf("%s%.3f", j ? ", " : "", ms * per1e7);
            }
            printf("]}\n");
            fflush(stdout);
        }
        /* triples / sets built from the pair matrix are left to the reader
         * of the matrix: part Q takes sets from argv */
        return 0;
    }
    /* --- A: engine-sized rows, 12 streams per chunk, random 7-sets ------ */
    for (int trial = 0; trial < 30; ++trial) {
        int set[7];
        for (int i = 0; i < 7; ++i) {
            bool dup;
            do {
                set[i] = (int)rnd(NC);
                dup = false;
                for (int j = 0; j < i; ++j)
                    dup = dup || set[j] == set[i];
            } while (dup);
        }
        if (trial < 6) /* the first six: consecutive chunks */
            for (int i = 0; i < 7; ++i)
                set[i] = trial * 7 + i;
        table tb;
        for (int s = 0; s < NSTREAM; ++s)
            tb.row[s] = base(set[s / 12]) + (long long)(s % 12) * nl;
        const double ms = run_ms(tb, nl, LDS, 6);
        /* the same chunks, streams dealt round-robin instead of in blocks */
        for (int s = 0; s < NSTREAM; ++s)
            tb.row[s] = base(set[s % 7]) + (long long)(s / 7) * nl;
        const double ms_rr = run_ms(tb, nl, LDS, 6);
        printf("{\"part\": \"A\", \"trial\": %d, \"set\": [%d,%d,%d,%d,%d,%d,%d], "
               "\"blocked_ms\": %.4f, \"round_robin_ms\": %.4f}\n",
               trial, set[0], set[1], set[2], set[3], set[4], set[5], set[6], ms, ms_rr);
        fflush(stdout);
    }
    /* --- B: short rows; k chunks share the 84 streams ------------------- */
    const int KS[] = {1, 2, 3, 4, 6, 7, 12, 14, 21, 28, 42, 84};
    for (int ki = 0; ki < 12; ++ki) {
        const int k = KS[ki];
        const int per = NSTREAM / k;
        for (int trial = 0; trial < (k == 1 ? 12 : 8); ++trial) {
            std::vector<int> set(k);
            for (int i = 0; i < k; ++i) {
                bool dup;
                do {
                    set[i] = (int)rnd(NC);
                    dup = false;
                    for (int j = 0; j < i; ++j)
                        dup = dup || set[j] == set[i];
                } while (dup);
            }
            table tb;
            for (int s = 0; s < NSTREAM; ++s)
                tb.row[s] = base(set[s / per]) + (long long)(s % per) * ns;
            const double ms = run_ms(tb, ns, LDS, 20);
            /* same chunks, every stream at the SAME offsets pattern but
             * packed to the chunk's start with 2 MiB-aligned row starts */
            printf("{\"part\": \"B\", \"k\": %d, \"trial\": %d, \"first_chunks\": [%d,%d], "
                   "\"ms\": %.5f, \"ms_per_1e7\": %.4f}\n",
                   k, trial, set[0], set[k > 1 ? 1 : 0], ms, ms * per1e7);
            fflush(stdout);
        }
    }
    /* --- C: k = 84, one stream per chunk, at different in-chunk offsets -- */
    for (int var = 0; var < 4; ++var) {
        for (int trial = 0; trial < 4; ++trial) {
            table tb;
            const int first = (int)rnd(NC - NSTREAM);
            for (int s = 0; s < NSTREAM; ++s) {
                long long off = 0;
                if (var == 1)
                    off = (long long)s * ns;                /* as if in one chunk */
                else if (var == 2)
                    off = (long long)(s * 1237 % 97) * 262144; /* scattered 2 MiB steps */
                else if (var == 3)
                    off = (long long)s * 512;               /* 4 KiB steps */
                if ((off + ns) * 8 > (long long)chunk)
                    off = 0;
                tb.row[s] = base(first + s) + off;
            }
            const double ms = run_ms(tb, ns, LDS, 20);
            printf("{\"part\": \"C\", \"offsets\": \"%s\", \"first_chunk\": %d, \"ms\": %.5f, "
                   "\"ms_per_1e7\": %.4f}\n",
                   var == 0 ? "zero" : var == 1 ? "s*row" : var == 2 ? "scattered_2MiB" : "s*4KiB",
                   first, ms, ms * per1e7);
            fflush(stdout);
        }
    }
    /* --- D: one chunk, 84 short rows at different spacings -------------- */
    for (int trial = 0; trial < 3; ++trial) {
        const int c = (int)rnd(NC);
        const long long spacings[] = {ns, ns + 512, ns + 4096, ns + 32768, ns + 65536};
        for (int v = 0; v < 5; ++v) {
            if ((spacings[v] * (NSTREAM - 1) + ns) * 8 > (long long)chunk)
                continue;
            table tb;
            for (int s = 0; s < NSTREAM; ++s)
                tb.row[s] = base(c) + (long long)s * spacings[v];
            const double ms = run_ms(tb, ns, LDS, 20);
            printf("{\"part\": \"D\", \"chunk\": %d, \"row_spacing_bytes\": %lld, \"ms\": %.5f, "
                   "\"ms_per_1e7\": %.4f}\n", c, spacings[v] * 8, ms, ms * per1e7);
        }
        fflush(stdout);
    }
    return 0;
}
