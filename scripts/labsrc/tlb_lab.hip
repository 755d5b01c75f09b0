/*
 * tlb_lab.hip -- LABORATORY (round 5): does a kernel see the NEW memory after
 * an address range has been unmapped and mapped again with other physical
 * pieces?  (rt_place_settle moved arrays between ranges and sets of pieces
 * and a trace then returned rows that the downloads did not find: session
 * 17.)  Pieces A are mapped behind range R and filled by a kernel with
 * pattern 1; R is unmapped and pieces B are mapped behind the SAME addresses
 * (and, second variant, behind a range that was freed and reserved again);
 * a kernel fills R with pattern 2; then R is read back three ways -- by a
 * kernel (sum), by hipMemcpy, and A is mapped elsewhere and checked to still
 * hold pattern 1.
 *   hipcc --offload-arch=gfx950 -O3 -o tlb_lab tlb_lab.hip
 */
#include <hip/hip_runtime.h>
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

__global__ void fill(double *p, long long n, double v)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x)
        __builtin_nontemporal_store(__builtin_fma(1e-9, (double)(i & 1023), v), &p[i]);
}

__global__ void count_not(const double *p, long long n, double v, unsigned long long *bad)
{
    unsigned long long b = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x)
        b += p[i] != __builtin_fma(1e-9, (double)(i & 1023), v);
    if (b)
        atomicAdd(bad, b);
}

static hipMemAllocationProp prop;
static hipMemAccessDesc acc;
static const size_t G = (size_t)1 << 30;

static void map(void *base, std::vector<hipMemGenericAllocationHandle_t> &h)
{
    for (size_t k = 0; k < h.size(); ++k)
        CK(hipMemMap((char *)base + k * G, G, 0, h[k], 0));
    CK(hipMemSetAccess(base, h.size() * G, &acc, 1));
}

static unsigned long long kernel_bad(double *p, long long n, double v, unsigned long long *d_bad)
{
    CK(hipMemset(d_bad, 0, 8));
    hipLaunchKernelGGL(count_not, dim3(2048), dim3(256), 0, 0, p, n, v, d_bad);
    unsigned long long b = 0;
    CK(hipMemcpy(&b, d_bad, 8, hipMemcpyDeviceToHost));
    return b;
}

static unsigned long long copy_bad(double *p, long long n, double v)
{
    /* every 4097th double through hipMemcpy of 1 MiB windows */
    std::vector<double> h(131072);
    unsigned long long b = 0;
    for (long long off = 0; off + 131072 <= n; off += 131072 * 61) {
        CK(hipMemcpy(h.data(), p + off, 131072 * 8, hipMemcpyDeviceToHost));
        for (long long i = 0; i < 131072; i += 97)
            b += h[i] != __builtin_fma(1e-9, (double)((off + i) & 1023), v);
    }
    return b;
}

int main(int argc, char **argv)
{
    const int only = argc > 1 ? atoi(argv[1]) : -1;
    CK(hipSetDevice(0));
    prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = 0;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    const int NP = 4;
    const long long n = (long long)NP * (G / 8);
    std::vector<hipMemGenericAllocationHandle_t> A(NP), B(NP);
    for (int k = 0; k < NP; ++k) {
        CK(hipMemCreate(&A[k], G, &prop, 0));
        CK(hipMemCreate(&B[k], G, &prop, 0));
    }
    unsigned long long *d_bad;
    CK(hipMalloc(&d_bad, 8));
    for (int variant = 0; variant < 7; ++variant) {
        if (only >= 0 && variant != only)
            continue;
        for (int rep = 0; rep < (only >= 0 ? 2 : 1); ++rep) {
            void *R = NULL, *S = NULL;
            (void)S;
            CK(hipMemAddressReserve(&R, NP * G, G, NULL, 0));
            map(R, A);
            hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, (double *)R, n, 1.);
            if (variant != 2)
                CK(hipDeviceSynchronize());
            /* (variant 2: NO synchronisation before the unmap) */
            CK(hipMemUnmap(R, NP * G));
            void *R2 = R;
            if (variant == 1 || variant == 2) { /* the range goes back and is reserved again */
                CK(hipMemAddressFree(R, NP * G));
                CK(hipMemAddressReserve(&R2, NP * G, G, NULL, 0));
            }
            map(R2, B);
            if (variant == 3) { /* a plain allocation comes and goes */
                void *t = NULL;
                CK(hipMalloc(&t, (size_t)64 << 20));
                CK(hipMemset(t, 0, (size_t)64 << 20));
                CK(hipDeviceSynchronize());
                CK(hipFree(t));
            } else if (variant == 4) { /* a large one */
                void *t = NULL;
                CK(hipMalloc(&t, (size_t)2 << 30));
                CK(hipFree(t));
            } else if (variant == 5) { /* the access rights set once more */
                CK(hipMemSetAccess(R2, NP * G, &acc, 1));
                CK(hipDeviceSynchronize());
            } else if (variant == 6) { /* a kernel that streams through 16 GiB
                                          of other memory (evictions) */
                void *t = NULL;
                CK(hipMalloc(&t, (size_t)16 << 30));
                hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, (double *)t,
                                   (long long)2 << 30, 3.);
                CK(hipDeviceSynchronize());
                CK(hipFree(t));
            }
            hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, (double *)R2, n, 2.);
            CK(hipDeviceSynchronize());
            /* what the kernel / a copy find behind the range now */
            const unsigned long long kb = kernel_bad((double *)R2, n, 2., d_bad);
            const unsigned long long cb = copy_bad((double *)R2, n, 2.);
            /* ... and what the pieces themselves hold, each set behind a
             * range nobody has used before */
            void *T = NULL;
            CK(hipMemUnmap(R2, NP * G));
            CK(hipMemAddressReserve(&S, NP * G, G, NULL, 0));
            CK(hipMemAddressReserve(&T, NP * G, G, NULL, 0));
            map(S, A);
            map(T, B);
            const unsigned long long a1 = kernel_bad((double *)S, n, 1., d_bad);
            const unsigned long long a2 = kernel_bad((double *)S, n, 2., d_bad);
            const unsigned long long b2 = kernel_bad((double *)T, n, 2., d_bad);
            const unsigned long long a1c = copy_bad((double *)S, n, 1.);
            const unsigned long long b2c = copy_bad((double *)T, n, 2.);
            printf("{\"variant\": \"%s\", \"rep\": %d, \"same_address\": %s, "
                   "\"behind_the_range_not_pattern2\": {\"kernel\": %llu, \"copy_of_%d_samples\": %llu}, "
                   "\"old_pieces_A\": {\"not_pattern1_kernel\": %llu, \"not_pattern2_kernel\": %llu, \"not_pattern1_copy\": %llu}, "
                   "\"new_pieces_B\": {\"not_pattern2_kernel\": %llu, \"not_pattern2_copy\": %llu}, \"doubles\": %lld}\n",
                   variant == 0 ? "unmap, map other pieces behind the same range"
                   : variant == 1 ? "range freed and reserved again"
                   : variant == 2 ? "freed and reserved again, no synchronisation before the unmap"
                   : variant == 3 ? "same range; a 64 MiB hipMalloc + hipFree before the kernel"
                   : variant == 4 ? "same range; a 2 GiB hipMalloc + hipFree"
                   : variant == 5 ? "same range; hipMemSetAccess once more"
                                  : "same range; 16 GiB of other memory written first",
                   rep, R2 == R ? "true" : "false", kb, 90584, cb, a1, a2, a1c, b2, b2c, n);
            fflush(stdout);
            CK(hipMemUnmap(T, NP * G));
            CK(hipMemAddressFree(T, NP * G));
            R2 = NULL;
            CK(hipMemUnmap(S, NP * G));
            CK(hipMemAddressFree(S, NP * G));
        }
    }
    return 0;
}
