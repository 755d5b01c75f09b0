// What the placement search pays for: hipMemCreate / hipMemMap /
// hipMemSetAccess / hipMemUnmap / hipMemRelease per piece size.
// hipcc --offload-arch=gfx950 -O2 -o create_lab create_lab.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() {
    return std::chrono::duration<double, std::milli>(
               std::chrono::steady_clock::now().time_since_epoch()).count();
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    CK(hipSetDevice(0));
    CK(hipFree(0));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = 0;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    const size_t sizes[] = {(size_t)1 << 29, (size_t)1 << 30, (size_t)4 << 30, (size_t)8 << 30};
    for (size_t sz : sizes) {
        double tc = 0, tm = 0, ta = 0, tu = 0, tr = 0;
        const int reps = 6;
        void *va = nullptr;
        CK(hipMemAddressReserve(&va, sz, sz < ((size_t)1 << 30) ? sz : (size_t)1 << 30, nullptr, 0));
        for (int k = 0; k < reps; ++k) {
            hipMemGenericAllocationHandle_t h;
            double t0 = now();
            CK(hipMemCreate(&h, sz, &prop, 0));
            double t1 = now();
            CK(hipMemMap(va, sz, 0, h, 0));
            double t2 = now();
            CK(hipMemSetAccess(va, sz, &acc, 1));
            double t3 = now();
            CK(hipMemUnmap(va, sz));
            double t4 = now();
            CK(hipMemRelease(h));
            double t5 = now();
            if (k) { tc += t1 - t0; tm += t2 - t1; ta += t3 - t2; tu += t4 - t3; tr += t5 - t4; }
        }
        CK(hipMemAddressFree(va, sz));
        printf("{\"MiB\": %zu, \"create_ms\": %.3f, \"map_ms\": %.3f, \"set_access_ms\": %.3f, \"unmap_ms\": %.3f, \"release_ms\": %.3f}\n",
               sz >> 20, tc / (reps - 1), tm / (reps - 1), ta / (reps - 1), tu / (reps - 1), tr / (reps - 1));
    }
    // held ballast: 12 blocks of 8 GiB created one after another, then released
    std::vector<hipMemGenericAllocationHandle_t> hs;
    double t0 = now();
    for (int k = 0; k < 12; ++k) {
        hipMemGenericAllocationHandle_t h;
        if (hipMemCreate(&h, (size_t)8 << 30, &prop, 0) != hipSuccess) break;
        hs.push_back(h);
    }
    double t1 = now();
    for (auto h : hs) (void)hipMemRelease(h);
    double t2 = now();
    printf("{\"ballast_blocks_8GiB\": %zu, \"create_all_ms\": %.1f, \"release_all_ms\": %.1f}\n", hs.size(), t1 - t0, t2 - t1);
    return 0;
}
