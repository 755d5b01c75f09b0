// D2H copies into pinned host memory by size: one at a time, two and four
// queued back to back, from hipMalloc'ed and from hipMemCreate'd memory.
// hipcc --offload-arch=gfx950 -O2 -o d2h_lab d2h_lab.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void copy_kernel(const double2 *__restrict__ src,
                            double2 *__restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x)
        dst[i] = src[i];
}

int main()
{
    const size_t big = (size_t)256 << 20;
    char *d = NULL, *h = NULL;
    CK(hipMalloc(&d, big));
    CK(hipMemset(d, 1, big));
    CK(hipHostMalloc(&h, big, 0));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipDeviceSynchronize());
    // pieces behind a reserved range, like the engine's arrays
    char *v = NULL;
    {
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        hipMemGenericAllocationHandle_t hd;
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        if (hipMemCreate(&hd, big, &prop, 0) == hipSuccess &&
            hipMemAddressReserve((void **)&v, big, (size_t)1 << 30, NULL, 0) == hipSuccess &&
            hipMemMap(v, big, 0, hd, 0) == hipSuccess &&
            hipMemSetAccess(v, big, &acc, 1) == hipSuccess) {
            CK(hipMemset(v, 2, big));
            CK(hipDeviceSynchronize());
        } else {
            v = NULL;
        }
    }
    const size_t sizes[] = {1, 4, 8, 16, 20, 32, 64, 128, 240};
    for (int src = 0; src < 2; ++src) {
        char *from = src ? v : d;
        if (!from)
            continue;
        for (size_t mb : sizes) {
            const size_t n = mb << 20;
            double t1 = 1e9, t2 = 1e9, t4 = 1e9, ts = 1e9;
            for (int r = 0; r < 6; ++r) {
                double t0 = now();
                CK(hipMemcpyAsync(h, from, n, hipMemcpyDeviceToHost, s));
                CK(hipStreamSynchronize(s));
                double t = now() - t0;
                t1 = t < t1 ? t : t1;
                t0 = now();
                CK(hipMemcpy(h, from, n, hipMemcpyDeviceToHost));
                t = now() - t0;
                ts = t < ts ? t : ts;
                if (2 * n <= big) {
                    t0 = now();
                    CK(hipMemcpyAsync(h, from, n, hipMemcpyDeviceToHost, s));
                    CK(hipMemcpyAsync(h + n, from + n, n, hipMemcpyDeviceToHost, s));
                    CK(hipStreamSynchronize(s));
                    t = now() - t0;
                    t2 = t < t2 ? t : t2;
                }
                if (4 * n <= big) {
                    t0 = now();
                    for (int k = 0; k < 4; ++k)
                        CK(hipMemcpyAsync(h + k * n, from + k * n, n, hipMemcpyDeviceToHost, s));
                    CK(hipStreamSynchronize(s));
                    t = now() - t0;
                    t4 = t < t4 ? t : t4;
                }
            }
            // the same copy waited for through an event (the engine's way)
            double te = 1e9, tq = 1e9, tb = 1e9;
            {
                hipEvent_t ev, evb;
                CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
                CK(hipEventCreateWithFlags(&evb, hipEventDisableTiming | hipEventBlockingSync));
                for (int r = 0; r < 6; ++r) {
                    double t0 = now();
                    CK(hipMemcpyAsync(h, from, n, hipMemcpyDeviceToHost, s));
                    CK(hipEventRecord(ev, s));
                    CK(hipEventSynchronize(ev));
                    double t = now() - t0;
                    te = t < te ? t : te;
                    t0 = now();
                    CK(hipMemcpyAsync(h, from, n, hipMemcpyDeviceToHost, s));
                    CK(hipEventRecord(ev, s));
                    while (hipEventQuery(ev) == hipErrorNotReady)
                        ;
                    t = now() - t0;
                    tq = t < tq ? t : tq;
                    t0 = now();
                    CK(hipMemcpyAsync(h, from, n, hipMemcpyDeviceToHost, s));
                    CK(hipEventRecord(evb, s));
                    CK(hipEventSynchronize(evb));
                    t = now() - t0;
                    tb = t < tb ? t : tb;
                }
                CK(hipEventDestroy(ev));
                CK(hipEventDestroy(evb));
            }
            // a copy KERNEL writing the pinned host buffer over PCIe
            double tk[4] = {1e9, 1e9, 1e9, 1e9};
            const int grids[4] = {64, 256, 1024, 4096};
            for (int gi = 0; gi < 4; ++gi)
                for (int r = 0; r < 6; ++r) {
                    double t0 = now();
                    hipLaunchKernelGGL(copy_kernel, dim3(grids[gi]), dim3(256), 0, s,
                                       (const double2 *)from, (double2 *)h, n / 16);
                    CK(hipStreamSynchronize(s));
                    double t = now() - t0;
                    tk[gi] = t < tk[gi] ? t : tk[gi];
                }
            printf("{\"kernel_copy_ms_grid_64_256_1024_4096\": [%.3f, %.3f, %.3f, %.3f], ", tk[0], tk[1], tk[2], tk[3]);
            printf("\"event_sync_ms\": %.3f, \"event_query_spin_ms\": %.3f, \"event_blocking_sync_ms\": %.3f, ", te, tq, tb);
            printf("\"source\": \"%s\", \"MB\": %zu, \"async_one_ms\": %.3f, \"GBps\": %.1f, \"sync_hipMemcpy_ms\": %.3f, \"two_queued_ms\": %.3f, \"four_queued_ms\": %.3f}\n",
                   src ? "hipMemCreate piece" : "hipMalloc", mb, t1, n / t1 / 1e6, ts, t2 < 1e8 ? t2 : -1., t4 < 1e8 ? t4 : -1.);
            fflush(stdout);
        }
    }
    return 0;
}
