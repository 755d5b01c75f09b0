/*
 * rt_probes.hip -- LABORATORY.  Compiled only with -DRT_BUILD_PROBES into
 * librt_mi355_probes.so (rayopt_amd/_build.py: build_probes), which the
 * measurement scripts load through RT_MI355_LIB; the shipped librt_mi355.so
 * contains none of it.  Rejected variants of the trace kernel (2 / 4 rays per
 * lane, non-temporal stores, XCD-contiguous dealing, workgroup size, unused
 * LDS to cap occupancy, tile-major result layout, gated and fake-uniform
 * input reads) and the memory-system probes (rt_probe) that calibrate the
 * ceiling the trace kernel is judged against.  What they showed:
 * profiles/HISTORY.md, DESIGN.md section 3.
 */
#ifndef RT_BUILD_PROBES
#error "rt_probes.hip is part of the laboratory build (-DRT_BUILD_PROBES)"
#endif

#include "rt_ctx.h"
#include "rt_probe_kernels.h"
#include "../../include/rt_mi355_probes.h"

void rt_lab_init(rt_ctx *c)
{
    memset(&c->lab, 0, sizeof c->lab);
    c->lab.r = 1;
    c->lab.block = RT_BLOCK;
    c->lab.gate_window = 1;
}

void rt_lab_destroy(rt_ctx *c)
{
    if (c->lab.d_probe_in)
        (void)hipFree(c->lab.d_probe_in);
    c->lab.d_probe_in = NULL;
}

hipError_t rt_lab_free(rt_ctx *c, void *p)
{
    rt_lab &l = c->lab;
    if (!p || p != l.vmm_base)
        return p ? hipFree(p) : hipSuccess;
    hipMemGenericAllocationHandle_t *h =
        (hipMemGenericAllocationHandle_t *)l.vmm_handles;
    (void)hipMemUnmap(l.vmm_base, l.vmm_bytes);
    for (size_t k = 0; k < l.vmm_n; ++k)
        (void)hipMemRelease(h[k]);
    (void)hipMemAddressFree(l.vmm_base, l.vmm_bytes);
    free(h);
    l.vmm_base = NULL;
    l.vmm_handles = NULL;
    l.vmm_n = 0;
    return hipSuccess;
}

/* chunk k of the address range <- physical chunk order[k]: the identity, or
 * a pseudo-random permutation drawn from vmm_seed */
static hipError_t rt_lab_vmm_map(rt_ctx *c)
{
    rt_lab &l = c->lab;
    const size_t n = l.vmm_n, chunk = l.vmm_chunk;
    hipMemGenericAllocationHandle_t *h =
        (hipMemGenericAllocationHandle_t *)l.vmm_handles;
    size_t *order = (size_t *)malloc(n * sizeof(size_t));
    for (size_t k = 0; k < n; ++k)
        order[k] = k;
    if (l.vmm_shuffle) {
        unsigned long long x = 0x9E3779B97F4A7C15ull * (l.vmm_seed + 1);
        for (size_t k = n - 1; k > 0; --k) {
            x ^= x << 13;
            x ^= x >> 7;
            x ^= x << 17;
            const size_t j = (size_t)(x % (k + 1));
            const size_t tmp = order[k];
            order[k] = order[j];
            order[j] = tmp;
        }
    }
    hipError_t e = hipSuccess;
    for (size_t k = 0; k < n && e == hipSuccess; ++k)
        e = hipMemMap((char *)l.vmm_base + k * chunk, chunk, 0, h[order[k]],
                      0);
    if (e == hipSuccess) {
        hipMemAccessDesc acc = {};
        acc.location.type = hipMemLocationTypeDevice;
        acc.location.id = c->device;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        e = hipMemSetAccess(l.vmm_base, l.vmm_bytes, &acc, 1);
    }
    free(order);
    return e;
}

hipError_t rt_lab_alloc(rt_ctx *c, void **out, size_t bytes)
{
    rt_lab &l = c->lab;
    if (l.vmm_mb <= 0)
        return hipMalloc(out, bytes);
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = c->device;
    size_t gran = 0;
    hipError_t e = hipMemGetAllocationGranularity(
        &gran, &prop, hipMemAllocationGranularityRecommended);
    if (e != hipSuccess)
        return e;
    size_t chunk = (size_t)l.vmm_mb << 20;
    chunk = (chunk + gran - 1) / gran * gran;
    const size_t n = (bytes + chunk - 1) / chunk;
    const size_t total = n * chunk;
    void *base = NULL;
    e = hipMemAddressReserve(&base, total,
                             (size_t)(l.vmm_align_mb > 0 ? l.vmm_align_mb : 2)
                                 << 20,
                             NULL, 0);
    if (e != hipSuccess)
        return e;
    hipMemGenericAllocationHandle_t *h =
        (hipMemGenericAllocationHandle_t *)calloc(
            n, sizeof(hipMemGenericAllocationHandle_t));
    for (size_t k = 0; k < n && e == hipSuccess; ++k)
        e = hipMemCreate(&h[k], chunk, &prop, 0);
    l.vmm_base = base;
    l.vmm_bytes = total;
    l.vmm_chunk = chunk;
    l.vmm_handles = h;
    l.vmm_n = n;
    if (e == hipSuccess)
        e = rt_lab_vmm_map(c);
    if (e != hipSuccess) {
        (void)rt_lab_free(c, base);
        return e;
    }
    *out = base;
    return hipSuccess;
}

bool rt_lab_variant(const rt_ctx *c)
{
    const rt_lab &l = c->lab;
    return l.r != 1 || l.nt || l.xcd || l.block != RT_BLOCK || l.lds ||
           (l.tile && !l.tile_shipped) || l.uniform_fix || l.gate_log2;
}

int rt_lab_set_option(rt_ctx *ctx, const char *key, int value)
{
    rt_lab &l = ctx->lab;
    if (!strcmp(key, "rays_per_thread")) {
        if (value != 1 && value != 2 && value != 4)
            return rt_fail(ctx, RT_ERR_ARG, "rays_per_thread must be 1, 2, 4");
        l.r = value;
    } else if (!strcmp(key, "nontemporal")) {
        l.nt = value ? 1 : 0;
    } else if (!strcmp(key, "xcd_remap")) {
        l.xcd = value ? 1 : 0;
    } else if (!strcmp(key, "gate_log2")) {
        if (value < 0 || value > 20)
            return rt_fail(ctx, RT_ERR_ARG, "gate_log2 must be in [0, 20]");
        l.gate_log2 = value;
    } else if (!strcmp(key, "gate_window")) {
        l.gate_window = value < 1 ? 1 : value;
    } else if (!strcmp(key, "uniform_fix")) {
        l.uniform_fix = value & 63;
    } else if (!strcmp(key, "probe_store")) {
        if (value < 0 || value > 3)
            return rt_fail(ctx, RT_ERR_ARG, "probe_store must be 0..3");
        l.probe_store = value;
    } else if (!strcmp(key, "tile_rays")) {
        /* tile-major layout (rt_lay.h); takes effect with the next
         * rt_reserve / rt_set_rays */
        if (value && (value < 64 || value > (1 << 24) ||
                      (value & (value - 1))))
            return rt_fail(ctx, RT_ERR_ARG,
                           "tile_rays must be 0 or a power of two in "
                           "[64, 2^24]");
        if (value != l.tile) {
            RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
            l.tile = value;
            ctx->ld = 0; /* the next rt_reserve lays the arrays out anew */
            ctx->n = 0;
            memset(ctx->valid, 0, sizeof ctx->valid);
        }
    } else if (!strcmp(key, "tile_planes")) {
        /* set before tile_rays (which lays the arrays out anew) */
        l.tile_planes = value ? 1 : 0;
    } else if (!strcmp(key, "t_before_i")) {
        if ((value != 0) != l.t_before_i) {
            RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
            l.t_before_i = value ? 1 : 0;
            ctx->ld = 0; /* the next rt_reserve lays the arrays out anew */
            ctx->n = 0;
            memset(ctx->valid, 0, sizeof ctx->valid);
        }
    } else if (!strcmp(key, "tile_shipped_kernel")) {
        /* a tile layout alone does not select the laboratory kernel: the
         * shipped rt_trace_kernel (which addresses through rt_col in this
         * build) runs in it */
        l.tile_shipped = value ? 1 : 0;
    } else if (!strcmp(key, "tile_pad")) {
        if (value < 0 || value > (1 << 22) || value % 64)
            return rt_fail(ctx, RT_ERR_ARG, "tile_pad: k*64 doubles");
        l.tile_pad = value;
    } else if (!strcmp(key, "alloc_vmm_mb")) {
        /* takes effect with the next allocation */
        if (value < 0 || value > 65536)
            return rt_fail(ctx, RT_ERR_ARG, "alloc_vmm_mb: 0..65536");
        l.vmm_mb = value;
    } else if (!strcmp(key, "alloc_vmm_align_mb")) {
        l.vmm_align_mb = value; /* alignment of the address range */
    } else if (!strcmp(key, "alloc_vmm_shuffle")) {
        l.vmm_shuffle = value ? 1 : 0;
    } else if (!strcmp(key, "alloc_vmm_seed")) {
        /* another permutation of the physical chunks behind the SAME
         * addresses, at once; what the rows held is gone */
        l.vmm_seed = (unsigned)value;
        if (l.vmm_base) {
            RT_HIP(ctx, hipDeviceSynchronize());
            RT_HIP(ctx, hipMemUnmap(l.vmm_base, l.vmm_bytes));
            RT_HIP(ctx, rt_lab_vmm_map(ctx));
            ctx->n = 0;
            ctx->ld = 0;
            memset(ctx->valid, 0, sizeof ctx->valid);
        }
    } else if (!strcmp(key, "alloc_round")) {
        /* size of the allocation behind the arrays: rounded up to a multiple
         * of 2^value bytes (4..40), or to a power of two (99); takes effect
         * with the next allocation */
        if (value && value != 99 && (value < 4 || value > 40))
            return rt_fail(ctx, RT_ERR_ARG, "alloc_round: 0, 4..40 or 99");
        l.alloc_round = value;
    } else if (!strcmp(key, "base_offset_kb")) {
        /* where in its allocation the result arrays start (placement
         * experiments); takes effect with the next rt_reserve / rt_set_rays */
        if (value < 0 || (size_t)value * 128 >= RT_LAB_SLACK)
            return rt_fail(ctx, RT_ERR_ARG, "base_offset_kb must be in "
                           "[0, %zu)", RT_LAB_SLACK / 128);
        RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        l.base_off = (size_t)value * 128;
        ctx->ld = 0;
        ctx->n = 0;
        memset(ctx->valid, 0, sizeof ctx->valid);
    } else if (!strcmp(key, "lds_pad")) {
        /* dynamic LDS the kernel never touches, to cap the workgroups
         * resident per CU (160 KB / lds_pad) */
        if (value < 0 || value > 65536)
            return rt_fail(ctx, RT_ERR_ARG, "lds_pad must be in [0, 65536]");
        l.lds = value;
    } else if (!strcmp(key, "block")) {
        if (value < 64 || value > 1024 || value % 64)
            return rt_fail(ctx, RT_ERR_ARG, "block must be k*64 in [64,1024]");
        l.block = value;
    } else {
        return 0; /* not a laboratory key */
    }
    return 1;
}

template <int R, bool NT, bool XCD>
static void rt_lab_launch_as(rt_ctx *c, int start, int stop, int clip)
{
    const int block = c->lab.block;
    const int64_t per_block = (int64_t)block * R;
    const int64_t nblocks = (c->ld + per_block - 1) / per_block;
    const int64_t grid = XCD ? (nblocks + 7) / 8 * 8 : nblocks;
    hipLaunchKernelGGL((rt_trace_lab_kernel<R, NT, XCD>), dim3((unsigned)grid),
                       dim3(block), (size_t)c->lab.lds, c->stream, c->d_surf,
                       start, stop, clip, rt_layout(c), c->ld, nblocks,
                       c->ngroups > 1 ? c->n / c->ngroups : (int64_t)0,
                       c->nsurf, (const unsigned *)NULL,
                       (unsigned)(start == 1 ? c->lab.uniform_fix : 0),
                       c->lab.gate_log2 ? (1u << c->lab.gate_log2) - 1u : 0u,
                       (unsigned)c->lab.gate_window);
}

int rt_lab_launch(rt_ctx *ctx, int start, int stop, int clip)
{
    const int key = ctx->lab.r * 4 + ctx->lab.nt * 2 + ctx->lab.xcd;
    switch (key) {
#define RT_CASE(R, NT, X)                                                     \
    case (R) * 4 + (NT) * 2 + (X):                                            \
        rt_lab_launch_as<R, NT, X>(ctx, start, stop, clip);                   \
        break;
        RT_CASE(1, 0, 0) RT_CASE(1, 0, 1) RT_CASE(1, 1, 0) RT_CASE(1, 1, 1)
        RT_CASE(2, 0, 0) RT_CASE(2, 0, 1) RT_CASE(2, 1, 0) RT_CASE(2, 1, 1)
        RT_CASE(4, 0, 0) RT_CASE(4, 0, 1) RT_CASE(4, 1, 0) RT_CASE(4, 1, 1)
#undef RT_CASE
    default:
        return rt_fail(ctx, RT_ERR_STATE, "rt_trace: bad variant %d", key);
    }
    RT_HIP(ctx, hipGetLastError());
    return RT_OK;
}

extern "C" {

int rt_probes_built(void) { return 1; }

int rt_probe(rt_ctx *ctx, int mode, double *ms, double *bytes)
{
    if (!ctx || !ms || !bytes)
        return rt_fail(ctx, RT_ERR_ARG, "rt_probe: NULL argument");
    if (!ctx->d_buf || ctx->nsurf < 2)
        return rt_fail(ctx, RT_ERR_STATE, "rt_probe: set rays first");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    const int L = ctx->buf_nsurf;
    const int64_t ld = ctx->ld;
    if (ctx->lab.tile && (mode >= 1 && mode <= 4))
        return rt_fail(ctx, RT_ERR_STATE,
                       "rt_probe: the linear fills address the SoA layout");
    /* rows 1..L-1 of the four arrays; row 0 (the input rays) is preserved */
    RT_HIP(ctx, hipEventRecord(ctx->k0, ctx->stream));
    if (mode == 0 || (mode >= 5 && mode <= 9) || mode == 13 || mode == 14) {
        /* the trace kernel's store pattern without its arithmetic, in the
         * layout in force (SoA or tile_rays):
         *   0  80 B/op, 16-byte stores, 48 B/ray input read from HBM
         *   5  same, input from an L2-resident window     6  no input read
         *   7  56 B/op (i served from u), 8-byte stores like the default
         *      kernel, input from HBM                     8  same, no read */
        const int block = ctx->lab.block;
        const int rp = mode >= 7 ? 1 : 2; /* 7..14: one ray per lane */
        const unsigned grid =
            (unsigned)((ld / rp + block - 1) / block);
        const rt_lay lay = rt_layout(ctx);
        const double *win = ctx->d_buf;
        if (mode == 13 || mode == 14) {
            /* 7 with the input rows in their own allocation: 13 = uncached
             * (MTYPE UC: reads bypass the L2), 14 = ordinary device memory */
            const size_t need = (size_t)6 * ld * sizeof(double);
            if (ctx->lab.probe_in_bytes != need || ctx->lab.probe_in_uc != (mode == 13)) {
                if (ctx->lab.d_probe_in)
                    (void)hipFree(ctx->lab.d_probe_in);
                ctx->lab.d_probe_in = NULL;
                ctx->lab.probe_in_bytes = 0;
                if (mode == 13)
                    RT_HIP(ctx, hipExtMallocWithFlags(&ctx->lab.d_probe_in, need,
                                                      hipDeviceMallocUncached));
                else
                    RT_HIP(ctx, hipMalloc(&ctx->lab.d_probe_in, need));
                RT_HIP(ctx, hipMemsetAsync(ctx->lab.d_probe_in, 0, need,
                                           ctx->stream));
                ctx->lab.probe_in_bytes = need;
                ctx->lab.probe_in_uc = mode == 13;
                RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
                RT_HIP(ctx, hipEventRecord(ctx->k0, ctx->stream));
            }
            win = (const double *)ctx->lab.d_probe_in;
        }
#define RT_PROBE(IN, RP, SI)                                                  \
    do {                                                                      \
        switch (ctx->lab.probe_store) {                                       \
        case 1: RT_PROBE_FL(IN, RP, SI, 1); break;                            \
        case 2: RT_PROBE_FL(IN, RP, SI, 2); break;                            \
        case 3: RT_PROBE_FL(IN, RP, SI, 3); break;                            \
        default: RT_PROBE_FL(IN, RP, SI, 0); break;                           \
        }                                                                     \
    } while (0)
#define RT_PROBE_FL(IN, RP, SI, FL)                                           \
    hipLaunchKernelGGL((rt_probe_pattern_kernel<IN, RP, FL>), dim3(grid),     \
                       dim3(block), (size_t)ctx->lab.lds, ctx->stream, 1, L, win, lay, \
                       ld, SI)
        switch (mode) {
        case 0: RT_PROBE(0, 2, 1); break;
        case 5: RT_PROBE(1, 2, 1); break;
        case 6: RT_PROBE(2, 2, 1); break;
        case 7: RT_PROBE(0, 1, 0); break;
        case 9: RT_PROBE(3, 1, 0); break; /* 7 with non-temporal loads */
        case 13:
        case 14: RT_PROBE(4, 1, 0); break;
        default: RT_PROBE(2, 1, 0); break;
        }
#undef RT_PROBE_FL
#undef RT_PROBE
        *bytes = (double)ld * ((mode >= 7 ? 56. : 80.) * (L - 1) +
                               ((mode == 0 || mode == 7 || mode == 9 ||
                                 mode >= 13) ? 48. : 0.));
    } else if (mode == 10 || mode == 11 || mode == 12) {
        /* 56 B pattern, K = 2 / 4 / 8 rays per lane one after the other,
         * inputs loaded up front */
        const int block = ctx->lab.block;
        const int K = mode == 10 ? 2 : (mode == 11 ? 4 : 8);
        const unsigned grid =
            (unsigned)((ld + (int64_t)block * K - 1) / ((int64_t)block * K));
        const rt_lay lay = rt_layout(ctx);
        if (K == 2)
            hipLaunchKernelGGL(rt_probe_seq_kernel<2>, dim3(grid), dim3(block),
                               0, ctx->stream, 1, L, lay, ld);
        else if (K == 4)
            hipLaunchKernelGGL(rt_probe_seq_kernel<4>, dim3(grid), dim3(block),
                               0, ctx->stream, 1, L, lay, ld);
        else
            hipLaunchKernelGGL(rt_probe_seq_kernel<8>, dim3(grid), dim3(block),
                               0, ctx->stream, 1, L, lay, ld);
        *bytes = (double)ld * (56. * (L - 1) + 48.);
    } else if (mode == 1) {
        double total = 0.;
        for (int w = RT_Y; w <= RT_T; ++w) {
            const int nc = rt_ncomp(w);
            const int64_t n2 = (int64_t)(L - 1) * nc * ld / 2;
            hipLaunchKernelGGL(rt_probe_fill_kernel, dim3(2048), dim3(256), 0,
                               ctx->stream, rt_arr(ctx, w) + (size_t)nc * ld,
                               n2);
            total += (double)n2 * 16.;
        }
        *bytes = total;
    } else if (mode == 3 || mode == 4) {
        double total = 0.;
        for (int w = RT_Y; w <= RT_T; ++w) {
            const int nc = rt_ncomp(w);
            const int64_t n2 = (int64_t)(L - 1) * nc * ld / 2;
            const unsigned grid = (unsigned)((n2 + 255) / 256);
            if (mode == 3)
                hipLaunchKernelGGL(rt_probe_fill_once_kernel<false>,
                                   dim3(grid), dim3(256), 0, ctx->stream,
                                   rt_arr(ctx, w) + (size_t)nc * ld, n2);
            else
                hipLaunchKernelGGL(rt_probe_fill_once_kernel<true>,
                                   dim3(grid), dim3(256), 0, ctx->stream,
                                   rt_arr(ctx, w) + (size_t)nc * ld, n2);
            total += (double)n2 * 16.;
        }
        *bytes = total;
    } else if (mode == 2) {
        /* copy rows 1..h of Y -> rows 1..h of U, h = L-1: read + write */
        const int64_t n2 = (int64_t)(L - 1) * 3 * ld / 2;
        hipLaunchKernelGGL(rt_probe_copy_kernel, dim3(2048), dim3(256), 0,
                           ctx->stream, rt_arr(ctx, RT_Y) + (size_t)3 * ld,
                           rt_arr(ctx, RT_I) + (size_t)3 * ld, n2);
        *bytes = (double)n2 * 32.;
    } else {
        return rt_fail(ctx, RT_ERR_ARG, "rt_probe: mode %d", mode);
    }
    RT_HIP(ctx, hipGetLastError());
    RT_HIP(ctx, hipEventRecord(ctx->k1, ctx->stream));
    RT_HIP(ctx, hipEventSynchronize(ctx->k1));
    float f = 0.f;
    RT_HIP(ctx, hipEventElapsedTime(&f, ctx->k0, ctx->k1));
    *ms = f;
    return RT_OK;
}

} /* extern "C" */
