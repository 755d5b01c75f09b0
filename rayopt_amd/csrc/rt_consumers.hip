/*
 * rt_consumers.hip -- what runs after (or around) a trace on device-resident
 * rows: the aiming kernel behind System.pupil,
 * GeometricTrace.rms / refocus / resize sums, per-bundle spot statistics and
 * the per-ray part of opd.  Deterministic two-level reductions, scalars (or
 * 24 B/ray for opd) cross PCIe instead of rows.  Part of librt_mi355.so.
 */
#include "rt_ctx.h"
#include "rt_consumer_kernels.h"

extern "C" {

int rt_sizeof_aim_seed(void) { return (int)sizeof(rt_aim_seed); }
int rt_sizeof_aim_args(void) { return (int)sizeof(rt_aim_args); }

int rt_aim_pupil(rt_ctx *ctx, const rt_aim_seed *seeds, int nfields,
                 const rt_aim_args *args, double *z, double *a,
                 int32_t *status)
{
    if (!ctx || !seeds || !args || !z || !a || !status || nfields < 1)
        return rt_fail(ctx, RT_ERR_ARG, "rt_aim_pupil: bad argument");
    if (ctx->nsurf < 3)
        return rt_fail(ctx, RT_ERR_STATE,
                       "rt_aim_pupil: rt_upload_system must come first");
    if (args->stop < 1 || args->stop > ctx->nsurf - 2 || args->maxiter < 1)
        return rt_fail(ctx, RT_ERR_ARG, "rt_aim_pupil: stop %d of %d elements",
                       args->stop, ctx->nsurf);
    for (int f = 0; f < nfields; ++f)
        if (seeds[f].group < 0 || seeds[f].group >= ctx->ngroups)
            return rt_fail(ctx, RT_ERR_ARG,
                           "rt_aim_pupil: field %d names table %d of %d", f,
                           seeds[f].group, ctx->ngroups);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    /* scratch: tables | seeds | z | a | status, each 256-byte aligned; the
     * same layout in one pinned host buffer, so that a call costs one copy
     * in, the kernel, one copy out */
    const size_t ntab = (size_t)ctx->nsurf * ctx->ngroups;
    const size_t tb = (sizeof(rt_surface) * ntab + 255) / 256 * 256;
    const size_t sb = (sizeof(rt_aim_seed) * nfields + 255) / 256 * 256;
    const size_t zb = (sizeof(double) * nfields + 255) / 256 * 256;
    const size_t ab = (sizeof(double) * 4 * nfields + 255) / 256 * 256;
    const size_t cb = (sizeof(int32_t) * nfields + 255) / 256 * 256;
    const size_t all = tb + sb + zb + ab + cb;
    int rc = rt_need_scratch(ctx, all);
    if (rc != RT_OK)
        return rc;
    if (all > ctx->h_aim_bytes) {
        if (ctx->h_aim)
            (void)hipHostFree(ctx->h_aim);
        ctx->h_aim = NULL;
        ctx->h_aim_bytes = 0;
        const size_t want = all + all / 2;
        RT_HIP(ctx, hipHostMalloc((void **)&ctx->h_aim, want));
        ctx->h_aim_bytes = want;
    }
    char *base = (char *)ctx->d_scratch, *host = ctx->h_aim;
    rt_surface *d_tab = (rt_surface *)base;
    rt_aim_seed *d_seeds = (rt_aim_seed *)(base + tb);
    double *d_z = (double *)(base + tb + sb);
    double *d_a = (double *)(base + tb + sb + zb);
    int32_t *d_status = (int32_t *)(base + tb + sb + zb + ab);
    memcpy(host, ctx->h_surf, sizeof(rt_surface) * ntab);
    memcpy(host + tb, seeds, sizeof(rt_aim_seed) * nfields);
    RT_HIP(ctx, hipMemcpyAsync(base, host, tb + sb, hipMemcpyHostToDevice,
                               ctx->stream));
    /* one wavefront per field while they are all resident at once, 16
     * fields per wavefront beyond that (rt_kernels.h) */
    if (nfields <= 32768)
        hipLaunchKernelGGL(rt_aim_kernel<true>, dim3((unsigned)nfields),
                           dim3(4), 0, ctx->stream, d_tab, ctx->nsurf, d_seeds,
                           nfields, *args, d_z, d_a, d_status);
    else
        hipLaunchKernelGGL(rt_aim_kernel<false>,
                           dim3((unsigned)(((int64_t)nfields * 4 + 63) / 64)),
                           dim3(64), 0, ctx->stream, d_tab, ctx->nsurf,
                           d_seeds, nfields, *args, d_z, d_a, d_status);
    RT_HIP(ctx, hipGetLastError());
    RT_HIP(ctx, hipMemcpyAsync(host + tb + sb, base + tb + sb, zb + ab + cb,
                               hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(z, host + tb + sb, sizeof(double) * nfields);
    memcpy(a, host + tb + sb + zb, sizeof(double) * 4 * nfields);
    memcpy(status, host + tb + sb + zb + ab, sizeof(int32_t) * nfields);
    return RT_OK;
}

int rt_set_weights(rt_ctx *ctx, const double *w)
{
    if (!ctx)
        return rt_fail(ctx, RT_ERR_ARG, "rt_set_weights: NULL context");
    if (ctx->n < 1)
        return rt_fail(ctx, RT_ERR_STATE, "rt_set_weights: set rays first");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (!w) {
        if (ctx->d_w)
            RT_HIP(ctx, hipFree(ctx->d_w));
        ctx->d_w = NULL;
        ctx->w_cap = 0;
        ctx->w_n = 0;
        return RT_OK;
    }
    if ((size_t)ctx->n > ctx->w_cap) {
        if (ctx->d_w)
            RT_HIP(ctx, hipFree(ctx->d_w));
        ctx->d_w = NULL; /* uniform weights if the allocation fails */
        ctx->w_cap = 0;
        ctx->w_n = 0;
        RT_HIP(ctx, hipMalloc((void **)&ctx->d_w, ctx->n * sizeof(double)));
        ctx->w_cap = (size_t)ctx->n;
    }
    RT_HIP(ctx, hipMemcpyAsync(ctx->d_w, w, ctx->n * sizeof(double),
                               hipMemcpyHostToDevice, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->w_n = ctx->n;
    return RT_OK;
}

/* fetch and add the per-workgroup partials in index order */
static int rt_consumer_ready(rt_ctx *ctx, int surf, const char *who)
{
    if (!ctx)
        return rt_fail(ctx, RT_ERR_ARG, "%s: NULL context", who);
    if (!ctx->d_buf || ctx->n < 1 || surf < 0 || surf >= ctx->buf_nsurf ||
        !ctx->valid[surf])
        return rt_fail(ctx, RT_ERR_STATE, "%s: row %d holds no data", who,
                       surf);
    RT_ROWS_WHOLE(ctx, who);
    if (ctx->d_w && ctx->w_n != ctx->n)
        return rt_fail(ctx, RT_ERR_STATE,
                       "%s: the weights on the device were set for a batch "
                       "of %lld rays, this one has %lld: call rt_set_weights "
                       "after seeding (NULL for uniform weights)", who,
                       (long long)ctx->w_n, (long long)ctx->n);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->d_partials)
        RT_HIP(ctx, hipMalloc((void **)&ctx->d_partials,
                              sizeof(double) * (RT_RED_BLOCKS * 16 + 16)));
    if (!ctx->h_res) { /* 8 results | the finishing kernel's ticket */
        RT_HIP(ctx, hipHostMalloc((void **)&ctx->h_res, 16 * sizeof(double)));
        memset(ctx->h_res, 0, 16 * sizeof(double));
    }
    return rt_gen_flush(ctx);
}

/* measurement only (option "consumer_events"): the kernels of a consumer
 * between the events rt_kernel_ms reads */
#define RT_CONSUMER_BEGIN(ctx)                                              \
    do {                                                                    \
        if ((ctx)->opt_cevents)                                             \
            RT_HIP(ctx, hipEventRecord((ctx)->k0, (ctx)->stream));          \
    } while (0)
#define RT_CONSUMER_END(ctx)                                                \
    do {                                                                    \
        if ((ctx)->opt_cevents) {                                           \
            RT_HIP(ctx, hipEventRecord((ctx)->k1, (ctx)->stream));          \
            (ctx)->traced = 1;                                              \
        }                                                                   \
    } while (0)

/* where a finishing kernel signs (rt_sign), and the host's wait for it: a
 * spin on pinned memory instead of a stream synchronisation (with
 * "consumer_events": the stream, so that the events are complete); a device
 * that does not sign within 2 s is left to the stream's error reporting */
static inline unsigned long long *rt_res_ticket(rt_ctx *ctx)
{
    return (unsigned long long *)(ctx->h_res + 15);
}

static int rt_wait_signed(rt_ctx *ctx, unsigned long long *ticket,
                          unsigned long long seq, const char *who)
{
    if (ctx->opt_cevents) {
        RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return RT_OK;
    }
    volatile unsigned long long *t = ticket;
    const double t0 = rt_now_ms();
    unsigned spins = 0;
    while (*t != seq) {
        __builtin_ia32_pause();
        if (!(++spins & 0xfff) && rt_now_ms() - t0 > 2000.) {
            RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
            break;
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    if (*t != seq)
        return rt_fail(ctx, RT_ERR_HIP, "%s: the finishing kernel never "
                                        "signed", who);
    return RT_OK;
}

/* how the rows lie in memory, as the reduction kernels take it */
static inline rt_pitch rt_pitch_of(const rt_ctx *ctx)
{
    rt_pitch p = {ctx->bs, ctx->bts};
    return p;
}

/* workgroups of a reduction over n rays: no more than there is work for */
static inline unsigned rt_red_blocks(int64_t n)
{
    const int64_t b = (n + RT_RED_THREADS - 1) / RT_RED_THREADS;
    return (unsigned)(b < 1 ? 1 : (b > RT_RED_BLOCKS ? RT_RED_BLOCKS : b));
}

/* device-side second level of a reduction: k sums -> ctx->d_partials tail */
static inline double *rt_reduced(rt_ctx *ctx, int slot)
{
    return ctx->d_partials + (size_t)RT_RED_BLOCKS * 16 + slot;
}

int rt_rms(rt_ctx *ctx, int surf, int64_t ref, double *rms)
{
    int rc = rt_consumer_ready(ctx, surf, "rt_rms");
    if (rc != RT_OK)
        return rc;
    if (!rms || ref >= ctx->n)
        return rt_fail(ctx, RT_ERR_ARG, "rt_rms: bad argument");
    const double *Yrow = rt_row(ctx, RT_Y, surf);
    const unsigned blocks = rt_red_blocks(ctx->n);
    if (ctx->opt_onepass) {
        /* one pass over the row, shifted by a ray of the bundle; its second
         * level leaves the scalars in pinned memory.  (Finishing in the last
         * workgroup to arrive -- one launch, ticket + device-scope fences --
         * was built and measured: the 1024 L2 write-backs / invalidates of
         * the fences cost 17 us, the launch they save 5:
         * profiles/r04_probes/session30.) */
        const unsigned long long seq = ++ctx->row_seq;
        RT_CONSUMER_BEGIN(ctx);
        if (ctx->d_w) {
            hipLaunchKernelGGL(rt_rms_shifted_kernel<true>, dim3(blocks),
                               dim3(RT_RED_THREADS), 0, ctx->stream, Yrow,
                               ctx->d_w, ref, ctx->n, rt_pitch_of(ctx),
                               ctx->d_partials);
            hipLaunchKernelGGL(rt_rms_finish_kernel<true>, dim3(1), dim3(64),
                               0, ctx->stream, ctx->d_partials, (int)blocks,
                               ref < 0 ? 1 : 0, (double)ctx->n, ctx->h_res,
                               rt_res_ticket(ctx), seq);
        } else {
            hipLaunchKernelGGL(rt_rms_shifted_kernel<false>, dim3(blocks),
                               dim3(RT_RED_THREADS), 0, ctx->stream, Yrow,
                               ctx->d_w, ref, ctx->n, rt_pitch_of(ctx),
                               ctx->d_partials);
            hipLaunchKernelGGL(rt_rms_finish_kernel<false>, dim3(1), dim3(64),
                               0, ctx->stream, ctx->d_partials, (int)blocks,
                               ref < 0 ? 1 : 0, (double)ctx->n, ctx->h_res,
                               rt_res_ticket(ctx), seq);
        }
        RT_CONSUMER_END(ctx);
        RT_HIP(ctx, hipGetLastError());
        rc = rt_wait_signed(ctx, rt_res_ticket(ctx), seq, "rt_rms");
        if (rc != RT_OK)
            return rc;
        const double r = ctx->h_res[0], a = ctx->h_res[1];
        /* NaN stays NaN (one vignetted ray poisons the reference's mean
         * too); otherwise the subtraction may have cost six bits */
        if (ref >= 0 || r != r || a != a || (r >= 0. && r * 64. >= a)) {
            *rms = sqrt(r);
            return RT_OK;
        }
    }
    /* two passes (mean, then spread about it) and their second levels,
     * queued back to back; the host waits once */
    RT_CONSUMER_BEGIN(ctx);
    if (ref < 0) {
        hipLaunchKernelGGL(rt_sum_xy_kernel, dim3(blocks),
                           dim3(RT_RED_THREADS), 0, ctx->stream, Yrow, ctx->n,
                           rt_pitch_of(ctx), ctx->d_partials);
        hipLaunchKernelGGL(rt_finalize_kernel, dim3(1), dim3(64), 0,
                           ctx->stream, ctx->d_partials, (int)blocks, 2,
                           rt_reduced(ctx, 0));
    }
    hipLaunchKernelGGL(rt_rms_kernel, dim3(blocks), dim3(RT_RED_THREADS), 0,
                       ctx->stream, Yrow, ctx->d_w, 1. / (double)ctx->n,
                       rt_reduced(ctx, 0), ref, ctx->n, rt_pitch_of(ctx),
                       ctx->d_partials);
    hipLaunchKernelGGL(rt_finalize_kernel, dim3(1), dim3(64), 0, ctx->stream,
                       ctx->d_partials, (int)blocks, 1, ctx->h_res);
    RT_CONSUMER_END(ctx);
    RT_HIP(ctx, hipGetLastError());
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *rms = sqrt(ctx->h_res[0]);
    return RT_OK;
}

int rt_row_rmax(rt_ctx *ctx, int surf, double *rmax)
{
    int rc = rt_consumer_ready(ctx, surf, "rt_row_rmax");
    if (rc != RT_OK)
        return rc;
    if (!rmax)
        return rt_fail(ctx, RT_ERR_ARG, "rt_row_rmax: NULL");
    const unsigned blocks = rt_red_blocks(ctx->n);
    const unsigned long long seq = ++ctx->row_seq;
    RT_CONSUMER_BEGIN(ctx);
    hipLaunchKernelGGL(rt_r2max_kernel, dim3(blocks), dim3(RT_RED_THREADS), 0,
                       ctx->stream, rt_row(ctx, RT_Y, surf), ctx->n, rt_pitch_of(ctx),
                       ctx->d_partials);
    hipLaunchKernelGGL(rt_r2max_finish_kernel, dim3(1), dim3(64), 0,
                       ctx->stream, ctx->d_partials, (int)blocks, ctx->h_res,
                       rt_res_ticket(ctx), seq);
    RT_CONSUMER_END(ctx);
    RT_HIP(ctx, hipGetLastError());
    rc = rt_wait_signed(ctx, rt_res_ticket(ctx), seq, "rt_row_rmax");
    if (rc != RT_OK)
        return rc;
    *rmax = ctx->h_res[1] != 0. ? __builtin_nan("") : sqrt(ctx->h_res[0]);
    return RT_OK;
}

int rt_spot_stats(rt_ctx *ctx, int surf, int64_t group_rays, int ngroups,
                  double *out)
{
    int rc = rt_consumer_ready(ctx, surf, "rt_spot_stats");
    if (rc != RT_OK)
        return rc;
    if (!out || group_rays < 1 || ngroups < 1 || ngroups > 65535 ||
        group_rays * (int64_t)ngroups != ctx->n)
        return rt_fail(ctx, RT_ERR_ARG,
                       "rt_spot_stats: %d groups of %lld rays do not tile the "
                       "%lld rays of the batch", ngroups, (long long)group_rays,
                       (long long)ctx->n);
    /* enough workgroups per group to fill the chip, no more than it has rays
     * for */
    int64_t pb = 2048 / ngroups;
    const int64_t fit = (group_rays + RT_RED_THREADS - 1) / RT_RED_THREADS;
    pb = pb > fit ? fit : pb;
    pb = pb < 1 ? 1 : (pb > 256 ? 256 : pb);
    const size_t need = (size_t)ngroups * (RT_GRP_STATS + (size_t)pb * 4);
    if (need > ctx->group_cap) {
        if (ctx->d_group)
            (void)hipFree(ctx->d_group);
        ctx->d_group = nullptr;
        ctx->group_cap = 0;
        RT_HIP(ctx, hipMalloc((void **)&ctx->d_group, need * sizeof(double)));
        ctx->group_cap = need;
    }
    /* a few groups: the last kernel writes the finished stats into pinned
     * memory and no copy follows it (a pageable 240-byte D2H costs as much as
     * a pass over the row); the centroids stay on the device for pass B */
    if (ngroups <= RT_GROUP_PINNED && !ctx->h_group)
        RT_HIP(ctx, hipHostMalloc((void **)&ctx->h_group,
                                  sizeof(double) * RT_GRP_STATS *
                                      RT_GROUP_PINNED));
    double *stats = ctx->d_group;
    double *final = ngroups <= RT_GROUP_PINNED ? ctx->h_group : stats;
    double *partials = stats + (size_t)ngroups * RT_GRP_STATS;
    const double *Yrow = rt_row(ctx, RT_Y, surf);
    const dim3 grid((unsigned)pb, (unsigned)ngroups), block(RT_RED_THREADS);
    const dim3 fgrid((unsigned)ngroups), fblock(64); /* a wavefront each */
    RT_CONSUMER_BEGIN(ctx);
    hipLaunchKernelGGL(rt_group_sums_kernel, grid, block, 0, ctx->stream, Yrow,
                       ctx->d_w, group_rays, rt_pitch_of(ctx), partials);
    hipLaunchKernelGGL(rt_group_centroid_kernel, fgrid, fblock, 0, ctx->stream,
                       partials, (int)pb, ngroups, stats);
    hipLaunchKernelGGL(rt_group_spread_kernel, grid, block, 0, ctx->stream,
                       Yrow, ctx->d_w, group_rays, rt_pitch_of(ctx), stats,
                       partials);
    hipLaunchKernelGGL(rt_group_finish_kernel, fgrid, fblock, 0, ctx->stream,
                       partials, (int)pb, ngroups, stats, final);
    RT_CONSUMER_END(ctx);
    RT_HIP(ctx, hipGetLastError());
    if (final == stats)
        RT_HIP(ctx, hipMemcpyAsync(out, stats,
                                   sizeof(double) * RT_GRP_STATS * ngroups,
                                   hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (final != stats)
        memcpy(out, final, sizeof(double) * RT_GRP_STATS * ngroups);
    return RT_OK;
}

/*
 * The statistics of the image row in one pass (see rt_row_stats_kernel).
 * Two launches -- the pass, one wavefront per bundle to finish -- and no
 * stream synchronisation for up to RT_GROUP_PINNED bundles: the finishing
 * kernel writes the results into pinned memory and signs a ticket there, the
 * host spins on the ticket (a hipStreamSynchronize costs more than the
 * finishing kernel takes).
 */
int rt_row_stats(rt_ctx *ctx, int surf, int64_t group_rays, int ngroups,
                 int64_t ref, double *out)
{
    int rc = rt_consumer_ready(ctx, surf, "rt_row_stats");
    if (rc != RT_OK)
        return rc;
    if (!out || group_rays < 1 || ngroups < 1 || ngroups > 65535 ||
        group_rays * (int64_t)ngroups != ctx->n || ref >= group_rays)
        return rt_fail(ctx, RT_ERR_ARG,
                       "rt_row_stats: %d groups of %lld rays do not tile the "
                       "%lld rays of the batch, or the reference ray %lld is "
                       "not one of a group's", ngroups, (long long)group_rays,
                       (long long)ctx->n, (long long)ref);
    int64_t pb = 2048 / ngroups;
    const int64_t fit = (group_rays + RT_RED_THREADS - 1) / RT_RED_THREADS;
    pb = pb > fit ? fit : pb;
    pb = pb < 1 ? 1 : (pb > 256 ? 256 : pb);
    /* final (where it stays on the device) | shifts | partials */
    const size_t need =
        (size_t)ngroups * (RT_ROW_STATS + 4 + (size_t)pb * RT_ROW_ACC);
    if (need > ctx->group_cap) {
        RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->d_group)
            (void)hipFree(ctx->d_group);
        ctx->d_group = nullptr;
        ctx->group_cap = 0;
        RT_HIP(ctx, hipMalloc((void **)&ctx->d_group, need * sizeof(double)));
        ctx->group_cap = need;
    }
    const bool pinned = ngroups <= RT_GROUP_PINNED;
    if (pinned && !ctx->h_rows) {
        RT_HIP(ctx, hipHostMalloc((void **)&ctx->h_rows,
                                  sizeof(double) *
                                      (RT_ROW_STATS * RT_GROUP_PINNED + 2)));
        memset(ctx->h_rows, 0,
               sizeof(double) * (RT_ROW_STATS * RT_GROUP_PINNED + 2));
    }
    if (pinned && !ctx->d_arrived) {
        RT_HIP(ctx, hipMalloc((void **)&ctx->d_arrived, 64));
        RT_HIP(ctx, hipMemsetAsync(ctx->d_arrived, 0, 64, ctx->stream));
    }
    double *dfinal = ctx->d_group;
    double *shifts = dfinal + (size_t)ngroups * RT_ROW_STATS;
    double *partials = shifts + (size_t)ngroups * 4;
    double *final = pinned ? ctx->h_rows : dfinal;
    unsigned long long *ticket =
        pinned ? (unsigned long long *)(ctx->h_rows +
                                        RT_ROW_STATS * RT_GROUP_PINNED)
               : NULL;
    const unsigned long long seq = ++ctx->row_seq;
    const double *Yrow = rt_row(ctx, RT_Y, surf);
    const dim3 grid((unsigned)pb, (unsigned)ngroups), block(RT_RED_THREADS);
    RT_CONSUMER_BEGIN(ctx);
    if (ctx->d_w)
        hipLaunchKernelGGL(rt_row_stats_kernel<true>, grid, block, 0,
                           ctx->stream, Yrow, ctx->d_w, group_rays, ref,
                           rt_pitch_of(ctx), partials, shifts);
    else
        hipLaunchKernelGGL(rt_row_stats_kernel<false>, grid, block, 0,
                           ctx->stream, Yrow, ctx->d_w, group_rays, ref,
                           rt_pitch_of(ctx), partials, shifts);
    hipLaunchKernelGGL(rt_row_stats_finish_kernel, dim3((unsigned)ngroups),
                       dim3(64), 0, ctx->stream, partials, (int)pb, ngroups,
                       shifts, final, ticket, seq, ctx->d_arrived);
    RT_CONSUMER_END(ctx);
    RT_HIP(ctx, hipGetLastError());
    if (pinned) {
        rc = rt_wait_signed(ctx, ticket, seq, "rt_row_stats");
        if (rc != RT_OK)
            return rc;
    } else {
        if (!pinned)
            RT_HIP(ctx, hipMemcpyAsync(out, dfinal,
                                       sizeof(double) * RT_ROW_STATS * ngroups,
                                       hipMemcpyDeviceToHost, ctx->stream));
        RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (pinned)
        memcpy(out, final, sizeof(double) * RT_ROW_STATS * ngroups);
    /* a bundle whose shift ray lay far outside it lost bits in the
     * subtraction (the spread about the mean below 1/64 of the spread about
     * the shift): those -- rare: the shift is a ray of the bundle -- get the
     * textbook's two passes */
    bool redo = false;
    for (int g = 0; g < ngroups && !redo; ++g) {
        const double *f = out + (size_t)g * RT_ROW_STATS;
        redo = f[0] > 0. && f[4] == f[4] && !(f[4] >= 0. && f[4] * 64. >= f[9]);
    }
    if (redo) {
        double *two = (double *)malloc(sizeof(double) * RT_GRP_STATS * ngroups);
        if (!two)
            return rt_fail(ctx, RT_ERR_NOMEM, "rt_row_stats: host allocation");
        rc = rt_spot_stats(ctx, surf, group_rays, ngroups, two);
        for (int g = 0; g < ngroups && rc == RT_OK; ++g) {
            double *f = out + (size_t)g * RT_ROW_STATS;
            const double *s = two + (size_t)g * RT_GRP_STATS;
            /* (the spread about the reference ray stays: where that ray is
             * finite it WAS the shift and nothing was subtracted) */
            f[2] = s[1];
            f[3] = s[2];
            f[4] = s[3];
        }
        free(two);
        if (rc != RT_OK)
            return rc;
    }
    return RT_OK;
}

int rt_refocus_shift(rt_ctx *ctx, int surf, double *shift)
{
    int rc = rt_consumer_ready(ctx, surf, "rt_refocus_shift");
    if (rc != RT_OK)
        return rc;
    if (!shift)
        return rt_fail(ctx, RT_ERR_ARG, "rt_refocus_shift: NULL");
    const double *Yrow = rt_row(ctx, RT_Y, surf);
    const double *Irow = rt_row(ctx, RT_I, surf);
    const unsigned blocks = rt_red_blocks(ctx->n);
    if (ctx->opt_onepass) {
        const unsigned long long seq = ++ctx->row_seq;
        RT_CONSUMER_BEGIN(ctx);
        if (ctx->d_w) {
            hipLaunchKernelGGL(rt_refocus_shifted_kernel<true>, dim3(blocks),
                               dim3(RT_RED_THREADS), 0, ctx->stream, Yrow,
                               Irow, ctx->d_w, ctx->n, rt_pitch_of(ctx),
                               ctx->d_partials);
            hipLaunchKernelGGL(rt_refocus_finish_kernel<true>, dim3(1),
                               dim3(64), 0, ctx->stream, ctx->d_partials,
                               (int)blocks, ctx->h_res, rt_res_ticket(ctx),
                               seq);
        } else {
            hipLaunchKernelGGL(rt_refocus_shifted_kernel<false>, dim3(blocks),
                               dim3(RT_RED_THREADS), 0, ctx->stream, Yrow,
                               Irow, ctx->d_w, ctx->n, rt_pitch_of(ctx),
                               ctx->d_partials);
            hipLaunchKernelGGL(rt_refocus_finish_kernel<false>, dim3(1),
                               dim3(64), 0, ctx->stream, ctx->d_partials,
                               (int)blocks, ctx->h_res, rt_res_ticket(ctx),
                               seq);
        }
        RT_CONSUMER_END(ctx);
        RT_HIP(ctx, hipGetLastError());
        rc = rt_wait_signed(ctx, rt_res_ticket(ctx), seq, "rt_refocus_shift");
        if (rc != RT_OK)
            return rc;
        const double *h = ctx->h_res;
        /* ray 0 vignetted (nothing finite), or the shift by it cost more
         * than six bits of <u,u> or <y,y>: the two passes below */
        if (h[0] == h[0] && h[1] > 0. && h[1] * 64. >= h[3] && h[2] >= 0. &&
            h[2] * 64. >= h[4] && h[3] - h[3] == 0. && h[4] - h[4] == 0.) {
            *shift = -h[0] / h[1];
            return RT_OK;
        }
    }
    RT_CONSUMER_BEGIN(ctx);
    hipLaunchKernelGGL(rt_refocus_sums_kernel, dim3(blocks),
                       dim3(RT_RED_THREADS), 0, ctx->stream, Yrow, Irow, ctx->n,
                       rt_pitch_of(ctx), ctx->d_partials);
    hipLaunchKernelGGL(rt_finalize_kernel, dim3(1), dim3(64), 0, ctx->stream,
                       ctx->d_partials, (int)blocks, 5, rt_reduced(ctx, 0));
    hipLaunchKernelGGL(rt_refocus_dots_kernel, dim3(blocks),
                       dim3(RT_RED_THREADS), 0, ctx->stream, Yrow, Irow,
                       ctx->d_w, 1. / (double)ctx->n, rt_reduced(ctx, 0),
                       ctx->n, rt_pitch_of(ctx), ctx->d_partials);
    hipLaunchKernelGGL(rt_finalize_kernel, dim3(1), dim3(64), 0, ctx->stream,
                       ctx->d_partials, (int)blocks, 2, ctx->h_res);
    RT_CONSUMER_END(ctx);
    RT_HIP(ctx, hipGetLastError());
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *shift = -ctx->h_res[0] / ctx->h_res[1];
    return RT_OK;
}

/* what rt_opd_rays and rt_opd_stats share: the arguments checked, the rows
 * flushed, the reference-ray columns of every bundle gathered on the device */
static int rt_opd_prepare(rt_ctx *ctx, const rt_opd_args *args,
                          int64_t group_rays, int ngroups, const char *who)
{
    int rc = rt_consumer_ready(ctx, 0, who);
    if (rc != RT_OK)
        return rc;
    const int L = ctx->buf_nsurf;
    if (group_rays < 1 || ngroups < 1 || ngroups > RT_MAX_GROUPS ||
        group_rays * (int64_t)ngroups != ctx->n)
        return rt_fail(ctx, RT_ERR_ARG,
                       "%s: %d bundles of %lld rays do not tile the %lld rays "
                       "of the batch", who, ngroups, (long long)group_rays,
                       (long long)ctx->n);
    if (args->nrows < 0 || args->nrows > L || args->after < 0 ||
        args->after >= L || args->image < 0 || args->image >= L ||
        args->ref < 0 || args->ref >= group_rays)
        return rt_fail(ctx, RT_ERR_ARG, "%s: index out of range", who);
    for (int j = 0; j < L; ++j)
        if (!ctx->valid[j] &&
            (j < args->nrows || j == args->after || j == args->image))
            return rt_fail(ctx, RT_ERR_STATE, "%s: row %d holds no data", who,
                           j);
    /* Y and U are read at their natural addresses */
    rc = rt_detach(ctx, RT_U, args->after);
    if (rc != RT_OK)
        return rc;
    if ((size_t)ngroups > ctx->opd_ref_cap) {
        if (ctx->d_opd_ref)
            (void)hipFree(ctx->d_opd_ref);
        ctx->d_opd_ref = NULL;
        ctx->opd_ref_cap = 0;
        RT_HIP(ctx, hipMalloc((void **)&ctx->d_opd_ref,
                              sizeof(rt_opd_ref) * (size_t)ngroups));
        ctx->opd_ref_cap = (size_t)ngroups;
    }
    hipLaunchKernelGGL(rt_opd_refs_kernel, dim3((unsigned)((ngroups + 63) / 64)),
                       dim3(64), 0, ctx->stream, *args, rt_arr(ctx, RT_Y),
                       rt_arr(ctx, RT_U), rt_arr(ctx, RT_T), group_rays,
                       ngroups, rt_pitch_of(ctx), ctx->d_opd_ref);
    RT_HIP(ctx, hipGetLastError());
    return RT_OK;
}

/* the x | y | t array of the batch (24 B per ray), kept between calls */
static int rt_opd_buffer(rt_ctx *ctx)
{
    const size_t need = (size_t)ctx->n * 3;
    ctx->opd_n = 0;
    if (need > ctx->opd_cap) {
        RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->d_opd)
            (void)hipFree(ctx->d_opd);
        ctx->d_opd = NULL;
        ctx->opd_cap = 0;
        hipError_t e = hipMalloc((void **)&ctx->d_opd, need * sizeof(double));
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return rt_fail(ctx, RT_ERR_NOMEM, "opd: hipMalloc(%zu): %s",
                           need * sizeof(double), hipGetErrorString(e));
        }
        ctx->opd_cap = need;
    }
    return RT_OK;
}

int rt_opd_rays(rt_ctx *ctx, const rt_opd_args *args, double *out_soa)
{
    if (!ctx || !args || !out_soa)
        return rt_fail(ctx, RT_ERR_ARG, "rt_opd_rays: NULL argument");
    int rc = rt_opd_prepare(ctx, args, ctx->n, 1, "rt_opd_rays");
    if (rc == RT_OK)
        rc = rt_opd_buffer(ctx);
    if (rc != RT_OK)
        return rc;
    const unsigned grid = (unsigned)((ctx->n + 255) / 256);
    RT_HIP(ctx, hipEventRecord(ctx->k0, ctx->stream));
    hipLaunchKernelGGL(rt_opd_kernel, dim3(grid), dim3(256), 0, ctx->stream,
                       *args, ctx->d_opd_ref, rt_arr(ctx, RT_Y),
                       rt_arr(ctx, RT_U), rt_arr(ctx, RT_T), ctx->n,
                       rt_pitch_of(ctx), ctx->d_opd);
    RT_HIP(ctx, hipGetLastError());
    RT_HIP(ctx, hipEventRecord(ctx->k1, ctx->stream));
    ctx->traced = 1;
    ctx->opd_n = ctx->n;
    return rt_d2h(ctx, out_soa, ctx->d_opd,
                  (size_t)ctx->n * 3 * sizeof(double));
}

int rt_opd_stats(rt_ctx *ctx, const rt_opd_args *args, int64_t group_rays,
                 int ngroups, int keep, double *out)
{
    if (!ctx || !args || !out)
        return rt_fail(ctx, RT_ERR_ARG, "rt_opd_stats: NULL argument");
    int rc = rt_opd_prepare(ctx, args, group_rays, ngroups, "rt_opd_stats");
    if (rc == RT_OK && keep)
        rc = rt_opd_buffer(ctx);
    if (rc != RT_OK)
        return rc;
    /* enough workgroups per bundle to fill the chip (eight per CU), no more
     * than it has rays for */
    int64_t pb = 2048 / ngroups;
    const int64_t fit = (group_rays + RT_RED_THREADS - 1) / RT_RED_THREADS;
    pb = pb > fit ? fit : pb;
    pb = pb < 1 ? 1 : pb;
    const size_t need = (size_t)ngroups * (RT_OPD_STATS + (size_t)pb * RT_OPD_SUMS);
    if (need > ctx->group_cap) {
        RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->d_group)
            (void)hipFree(ctx->d_group);
        ctx->d_group = nullptr;
        ctx->group_cap = 0;
        RT_HIP(ctx, hipMalloc((void **)&ctx->d_group, need * sizeof(double)));
        ctx->group_cap = need;
    }
    if (ngroups <= RT_GROUP_PINNED && !ctx->h_opd)
        RT_HIP(ctx, hipHostMalloc((void **)&ctx->h_opd,
                                  sizeof(double) * RT_OPD_STATS *
                                      RT_GROUP_PINNED));
    double *stats = ctx->d_group;
    double *final = ngroups <= RT_GROUP_PINNED ? ctx->h_opd : stats;
    double *partials = stats + (size_t)ngroups * RT_OPD_STATS;
    RT_HIP(ctx, hipEventRecord(ctx->k0, ctx->stream));
    hipLaunchKernelGGL(rt_opd_stats_kernel, dim3((unsigned)pb, (unsigned)ngroups),
                       dim3(RT_RED_THREADS), 0, ctx->stream, *args,
                       ctx->d_opd_ref, rt_arr(ctx, RT_Y), rt_arr(ctx, RT_U),
                       rt_arr(ctx, RT_T), ctx->d_w, group_rays, ctx->n,
                       rt_pitch_of(ctx), keep ? ctx->d_opd : (double *)NULL,
                       partials);
    hipLaunchKernelGGL(rt_opd_finish_kernel, dim3((unsigned)ngroups), dim3(64),
                       0, ctx->stream, partials, (int)pb, final);
    RT_HIP(ctx, hipGetLastError());
    RT_HIP(ctx, hipEventRecord(ctx->k1, ctx->stream));
    ctx->traced = 1;
    if (final == stats)
        RT_HIP(ctx, hipMemcpyAsync(out, stats,
                                   sizeof(double) * RT_OPD_STATS * ngroups,
                                   hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (final != stats)
        memcpy(out, final, sizeof(double) * RT_OPD_STATS * ngroups);
    if (keep)
        ctx->opd_n = ctx->n;
    return RT_OK;
}

int rt_opd_device(rt_ctx *ctx, double **x_y_t, int64_t *nrays)
{
    if (!ctx || !x_y_t)
        return rt_fail(ctx, RT_ERR_ARG, "rt_opd_device: NULL argument");
    if (!ctx->d_opd || ctx->opd_n < 1 || ctx->opd_n != ctx->n)
        return rt_fail(ctx, RT_ERR_STATE,
                       "rt_opd_device: no path differences of this batch on "
                       "the device (rt_opd_stats with keep = 1, or "
                       "rt_opd_rays, first)");
    *x_y_t = ctx->d_opd;
    if (nrays)
        *nrays = ctx->opd_n;
    return RT_OK;
}

} /* extern "C" */
