/*
 * rt_mi355.h -- C ABI of librt_mi355.so, the MI355X (gfx950) engine behind
 * rayopt's GeometricTrace.propagate() hot path.
 *
 * The reference (quartiq/rayopt) has no FFI seam on this path: it is plain
 * Python method dispatch.  The seam adopted here is the narrowest one that
 * keeps every consumer of GeometricTrace working; each
 * entry point below names the reference interface it replaces.  All functions
 * are extern "C", take plain pointers and sizes, return 0 on success and a
 * negative rt_status on failure (text via rt_last_error); nothing throws
 * across the boundary.  Per-ray numerical failure (clipped ray, missed
 * surface, total internal reflection, Newton non-convergence) is NOT an
 * error: it is an in-band NaN exactly as in the reference
 * (rayopt/elements.py:206-209, :496, :367, :347-348).
 *
 * Threading: a context is not thread safe; use one context per (thread,
 * device).  Calls that enqueue work are asynchronous with respect to the
 * host and ordered on the context's private HIP stream; rt_download,
 * rt_sync and rt_kernel_ms synchronise.
 *
 * Device-resident result layout (all float64), L = number of elements of the
 * System, ld = padded ray count (rt_ld):
 *
 *      Y, U, I : [L][3][ld]     (surface, component, ray)   "SoA"
 *      T       : [L][ld]
 *
 * The Python side exposes them with the reference's shapes (L,N,3)/(L,N)
 * (rayopt/geometric_trace.py:41-47) as strided numpy views, no transpose.
 *
 * Large batches are cut into BLOCKS (rt_blocks: nblk blocks of bs rays,
 * ld = nblk * bs): every block holds its rays in the layout above with
 * ld -> bs, and block b begins bts doubles after block 0:
 *
 *      element (array, s, c) of ray j = base[array] + (j / bs) bts
 *                                       + (s 3 + c) bs + j % bs
 *
 * One block (nblk = 1: every batch whose arrays stay below 8.5 GB, i.e. up
 * to 8.1e6 rays through 13 elements) is the plain layout.  Rows further
 * apart than that are written more slowly (the device's address translation
 * falls behind streams that reach over more than ~10 one-GiB regions); see
 * csrc/rt_lay.h.  rt_download / rt_upload_row / the reductions / the gather
 * hide the blocks; rt_device_ptr users see them.
 */
#ifndef RT_MI355_H
#define RT_MI355_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 5 /* 2: rt_surface.rc, rt_selftest_arith, rt_comm_info;
                           the default asphere arithmetic; no rt_probe
                           3: rt_placement fills ms[8] (search times);
                              large batches in blocks (rt_blocks)
                           4: rt_opd_stats, rt_opd_device, rt_download_rays;
                              rt_placement(info[16], ms[16]) reports the
                              address ranges measured
                           5: rt_row_stats, rt_download_xy, rt_newton_census,
                              rt_input_completed;
                              rt_placement reports the search's time budget */
#define RT_MAX_ASPH 10      /* even-asphere terms r^2 .. r^20 */
#define RT_MAX_SURFACES 256 /* elements per System */

/* rt_surface.flags */
#define RT_F_ROTATED 0x01u /* element.rotated (elements.py:139): apply rot */
#define RT_F_CURVED  0x02u /* curvature != 0 (elements.py:482) */
#define RT_F_CONIC   0x04u /* conic != 0 (elements.py:484) */
#define RT_F_ASPH    0x08u /* aspherics is not None -> Newton (elements.py:478) */
#define RT_F_ALT     0x10u /* alternate_intersection (elements.py:497) */
#define RT_F_REFRACT 0x20u /* mu != 0 and mu != 1: Snell (elements.py:313,356) */
#define RT_F_MIRROR  0x40u /* mu == -1: reflection (elements.py:363) */
/* set by the library, not by the caller: row I[j] must be materialised
 * because element j or j-1 is rotated; otherwise i[j] == u[j-1] bit for bit
 * (system.py:461,464 with rot None) and I[j] is served from U[j-1] */
#define RT_F_STORE_I 0x80u
/* set by the library: this row is not stored (rt_set_keep_rows) */
#define RT_F_NOSTORE 0x100u
/* set by the library: row U[j] is not written because it is I[j] bit for
 * bit -- the element does not bend the ray (no material change, mu == 1:
 * stop, image, dummy surface; elements.py:311-313) and the trace does not
 * clip (Element.clip is what would poison u, :206-209) -- and is served
 * from I[j] (which may in turn be U[j-1]) */
#define RT_F_SKIP_U 0x200u
/* set by the library (default; rt_set_option "exact_asphere" clears it): this aspheric element
 * runs its Newton intercept and refraction on FMA / rcp / rsq arithmetic
 * (csrc/rt_math.h) -- same iteration, results inside the 1e-8 contract for
 * iterated aspheres instead of bit-identical to the reference */
#define RT_F_FAST 0x400u
/* set by the library (rt_set_option "range_shortcuts" = 0 clears it): the
 * element's wave-uniform operands (c of a sphere, mu^2 - 1 of a refracting
 * surface) have magnitudes in [2^-100, 2^100], so its IEEE quotients and
 * square roots may run without the compiler's range scaffolding wherever the
 * per-ray operands are in that range too -- the same bits from fewer
 * instructions (csrc/rt_math.h, "IEEE quotients and square roots ...") */
#define RT_F_RANGE 0x800u

/* which array (rt_download / rt_upload_row / rt_device_ptr) */
#define RT_Y 0 /* intercepts, element-normal frame, relative to vertex */
#define RT_U 1 /* outgoing direction (after clip + refract) */
#define RT_I 2 /* incoming direction in the element frame (pre-clip) */
#define RT_T 3 /* optical path length of the segment (t * n0) */

/* host ray layouts */
#define RT_LAYOUT_SOA 0 /* (3, N) component-major */
#define RT_LAYOUT_AOS 1 /* (N, 3) ray-major, the reference's layout */

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_ARG = -1,     /* bad argument */
    RT_ERR_STATE = -2,   /* call order (no system / no rays uploaded) */
    RT_ERR_HIP = -3,     /* HIP runtime error, see rt_last_error */
    RT_ERR_RCCL = -4,    /* RCCL error, see rt_last_error */
    RT_ERR_NOMEM = -5    /* device allocation failed */
} rt_status;

/*
 * One element of the System for ONE wavelength, evaluated on the host.
 * Everything that the reference computes as a Python scalar per element is
 * computed by the host with the reference's own expression and handed over,
 * so the kernel (and the oracle, which consumes the same table) reproduce
 * the reference's rounding:
 *   c, k           Spheroid.curvature / .conic       (elements.py:419-420)
 *   kw             1 + k      z-weight of the conic dot products (:489)
 *   kc2            (1 + k)*c**2                      (:451, :468)
 *   radius2        radius**2  clip limit             (:207)
 *   mu             n0/n | 1. | -1.  from get_n_mu    (:283-289)
 *   muf,smu,mu2m1  abs(mu), sign(mu), mu**2 - 1      (:358-366)
 *   n0             index in front of the element; t is stored as t*n0 (:315)
 *   offset[3]      element.offset  (distance*direction, :130)
 *   rot[9]         element.rot_normal row-major; identity if not rotated
 *   asph[i]        aspherics[i], coefficient of r^(2(i+1))      (:448-454)
 *   dasph[i]       2*(i + 1)*aspherics[i]                       (:469-473)
 *   rc             not the caller's: filled in by the library
 */
typedef struct rt_surface {
    double c, k, kw, kc2;
    double radius2;
    double mu, muf, smu, mu2m1;
    double n0;
    double offset[3];
    double rot[9];
    double asph[RT_MAX_ASPH];
    double dasph[RT_MAX_ASPH];
    int32_t nasph;
    uint32_t flags;
    double rc; /* set by the library ON THE DEVICE: 1/c as the device's
                  division sequence refines it (csrc/rt_math.h); callers
                  leave it 0 */
} rt_surface;

typedef struct rt_ctx rt_ctx;

int rt_abi_version(void);
/*
 * Self-test of the arithmetic the trace kernels rely on (csrc/rt_math.h):
 * n operand triples drawn on the device from `seed`, magnitudes 2^-span ..
 * 2^span plus edge values; every IEEE quotient and square root that runs
 * without the compiler's range scaffolding is compared with the compiler's
 * own sequence, bit for bit.  mismatches[0..2] (refraction quotient, table
 * quotient, square root) must be 0 for any span; mismatches[3] counts how
 * often the unguarded core alone would differ (0 for span <= 100).
 */
int rt_selftest_arith(rt_ctx *ctx, uint64_t seed, int64_t n, int span,
                      uint64_t mismatches[4]);
int rt_sizeof_surface(void); /* sizeof(rt_surface), layout check for FFI */
/* number of visible HIP devices (0 and RT_ERR_HIP text when there is none) */
int rt_device_count(int *count);

/* one context per device: owns the stream, events and all device buffers */
int rt_create(int device, rt_ctx **out);
int rt_destroy(rt_ctx *ctx);
const char *rt_last_error(const rt_ctx *ctx); /* ctx may be NULL: global */

/*
 * Replaces the per-call reads of element attributes in System.propagate /
 * Element.propagate (rayopt/system.py:459-464, elements.py:306-315).  surf[j]
 * describes element j (j = 0 is the object surface); nsurf = len(system).
 * Elements are mutable in the reference (refocus edits system[at].distance,
 * geometric_trace.py:98-99) so the host re-packs and re-uploads on every
 * propagate(); the copy is O(L) and tiny.
 */
int rt_upload_system(rt_ctx *ctx, const rt_surface *surf, int nsurf);
/*
 * Extension: ngroups surface tables (surf[g*nsurf + j]: the same geometry
 * evaluated at ngroups wavelengths, or ngroups variants of a system -- a
 * tolerancing run, the points of a finite-difference gradient; <= 65535).
 * The ray batch is then
 * read as ngroups equal, contiguous groups and group g is traced through
 * table g in the same launch; the group size must be a multiple of 64 rays
 * so every wavefront stays inside one group and the table reads remain
 * scalar.  BASELINE config C2 (10^6 rays x 3 wavelengths) is one launch.
 */
int rt_upload_system_groups(rt_ctx *ctx, const rt_surface *surf, int nsurf,
                            int ngroups);

/*
 * GeometricTrace.allocate(nrays) (geometric_trace.py:37-47): size the device
 * result arrays for len(system) x nrays.  Grows only; contents are undefined
 * after a growth.
 */
int rt_reserve(rt_ctx *ctx, int64_t nrays);
int64_t rt_nrays(const rt_ctx *ctx);
int64_t rt_ld(const rt_ctx *ctx); /* ray slots of the arrays (nblk * bs) */
/* info[0] = blocks the batch is cut into, [1] = rays per block = distance of
 * the rows in doubles, [2] = doubles from one block to the next (0: one) */
int rt_blocks(const rt_ctx *ctx, int64_t info[3]);
int rt_nsurf(const rt_ctx *ctx);

/*
 * GeometricTrace.rays_given (geometric_trace.py:49-70): seed row 0 from host
 * arrays of n rays x 3 components in `layout` (the host completes missing
 * components exactly as the reference does, :60-66).  Also sets I[0] = U[0]
 * and T[0] = 0 (:67,:69).  Calls rt_reserve(n) itself.
 */
int rt_set_rays(rt_ctx *ctx, const double *y, const double *u, int64_t n,
                int layout);
/* same p rays replicated `copies` times on the device (one copy per group) */
int rt_set_rays_repeat(rt_ctx *ctx, const double *y, const double *u,
                       int64_t p, int copies, int layout);
/* same, y/u already in device memory (rays generated on the device) */
int rt_set_rays_device(rt_ctx *ctx, const double *d_y, const double *d_u,
                       int64_t n, int layout);
/*
 * On-device ray generation: the launch rays of
 * nfields field points x npupil pupil coordinates, ray r = f*npupil + p, as
 * System.aim(yo, yp, z, a, filter=False) builds them one field at a time on
 * the host (rayopt/system.py:504, Infinite/FiniteConjugate.aim,
 * rayopt/conjugates.py:137-166,236-255, Pupil.map rayopt/pupils.py:97-107;
 * the object-space projection lives in the host-evaluated direction).  Only npupil*16 B + nfields*sizeof(rt_field) cross
 * PCIe instead of 48 B per ray; seeds row 0 like rt_set_rays.  The per-field
 * frame is evaluated by the host (O(nfields)):
 *   infinite: u = direction, base = (0,0,z) - z u, y = base + am px s + am py m,
 *             then projected onto element 0 along u
 *   finite:   base = object point, u = (0,0,z) [- base],
 *             dir = u + z tan(am px) s + z tan(am py) m, normalised (flipped
 *             if z < 0)
 * Row 0 is not written by this call: the first rt_trace from element 1
 * builds the rays in registers, writes row 0 and marches on in the same
 * launch (no 48 B/ray read); any other access to the batch first runs the
 * stand-alone generation kernel (same arithmetic, same values).
 */
typedef struct rt_field {
    int32_t finite;
    int32_t flip;
    double am;
    double z;
    double base[3];
    double u[3];
    double s[3];
    double m[3];
} rt_field;
int rt_sizeof_field(void);
int rt_generate_rays(rt_ctx *ctx, const rt_field *fields, int nfields,
                     const double *pupil_xy, int64_t npupil);

/*
 * Batched aiming on the device -- System.pupil / _aim_pupil / aim_chief /
 * aim_marginal (rayopt/system.py:507-593) for F field points at once.  The
 * reference runs ~130 serial one-ray traces per field on the host; here one
 * lane owns one field and runs the whole root finding in registers: the
 * launch frame (Conjugate.aim, conjugates.py:137-166,236-255), the one-ray
 * trace up to the stop (or through every aperture, rim = 1) and the solver
 * updates never leave the kernel.  Secant for the pupil distance that puts
 * the chief ray on the stop centre, bracket + Illinois regula falsi for the
 * four marginal rays.  Uses the surface table(s) last handed to
 * rt_upload_system[_groups]; a field names its table (wavelength) in
 * rt_aim_seed.group, so every field at every wavelength is one launch.
 *
 * rt_aim_seed: what the frame of one field is built from -- the parts that
 * do not depend on the pupil distance (the host evaluates the projection of
 * an object at infinity, InfiniteConjugate.map :208-234, once per field).
 */
typedef struct rt_aim_seed {
    int32_t finite;      /* object at finite distance */
    int32_t telecentric; /* finite only: chief rays parallel to the axis */
    int32_t group;       /* surface table (wavelength) this field is aimed at */
    int32_t pad_;
    double yo[2];        /* fractional field coordinates */
    double dir[3];       /* infinite: unit direction of the field */
    double point[3];     /* finite: object point, sag of element 0 included */
    double z0;           /* starting pupil distance from the vertex of
                            element 0 */
    double a0;           /* starting pupil aperture (radius, or angle basis) */
} rt_aim_seed;

typedef struct rt_aim_args {
    int32_t stop;    /* index of the aperture stop */
    int32_t rim;     /* 1: marginal rays graze the first limiting aperture
                        of elements 1..nsurf-2 instead of the stop */
    int32_t maxiter; /* per root find */
    int32_t no_chief; /* 1: keep the pupil distance of the seeds -- the
                         object pupil is telecentric, or its `aim` flag is
                         off and only the rim is aimed at
                         (aim_chief, system.py:509-510) */
    double tol;      /* convergence: secant step / |margin| */
} rt_aim_args;
int rt_sizeof_aim_seed(void);
int rt_sizeof_aim_args(void);
/*
 * z[F]: aimed pupil distance per field; a[F][2][2]: aimed apertures
 * [[-sag, -mer], [+sag, +mer]]; status[F]: 0 ok, 1 chief ray did not
 * converge, 2 no marginal bracket, 3 marginal ray did not converge (the
 * values of a failed field are NaN).  Host arrays; synchronous.
 */
int rt_aim_pupil(rt_ctx *ctx, const rt_aim_seed *seeds, int nfields,
                 const rt_aim_args *args, double *z, double *a,
                 int32_t *status);

/* overwrite one surface row of one array from a host SoA buffer */
int rt_upload_row(rt_ctx *ctx, int which, int surf, const double *src_soa);

/*
 * GeometricTrace.propagate(start, stop, clip) (geometric_trace.py:72-80)
 * fused with System.propagate (system.py:459-464): reads rows start-1 of Y and
 * U (element start-1's normal frame), applies from_normal of element start-1,
 * then marches elements start..stop-1 and writes rows start..stop-1 of
 * Y,U,I,T.  stop <= 0 or stop > nsurf means nsurf.  Asynchronous.
 */
int rt_trace(rt_ctx *ctx, int start, int stop, int clip);
/*
 * The same trace for one CHUNK of the batch's rays: the rays are cut into
 * `nchunks` pieces of whole 256-ray workgroups (rt_chunk_bounds: [lo, hi) of
 * chunk k for a batch of n rays; the last pieces may be short or empty) and
 * chunk `chunk` is traced.  All chunks of a step, in any order, give what
 * rt_trace gives; until the last of them has been traced the rows hold new
 * and old columns side by side, and every entry point that reads WHOLE rows
 * (rt_download, rt_device_ptr, the reductions, rt_gather_final) returns
 * RT_ERR_STATE (rt_gather_chunk reads only its own chunk's columns).  For multi-GPU jobs: rt_gather_chunk of chunk k runs on the
 * communication stream while chunk k+1 is traced.  Not for batches with
 * several surface tables (ray groups).
 */
int rt_trace_chunk(rt_ctx *ctx, int start, int stop, int clip, int chunk,
                   int nchunks);
int rt_chunk_bounds(int64_t n, int chunk, int nchunks, int64_t *lo,
                    int64_t *hi);
/*
 * Extension over the reference: choose which surface rows propagate()
 * stores.  keep[j] != 0 keeps row j; NULL restores the reference behaviour
 * (every row).  Rows that are not kept are still traced (the ray state lives
 * in registers) but cost no HBM traffic; reading them afterwards is an error.
 * A merit function that only needs the image-plane intercepts keeps one row
 * and the kernel moves from the HBM roofline to the FP64 one.
 */
int rt_set_keep_rows(rt_ctx *ctx, const unsigned char *keep, int n);
int rt_sync(rt_ctx *ctx);
/* HIP-event time of the last rt_trace kernel in ms (synchronises) */
int rt_kernel_ms(rt_ctx *ctx, double *ms);
/*
 * Measurement support: record HIP event `slot` (0..7) on the stream the
 * trace kernel is launched on; rt_event_elapsed synchronises on `b`.
 */
int rt_event_record(rt_ctx *ctx, int slot);
int rt_event_elapsed(rt_ctx *ctx, int a, int b, double *ms);
/*
 * Engine options (defaults are the measured best): key
 * "alias_i" (1 = do not write I[j] where it is identical to U[j-1], nor U[j]
 * where an unclipped trace leaves it identical to I[j], the default; 0 =
 * materialise every row),
 * "fuse_generate" (1 = the first trace after rt_generate_rays builds the
 * rays in registers and writes row 0 itself, the default; 0 = a separate
 * generation kernel writes row 0 and the trace reads it),
 * "regenerate" (1 = default: later traces of a generated batch from element 1
 * build the launch rays again in registers -- the values row 0 holds, bit for
 * bit -- as long as row 0 is what the generator wrote; 0 = they read row 0),
 * "exact_asphere" / "fast_asphere" (one switch, two names: even aspheres run
 * the reference's Newton iteration -- rayopt/elements.py:333-349: x0 = plane
 * intercept, |step| <= 1e-7, five iterates, NaN on failure -- by DEFAULT on
 * fused multiply-adds and rcp/rsq + refinement with one reciprocal per
 * iterate: results within the 1e-8 contract for iterated aspheres (~1e-13 in
 * practice), identical NaN masks; "exact_asphere" = 1 reproduces scipy's
 * iteration operation for operation: the reference's bits, ~20 % slower on
 * aspheric systems.  The environment variable RT_MI355_EXACT_ASPHERE=1 makes
 * the exact path the default of every context the process creates),
 * "compact" (clipped-ray compaction: 0 = never, the default; 1 = traces that
 * drop rows (rt_set_keep_rows) run the compacting kernel -- rays whose
 * direction is NaN are retired, their remaining kept rows filled with NaN,
 * and the survivors of a 256-ray workgroup are packed into fewer wavefronts
 * with ballots and an LDS exchange; 2 = every trace.  Results are identical
 * to the plain kernel's, NaN payloads aside), "compact_every" (k: the
 * survivors of a workgroup are counted -- one barrier -- at every k-th element
 * only; default 4, the measured optimum: asking costs ~0.5 us per workgroup).
 * "uniform_input" (1 = default: launch components that are ONE bit pattern
 * across a 64-ray tile of row 0 -- the direction of a collimated bundle, the
 * origin of a bundle from an object point, z = 0 of rays starting on a plane --
 * are fetched once per wavefront instead of once per ray: the seeding kernels
 * note them per tile, anything that rewrites row 0 voids the notes; same
 * values, same results; 0 = every component of every ray is read),
 * "resident_lds" (-1 = default: chosen per trace; 0..65536 = bytes of unused
 * dynamic LDS per workgroup of the trace kernels, i.e. a cap of 160 KB / bytes
 * on the workgroups resident per CU: traces that store their rows run with two
 * workgroups per CU, which the memory side likes better than the seven the
 * registers allow, four where the arrays' own store pattern was measured at the fast
 * level, rt_placement; FP64-bound traces are not capped), "placement"
 * (rt_placement), "placement_good_gbps" (default 6900: the store pattern at
 * which rt_reserve stops looking for a better address range / set of pieces),
 * "placement_orders" (default -1: six orders of a set's pieces for arrays
 * below 4 GiB, three up to 16 GiB; 0: none), "placement_budget_ms" (default 250: wall time after which an allocation
 * stops choosing memory -- surplus pieces, hops, further sets; the pieces it
 * needs it creates whatever that takes; a single hipMemCreate above 200 ms
 * ends the choosing at once),
 * "range_shortcuts" (1 = default: IEEE quotients and square
 * roots run without the compiler's range scaffolding where the operands are
 * checked to be inside [2^-100, 2^100] -- the same bits from a third fewer
 * instructions, RT_F_RANGE; 0 = the compiler's sequences everywhere),
 * "block_rays" (0 = default: large batches are cut into blocks of <= 7 GB, see
 * the layout above; B > 0: every batch of more than B rays is cut into
 * blocks of about B rays -- for tests; RT_MI355_BLOCK_RAYS=B does the same
 * for every context of the process; takes effect with the next rt_reserve),
 * "turn_points" (0 = default: a generated batch of several bundles over MORE
 * than 128 MB of pupil points is traced in turns of 4 Mi points -- turn k of
 * every bundle before turn k + 1 of any -- so that the points, read once per
 * bundle, are still in the Infinity Cache the next time; -1 = never; P > 0:
 * turns of P points whatever the size -- for tests, like
 * RT_MI355_TURN_POINTS=P for the process.  The order in which workgroups
 * take the rays, nothing else: every result lands where it did),
 * "consumers_one_pass" (1 = default, see rt_rms; 0 = always two passes),
 * "consumer_events" (measurement: 1 = the reductions bracket their kernels
 * with the events rt_kernel_ms reads; default 0).
 */
int rt_set_option(rt_ctx *ctx, const char *key, int value);

/*
 * Measurement: how well the per-wavefront trip count of the asphere Newton
 * solve (a 64-bit ballot ends the loop when no lane iterates any more;
 * rayopt/elements.py:333-349 solves ray by ray) fits the rays of this batch.
 * The batch is marched again from row 0 (table and clip of the last trace,
 * which must have started at element 1) by a kernel that stores nothing:
 * out[0] = lane slots spent in the iteration (64 x wavefront trips),
 * out[1] = iterates the rays needed (lane slots that did work; out[1] / out[0]
 * is the lane utilisation), out[2] = wavefront trips, out[3] = wavefront
 * solves (one per wavefront and aspheric element).  A ray whose iterate is
 * NaN -- every ray that arrives dead -- is retired after that iterate.
 */
int rt_newton_census(rt_ctx *ctx, int clip, uint64_t out[4]);

/*
 * Lazy D2H of surface rows [surf_lo, surf_hi) of one array into a caller
 * owned host buffer, compact SoA: (rows,3,n) for Y/U/I, (rows,n) for T.
 * This is what backs the numpy attributes y,u,i,t of the drop-in
 * GeometricTrace.  Synchronises.
 */
int rt_download(rt_ctx *ctx, int which, int surf_lo, int surf_hi, double *dst);
/*
 * The first two components of ONE row of Y, U or I: dst (2,n).  The host's
 * spot consumers read `t.y[-1, :, :2]` (rayopt/geometric_trace.py:172,
 * rayopt/analysis.py:237-283): two thirds of the row's bytes over PCIe.
 * Synchronises.
 */
int rt_download_xy(rt_ctx *ctx, int which, int surf, double *dst);

/* all surfaces of ONE ray: dst[L][ncomp] (print_trace, reference-ray terms) */
int rt_download_ray(rt_ctx *ctx, int which, int64_t ray, double *dst);
/* all surfaces of the rays ray0, ray0 + stride, ... (count of them), gathered
 * on the device: dst[L][ncomp][count] -- a sample of a batch too large to
 * bring down (parity checks of 10^8-ray batches, plots).  Rows that were not
 * stored come back as NaN. */
int rt_download_rays(rt_ctx *ctx, int which, int64_t ray0, int64_t stride,
                     int64_t count, double *dst);

/*
 * Device-side consumers of the result arrays: the
 * O(N) reductions GeometricTrace offers on top of a trace, so that only a few
 * scalars (or 24 B/ray for opd) cross PCIe instead of whole rows.
 *
 * rt_set_weights: GeometricTrace.w (geometric_trace.py:57-59); NULL = 1/N.
 * rt_rms:     GeometricTrace.rms(i, ref) (:171-183): sqrt(sum_k w_k |y_k -
 *             y0|^2) over x,y of row `surf`; ref < 0: y0 = unweighted mean.
 * rt_refocus_shift: the least-squares focus shift of refocus(at) (:82-97):
 *             u = i_xy/i_z, rays with finite u, centred y and u,
 *             t = -<w y, u>/<w u, u>.  Does not touch the system.
 *             Both read their rows ONCE: the sums are taken of coordinates
 *             shifted by ray 0 of the batch and centred by subtraction
 *             afterwards; the last kernel reports what the subtraction
 *             started from, and if it cost more than six bits (ray 0 far
 *             outside the bundle, or vignetted) the call repeats with the
 *             two passes (mean, then spread) of the textbook.  Tolerance
 *             against numpy either way: 1e-12 (rms), 1e-9 (the ratio).
 * rt_opd_rays: per-ray part of GeometricTrace.opd (:101-131, the rows the
 *             reference returns for resample=0): out[3][n] = (x, y, t) on the
 *             reference sphere, t in waves.
 */
int rt_set_weights(rt_ctx *ctx, const double *w);
int rt_rms(rt_ctx *ctx, int surf, int64_t ref, double *rms);
int rt_refocus_shift(rt_ctx *ctx, int surf, double *shift);
/* max over rays of hypot(x, y) on row `surf` (NaN if any ray is NaN):
 * GeometricTrace.resize (geometric_trace.py:231-234) */
int rt_row_rmax(rt_ctx *ctx, int surf, double *rmax);
/*
 * Spot statistics of every bundle of a batch in one call: the batch is
 * `ngroups` contiguous groups of `group_rays` rays (field x wavelength bundles
 * in the order rt_generate_rays / rt_upload_system_groups lay them out).  For
 * each group, over the rays whose intercept on row `surf` is finite:
 *   out[g][0] = number of such rays      out[g][1..2] = centroid (plain mean)
 *   out[g][3] = sum w d^2 / sum w        out[g][4] = max d^2
 *   out[g][5] = sum w                    (d = distance from the centroid)
 * i.e. sqrt(out[g][3]) is GeometricTrace.rms() (geometric_trace.py:171-183) of
 * that bundle traced on its own with its weights normalised (what a caller of
 * the reference obtains with one rays_point() + rms() per field and
 * wavelength) when no ray of it is lost; with lost rays the reference's value
 * is NaN (test out[g][0] < group_rays), this one covers the survivors.
 * Weights: rt_set_weights, NULL = uniform.  out: ngroups x 6 doubles (host).
 */
int rt_spot_stats(rt_ctx *ctx, int surf, int64_t group_rays, int ngroups,
                  double *out);

/*
 * The statistics of a row in ONE pass over its x, y (and the weights): what
 * rt_rms (about the mean and about a reference ray), rt_spot_stats (count,
 * centroid, spread) and rt_row_rmax answer in three calls and four passes
 * (rayopt/geometric_trace.py:171-183 rms, :185-193 resize; the spot diagrams
 * of rayopt/analysis.py:250-283 are centred on the reference ray).  The batch
 * is `ngroups` contiguous bundles of `group_rays` rays as for rt_spot_stats;
 * `ref` >= 0: the index of the reference ray INSIDE every bundle (< 0: none).
 * For each bundle, over the rays whose intercept on row `surf` is finite:
 *   out[g][0] = count                   out[g][1] = sum w
 *   out[g][2..3] = centroid (plain mean, as y.mean(0))
 *   out[g][4] = sum w |y - mean|^2 / sum w
 *   out[g][5] = sum w |y - y_ref|^2 / sum w  (NaN: no reference ray, or it
 *               did not arrive)
 *   out[g][6] = max (x^2 + y^2)         (NaN for an empty bundle)
 *   out[g][7..8] = weighted centroid    out[g][9] = sum w |y - shift|^2 /
 *               sum w, what the subtractions started from
 * The sums are taken of coordinates shifted by a ray of the bundle (the
 * reference ray if it arrived, else the first of the bundle's first rays
 * that did) and centred afterwards; where that cost more than six bits the
 * call repeats with rt_spot_stats' two passes.  Tolerances against numpy:
 * counts exact, 1e-12 (centroids, sum w), 1e-9 (spreads).  Up to 4096 bundles
 * the result is written into pinned memory by the finishing kernel and the
 * call does not synchronise the stream (it spins on a ticket).
 * out: ngroups x 10 doubles (host).
 */
int rt_row_stats(rt_ctx *ctx, int surf, int64_t group_rays, int ngroups,
                 int64_t ref, double *out);

typedef struct rt_opd_args {
    int32_t nrows;      /* t rows 0..nrows-1 are summed (t[:after + 1]) */
    int32_t after;      /* element in front of the reference sphere */
    int32_t image;      /* image element */
    int32_t finite;     /* system.object.finite */
    int32_t rot_after;  /* element `after` is rotated: apply r_after */
    int32_t rot_image;  /* element `image` is rotated: apply r_image */
    int64_t ref;        /* reference ray index */
    double n0, n_after; /* n[0], n[after] */
    double radius;      /* reference sphere radius */
    double lscale;      /* l / system.scale */
    double shift[3];    /* origins[after] - origins[image] */
    double r_after[9];  /* rot_normal of element after (from_normal) */
    double r_image[9];  /* rot_normal of element image (to_normal) */
} rt_opd_args;
int rt_opd_rays(rt_ctx *ctx, const rt_opd_args *args, double *out_soa);
int rt_sizeof_opd_args(void);
/*
 * rt_opd_stats: the same path differences WITHOUT the 24 B per ray over PCIe
 * (GeometricTrace.opd, rayopt/geometric_trace.py:101-131, up to the point
 * where the reference starts to resample): what most callers ask of an OPD
 * map -- its weighted mean, rms and peak-to-valley per bundle -- is reduced
 * on the device.  The batch is `ngroups` contiguous bundles of `group_rays`
 * rays; args->ref is the index of the reference ray INSIDE a bundle (bundle
 * g: ray g * group_rays + ref).  Over the rays of bundle g whose x, y and t
 * are finite (the rays the reference keeps, :133-135), t in waves:
 *   out[g][0] = number of such rays     out[g][1] = sum w
 *   out[g][2] = sum w t / sum w         out[g][3] = rms about that mean
 *   out[g][4] = min t   out[g][5] = max t   out[g][6] = max - min
 *   out[g][7] = sqrt(sum w t^2 / sum w)  (rms about the reference ray)
 * Weights: rt_set_weights, NULL = uniform.  out: ngroups x RT_OPD_STATS
 * doubles (host).  keep != 0: x | y | t of every ray stay on the device as
 * [3][nrays] doubles (rt_opd_device), for callers that plot or resample a
 * part of them: rt_copy_to_host moves what they need.  rt_opd_rays leaves
 * the same array behind.
 */
#define RT_OPD_STATS 8
int rt_opd_stats(rt_ctx *ctx, const rt_opd_args *args, int64_t group_rays,
                 int ngroups, int keep, double *out);
int rt_opd_device(rt_ctx *ctx, double **x_y_t, int64_t *nrays);

/* raw device pointer to row `surf` of an array, i.e. to its part in block 0
 * (rt_blocks: the other blocks' parts follow bts doubles apart; one block
 * unless the batch is large) -- interop, collectives.  The
 * arrays of a large batch live in a virtual-memory mapping (rt_placement):
 * an ordinary device pointer for kernels, hipMemcpy and RCCL on this device,
 * but not something hipIpcGetMemHandle accepts -- copy a row out (rt_scratch)
 * to share it with another process */
int rt_device_ptr(rt_ctx *ctx, int which, int surf, void **out);

/*
 * Multi-GPU (one process per GPU).  The trace itself needs no exchange: rays
 * are independent and the ray batch is sharded contiguously.  The only
 * exchange is the gather of the last-surface intercepts to a root rank
 * (RCCL over xGMI): grouped ncclSend/ncclRecv, direct peer->root links, no
 * ring.  id is a 128-byte ncclUniqueId created on rank 0 and distributed by
 * the host (any out-of-band channel).  librccl.so is bound at first use;
 * the environment variable RT_TRANSPORT_LIBRARY names another library with
 * RCCL's entry points to bind instead (it must load: there is no fallback) --
 * the GPU tests use a shared-memory stand-in to run several ranks on one
 * device, which RCCL refuses.
 */
int rt_comm_unique_id(void *id128);
int rt_comm_init(rt_ctx *ctx, const void *id128, int nranks, int rank);
int rt_comm_destroy(rt_ctx *ctx);
/*
 * What the communicator says about itself, so that a multi-GPU line can be
 * checked against the transport rather than against what the launcher
 * claimed: info[0] = ncclCommCount (rt_comm_init already refuses a
 * communicator whose count or rank differ from what it was asked for),
 * info[1] = ncclCommUserRank, info[2] = ncclGetVersion's code (e.g. 22707),
 * info[3] = HIP devices visible to this process.  link_type[d] / hops[d] for
 * d < min(info[3], max_devices): hipExtGetLinkTypeAndHopCount from this
 * context's device to device d (HSA_AMD_LINK_INFO_TYPE_*: 4 = xGMI, 2 =
 * PCIe; -1 for the device itself or no answer).  max_devices may be 0.
 */
int rt_comm_info(rt_ctx *ctx, int info[4], int *link_type, int *hops,
                 int max_devices);
/*
 * Gather row `surf` of array `which` from every rank to `root`.  counts[r] =
 * rays held by rank r.  On root, d_dst is a device buffer of
 * ncomp*sum(counts) doubles laid out [component][global ray]; ignored
 * elsewhere.  Runs on the context's communication stream after the trace
 * that produced the row; rt_comm_sync waits for it.
 */
int rt_gather_final(rt_ctx *ctx, int which, int surf, const int64_t *counts,
                    int root, double *d_dst);
/*
 * The same for chunk `chunk` of `nchunks` of every rank's rays (the pieces
 * rt_trace_chunk traces; every rank cuts its own counts[r] rays with
 * rt_chunk_bounds): the gather of chunk k overlaps the trace of chunk k+1.
 * All chunks of a step together fill d_dst exactly as rt_gather_final does.
 * Up to RT_GATHER_SLOTS = 4 gathers may be in flight.
 */
int rt_gather_chunk(rt_ctx *ctx, int which, int surf, const int64_t *counts,
                    int root, double *d_dst, int chunk, int nchunks);
/*
 * Timing of the last gather (all its chunks) from HIP events: total_ms =
 * first to last operation on the communication stream, exposed_ms = what is
 * left of it after this rank's last trace kernel finished (0 if the exchange
 * was hidden completely).  Synchronises.
 */
int rt_gather_ms(rt_ctx *ctx, double *total_ms, double *exposed_ms);
int rt_comm_sync(rt_ctx *ctx);

/*
 * How much of the launch rows a trace from element 1 has to read: tiles7[c],
 * c = 0..5 (y0 y1 y2 u0 u1 u2) = number of 64-ray tiles of row 0 in which
 * that component is uniform (fetched once per tile), tiles7[6] = number of
 * tiles.  All zero but tiles7[6] when the notes are void or switched off.
 */
int rt_input_uniform(rt_ctx *ctx, int64_t *tiles7);
/*
 * ... and *tiles = number of 64-ray tiles whose u2 is not read either: the
 * seed kernels found every u2 of the tile bit for bit the completion of u0,
 * u1 -- sqrt(1 - (u0^2 + u1^2)), what the reference writes for two-component
 * directions (rayopt/geometric_trace.py:57-60), or sqrt((1 - u0^2) - u1^2) --
 * and the trace rebuilds it with the same operations (8 B per ray less among
 * the saturated stores; tiles in which u2 is uniform anyway are not counted).
 */
int rt_input_completed(rt_ctx *ctx, int64_t *tiles);

/*
 * Where the result arrays live.  The speed of a trace's 7-10 simultaneous row
 * streams is not one number (bare store pattern of C3: 7.0 ... 5.65 TB/s on
 * one box) and depends on WHICH pieces of device memory lie behind the arrays
 * -- pieces fall into classes; streams dealt over three classes run faster
 * than over two, and both faster than streams inside one, which is what a
 * plain hipMalloc of 10 GB gets; sets of equal class counts still differ
 * (csrc/rt_place.h).  Arrays of > 1.5 GiB are therefore built from pieces
 * (hipMemCreate, 1 GiB; 512 MiB below 3 GiB) whose class the library measures
 * at rt_reserve with a ~1 ms pair test each, an even mix of (if they can be
 * had: three) classes mapped behind one address range, the surplus released;
 * then the
 * batch's OWN store pattern (56 B per ray and element) is written over the
 * arrays and timed; while it stays below 6900 GB/s ANOTHER set of pieces is
 * searched, classified and measured while the first is held (arrays up to
 * 16 GiB; at most three sets, five where the best of three is 2.5 % below the
 * mark, eight below 4 GiB with two GiB of ballast between one and the next);
 * the best stays (option "placement", default 1;
 * RT_MI355_PLACEMENT=0 for the whole process; plain hipMalloc if anything on
 * the way fails; results never depend on it).
 * info[0] = pieces behind the arrays (0: hipMalloc), [1] = MiB per piece,
 * [2] = pieces created on the way, [3] = classes seen, [4..6] = pieces of
 * class 0 / 1 / 2 kept, [7] = 1: store-bound traces run four workgroups per CU
 * instead of two (the measured pattern is at or above 5950 GB/s; where no
 * pattern was measured: at least a third of the pieces lie outside the
 * largest class), [8] = hops: times the search created and held four to
 * eight blocks of 1 GiB of ballast so that it moved on through the device
 * memory (pieces come in runs of one class; search and ballast together
 * never hold more than half of the
 * memory that was free), [9] = what the classes alone said, [10] = the
 * search ended early: 1 = its time budget was up (everything that is choice
 * -- surplus pieces, hops, further sets -- ends 250 ms after the allocation
 * began), 2 = a single hipMemCreate took more than 200 ms; [11] = 1: these
 * arrays have had their search (a later rt_reserve that reuses the buffer for
 * another layout only measures the pattern), [12] =
 * sets of pieces tried, [13] = 1 if a placement of this context failed its
 * check -- tokens written by a kernel into every 2 MiB of the range, read back
 * by a copy -- and the context (and from then on the process) went back to
 * plain allocations: on ROCm 7.2 a kernel goes on using the translations of
 * an EARLIER mapping of an address range after hipMemUnmap + hipMemMap unless
 * a buffer is freed in between, which the library does after every mapping
 * (csrc/rt_place.h: rt_place_flush); [14] = hipMemUnmap / hipMemRelease /
 * hipMemAddressFree calls of this PROCESS that returned an error (rt_reserve
 * also leaves the first one's text for rt_last_error); [15] = other ORDERS of
 * a set's pieces along the range that were mapped and measured (before another
 * set is searched the same pieces are tried in up to six -- arrays of 4 GiB
 * and more: three -- seeded permutations: no memory, 5 ms each; the same five
 * pieces ran C2's pattern between 5.5 and 6.8 TB/s depending on the order;
 * option "placement_orders").
 * ms[0] / ms[1] = the pair test's launch time inside one piece / across two
 * classes, ms[2] = GB/s of the batch's store pattern over the arrays (0: not
 * measured -- a pattern below 0.5 GB tells nothing), ms[3] = wall
 * milliseconds the search took, of which ms[4] creating, mapping and testing
 * pieces, ms[5] creating and releasing ballast, ms[6] unmapping, releasing
 * the surplus and mapping the final range; ms[7] = measuring the pattern;
 * ms[8..12] = GB/s of each set of pieces tried (0: not tried; arrays below
 * 4 GiB try up to eight, ms[12] is then the best of the fifth and later
 * sets), ms[13] = the
 * longest single hipMemCreate of the search (ms), ms[14] = everything
 * rt_reserve spent on the placement of this buffer (ms: all sets, the
 * measurements, the coherence proof), ms[15] = 0.
 */
int rt_placement(rt_ctx *ctx, int info[16], double ms[16]);

/* device scratch owned by the context (e.g. gather destination on root) */
int rt_scratch(rt_ctx *ctx, int64_t bytes, void **out);
/* D2H copy helper for buffers obtained from rt_scratch / rt_device_ptr */
int rt_copy_to_host(rt_ctx *ctx, void *dst, const void *d_src, int64_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* RT_MI355_H */
