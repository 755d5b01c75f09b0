/*
 * rt_mi355_probes.h -- LABORATORY entry points, exported only by
 * librt_mi355_probes.so (built with -DRT_BUILD_PROBES by
 * `python -m rayopt_amd._build probes`; the measurement scripts load it via
 * RT_MI355_LIB).  The shipped librt_mi355.so has none of them.
 *
 * Additional rt_set_option keys of that build (A/B measurements; every one
 * was measured and rejected, profiles/HISTORY.md): "rays_per_thread" (1,2,4),
 * "nontemporal" (0,1), "xcd_remap" (0,1), "block" (64..1024), "lds_pad" (bytes
 * of unused dynamic LDS per workgroup: caps the resident workgroups per CU),
 * "tile_rays" (0 = SoA; TR = a power of two: results are written tile-major
 * [tile of TR rays][L][10][TR]; nothing can be read back in that layout),
 * "probe_store" (rt_probe pattern modes: 0 plain stores, 1 non-temporal,
 * 2 sc1 write-through, 3 sc0 sc1), "uniform_fix" (results are wrong unless
 * the data happen to be so: 6-bit mask of input components read from the
 * wavefront's first column), "gate_log2" / "gate_window" (input reads wait
 * for chip-wide windows of the 100 MHz reference counter), "base_offset_kb"
 * (the result arrays start this far into their allocation, which is 1 GiB
 * larger in this build: placement experiments, scripts/r03_offset_sweep.py),
 * "alloc_round" (the allocation behind the arrays is rounded up to a multiple
 * of 2^k bytes, 99 = to a power of two), "alloc_vmm_mb" / "alloc_vmm_align_mb"
 * / "alloc_vmm_shuffle" / "alloc_vmm_seed" (the arrays live in a virtual
 * address range backed by hipMemCreate chunks of that many MiB, mapped in
 * order or in a pseudo-random order; a new seed re-maps the same chunks at
 * once).  What they showed: profiles/r03_probes/README.md, "Placement".
 */
#ifndef RT_MI355_PROBES_H
#define RT_MI355_PROBES_H

#include "rt_mi355.h"

#ifdef __cplusplus
extern "C" {
#endif

int rt_probes_built(void); /* 1: this is the laboratory build */

/*
 * Memory-system calibration on the context's own result arrays (row 0 is
 * preserved): mode 0 = the trace kernel's 80 B store pattern without
 * arithmetic, 1 = grid-stride 16-byte fill, 2 = 16-byte copy, 3 = fill with one
 * 16-byte store per lane, 4 = same, non-temporal; 5 / 6 = mode 0 with the
 * 48 B/ray input read from an L2-resident window / not at all; 7 / 8 = the
 * default kernel's own pattern (56 B per op, 8-byte stores) with / without
 * the input read; 9 = mode 7 with non-temporal loads; 10 / 11 / 12 = mode 7
 * with 2 / 4 / 8 rays per lane marched one after the other, inputs loaded up
 * front; 13 / 14 = mode 7 with the input rows in an uncached / an ordinary
 * allocation of their own.  Modes 0 and 5-14 honour "tile_rays" and "block".
 * Returns kernel time and the bytes moved.  Overwrites rows >= 1.
 */
int rt_probe(rt_ctx *ctx, int mode, double *ms, double *bytes);

#ifdef __cplusplus
}
#endif
#endif /* RT_MI355_PROBES_H */
