"""Round-2 measurement session (one GPU): prints one JSON object per line.

 A  store-pattern ceiling: the default SoA layout against the tile-major
    layout (rt_set_option "tile_rays": a workgroup's whole output is one
    contiguous region) -- pattern probes without arithmetic (rt_probe modes
    0/6 = 80 B per op with/without the input read, 7/8 = the default kernel's
    56 B per op, 8-byte stores) and the real trace kernel, C3 at 10^7 rays.
 B  C4 (six even aspheres): exact Newton against "fast_asphere".
 C  dead rays: C3 overfilled (bundle radius x1.5), rays in random order and
    sorted so that the vignetted ones are contiguous (what an ideal
    compaction would achieve), full store and keep=[-1].
 D  wall time of one propagate() on small batches.
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rayopt_amd as ra
from rayopt_amd import prescriptions as P
from rayopt_amd.bundles import disc_bundle, multi_field_bundle
from bench import workload_rays


def out(**kw):
    print(json.dumps(kw), flush=True)


def kernel_ms(g, clip, reps=40, keep=None, last=10):
    ms = []
    for _ in range(reps):
        g.propagate(clip=clip, keep=keep)
        ms.append(g.kernel_ms())
    return float(np.median(ms[-last:]))


def probe_ms(eng, mode, reps=8):
    t = []
    for _ in range(reps):
        m, b = eng.probe(mode)
        t.append(m)
    m = float(np.median(t[2:]))
    return m, b/m/1e6


def part_a(n):
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    y, u = workload_rays(n, 0)
    S = len(system) - 1
    for tile in (0, 256, 1024, 4096, 32768):
        g = ra.GeometricTrace(system)
        eng = g.engine
        eng.set_option("tile_rays", tile)
        g.rays_given(y, u)
        for _ in range(60):         # clock settle
            g.propagate(clip=True)
        ms = kernel_ms(g, True)
        alg = n*(56*S + 48)
        out(part="A", what="trace kernel (56 B/op)", tile_rays=tile,
            kernel_ms=ms, GBs=alg/ms/1e6)
        for block in ((256, 1024) if tile in (0, 256) else (256,)):
            eng.set_option("block", block)
            for mode, name in ((7, "pattern 56B 8B-stores, input read"),
                               (8, "pattern 56B 8B-stores, no read"),
                               (0, "pattern 80B 16B-stores, input read"),
                               (6, "pattern 80B 16B-stores, no read")):
                m, gbs = probe_ms(eng, mode)
                out(part="A", what=name, tile_rays=tile, block=block, ms=m,
                    GBs=gbs)
        eng.set_option("block", 256)
        if tile == 0:
            for mode, name in ((3, "fill, one 16B store per lane"),
                               (4, "same, non-temporal"), (1, "grid-stride "
                                                           "fill"),
                               (2, "copy")):
                m, gbs = probe_ms(eng, mode)
                out(part="A", what=name, ms=m, GBs=gbs)
        g.engine.close()
        del g


def part_b(n):
    system = ra.system_from_yaml(P.ASPHERE_PHONE)
    S = len(system) - 1
    for deg in (0., 17.5):
        y, u = disc_bundle(n, 0.6, deg, 3)
        y[:, 1] -= 0.5*np.tan(np.radians(deg))
        g = ra.GeometricTrace(system)
        g.rays_given(y, u)
        for _ in range(60):
            g.propagate(clip=True)
        res = {}
        for fast in (0, 1, 0, 1):
            g.engine.set_option("fast_asphere", fast)
            res.setdefault(fast, []).append(kernel_ms(g, True, reps=30))
        g.engine.set_option("fast_asphere", 0)
        g.propagate(clip=True)
        exact = np.asarray(g.y[-1]).copy()
        g.engine.set_option("fast_asphere", 1)
        g.propagate(clip=True)
        fastv = np.asarray(g.y[-1])
        same_mask = bool(np.array_equal(np.isnan(exact), np.isnan(fastv)))
        fin = np.isfinite(exact)
        err = float(np.abs(fastv[fin] - exact[fin]).max()/
                    np.abs(exact[fin]).max())
        alg = n*(56*S + 48)
        for fast in (0, 1):
            ms = min(res[fast])
            out(part="B", config="C4 asphere, field %.1f deg" % deg,
                fast_asphere=fast, kernel_ms=ms, all_ms=res[fast],
                ops_per_s=n*S/ms*1e3, GBs=alg/ms/1e6,
                image_row_max_rel_dev_fast_vs_exact=err,
                nan_masks_identical=same_mask,
                dead_fraction=float(np.isnan(exact[:, 0]).mean()))
        # image row only: the arithmetic side
        for fast in (0, 1):
            g.engine.set_option("fast_asphere", fast)
            ms = kernel_ms(g, True, reps=20, keep=[0, -1])
            out(part="B", config="C4 field %.1f deg, keep=[-1]" % deg,
                fast_asphere=fast, kernel_ms=ms)
        g.engine.close()
        del g


def part_c(n):
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    S = len(system) - 1
    fields = [f*P.DOUBLE_GAUSS_FIELD_DEG for f in (0, .35, .5, .7, 1.)]
    for scale in (1., 1.15, 1.3, 1.5, 2.):
        y, u = multi_field_bundle(n, 17.*scale, fields, 0,
                                  P.DOUBLE_GAUSS_PUPIL_Z)
        g = ra.GeometricTrace(system)
        g.rays_given(y, u)
        for _ in range(60):
            g.propagate(clip=True)
        dead_at = np.isnan(np.asarray(g.t[1:])).argmax(0)   # first dead row
        dead = np.isnan(np.asarray(g.u[-1])[:, 0])
        frac = float(dead.mean())
        r = dict(part="C", bundle_radius_scale=scale, dead_fraction=frac)
        r["random_full_ms"] = kernel_ms(g, True, reps=20)
        r["random_image_only_ms"] = kernel_ms(g, True, reps=20, keep=[0, -1])
        # the compacting kernel (rt_set_option "compact"), same random order
        g.engine.set_option("compact", 1)
        r["random_image_only_compact_ms"] = kernel_ms(g, True, reps=20,
                                                      keep=[0, -1])
        for every in (2, 3, 4, 6, 64):
            g.engine.set_option("compact_every", every)
            r["random_image_only_compact_every%d_ms" % every] = kernel_ms(
                g, True, reps=20, keep=[0, -1])
        g.engine.set_option("compact_every", 1)
        g.engine.set_option("compact", 0)
        # ideal compaction: the same rays, the dead ones contiguous, ordered
        # by the surface they die at
        key = np.where(dead, dead_at, 10**6)
        order = np.argsort(-key, kind="stable")
        g.rays_given(y[order], u[order])
        r["sorted_full_ms"] = kernel_ms(g, True, reps=20)
        r["sorted_image_only_ms"] = kernel_ms(g, True, reps=20, keep=[0, -1])
        dead2 = np.isnan(np.asarray(g.u[-1])[:, 0])
        r["same_dead_count"] = bool(dead2.sum() == dead.sum())
        out(**r)
        g.engine.close()
        del g


def part_d():
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    for n in (64, 10_000, 1_000_000):
        y, u = disc_bundle(n, 17., 5., 1, P.DOUBLE_GAUSS_PUPIL_Z)
        g = ra.GeometricTrace(system)
        g.rays_given(y, u)
        for _ in range(50):
            g.propagate(clip=True)
        g.engine.sync()
        reps = 500
        t0 = time.perf_counter()
        for _ in range(reps):
            g.propagate(clip=True)
        g.engine.sync()
        dt = (time.perf_counter() - t0)/reps
        # the system changes between calls (refocus-style): the table is
        # re-sent every time, double buffered
        t0 = time.perf_counter()
        for k in range(reps):
            system[-1].distance += 1e-9
            g.propagate(clip=True)
        g.engine.sync()
        dt2 = (time.perf_counter() - t0)/reps
        out(part="D", rays=n, propagate_us=dt*1e6, kernel_us=g.kernel_ms()*1e3,
            propagate_changing_system_us=dt2*1e6)


if __name__ == "__main__":
    parts = sys.argv[1] if len(sys.argv) > 1 else "ABCD"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
    if "D" in parts:
        part_d()
    if "B" in parts:
        part_b(n)
    if "C" in parts:
        part_c(n)
    if "A" in parts:
        part_a(n)
