"""scripts/lab.py -- the measurement laboratory (round 4 on): one parameterised
driver instead of one file per session.  Every sub-command prints JSON lines;
the records that are kept live under profiles/.

    python scripts/lab.py placement [--contexts 5] [--vmm 3] [--launches 24]
        N contexts (N hipMalloc'ed sets of result arrays, then --vmm sets in
        1 GiB hipMemCreate chunks), each timed at two and at four resident
        workgroups per CU with a FIXED launch schedule, so that the dispatch
        order of a `rocprofv3 --pmc` pass of the same command can be mapped
        back to (context, setting): `pmc-summary`.
    python scripts/lab.py pmc-summary <dir with pass*/...counter_collection.csv + pass*.jsonl>
    python scripts/lab.py nsweep      [--shape c3|c2] (launch time / frac over N)
    python scripts/lab.py divab       same-process A/B of two builds of the library

Laboratory options (alloc_vmm_mb, lds_pad ...) need the laboratory build:
python -m rayopt_amd._build probes.
"""
import argparse
import collections
import csv
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _imports():
    import rayopt_amd as ra
    from rayopt_amd import prescriptions as P, _build
    from rayopt_amd.engine import Engine
    return ra, P, _build, Engine


def lab_lib():
    from rayopt_amd import _build
    return os.path.join(os.path.dirname(_build.LIB), "librt_mi355_probes.so")


def out(**kw):
    print(json.dumps(kw), flush=True)


def block_ms(eng, launches, clip=True, start=1, stop=0):
    """`launches` back-to-back traces bracketed by events: ms per launch."""
    eng.event_record(0)
    for _ in range(launches):
        eng.trace(start, stop, clip)
    eng.event_record(1)
    return eng.event_elapsed(0, 1)/launches


def steady(eng, seconds=.8, per=10, **kw):
    t_end = time.time() + seconds
    ms = []
    while time.time() < t_end:
        ms.append(block_ms(eng, per, **kw))
    return float(np.median(ms[len(ms)//3:]))


# ---------------------------------------------------------------------------
def cmd_placement(a):
    ra, P, _build, Engine = _imports()
    from bench import workload_rays
    n = a.rays
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    y, u = workload_rays(n, 0)
    lib = lab_lib()
    keep, count = [], 0          # count = rt_trace_kernel launches so far
    plan = [("hipMalloc", {})]*a.contexts + [
        ("vmm_%dMiB" % a.vmm_mb, {"alloc_vmm_mb": a.vmm_mb,
                                  "alloc_vmm_align_mb": min(a.vmm_mb, 1024)})
    ]*a.vmm
    for k, (kind, opts) in enumerate(plan):
        eng = Engine(0, lib_path=lib)
        for key, v in opts.items():
            eng.set_option(key, v)
        g = ra.GeometricTrace(system, engine=eng)
        keep.append(g)
        g.rays_given(y, u)
        g.propagate(clip=True)
        count += 1
        rec = {"context": k, "kind": kind, "blocks": []}
        eng.set_option("resident_lds", 65536)
        for _ in range(a.warm):
            eng.trace(1, 0, True)
        count += a.warm
        for lds in (65536, 32768, 65536, 32768):
            eng.set_option("resident_lds", lds)
            ms = []
            first = count
            for _ in range(a.launches//4):
                ms.append(block_ms(eng, 4))
                count += 4
            ser = []
            if a.serial:
                for _ in range(a.serial):   # one launch at a time, like --pmc
                    ser.append(block_ms(eng, 1))
                count += a.serial
            rec["blocks"].append({"resident_lds": lds, "first": first,
                                  "count": count - first,
                                  "ms": float(np.median(ms)),
                                  "single_launch_ms":
                                  float(np.median(ser)) if ser else None})
        if a.probe:
            for lds in (65536, 32768):
                eng.set_option("lds_pad", lds)
                pm = [eng.probe(8)[0] for _ in range(8)]
                rec["store_pattern_ms_lds_%d" % lds] = float(np.median(pm[2:]))
            eng.set_option("lds_pad", 0)
        eng.set_option("resident_lds", -1)
        out(**rec)


def cmd_layouts(a):
    """SoA against the tile-major layouts (laboratory option tile_rays) in the
    SAME allocations, each classified by the pure store pattern first."""
    ra, P, _build, Engine = _imports()
    from bench import workload_rays
    n = a.rays
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    y, u = workload_rays(n, 0)
    lib = lab_lib()
    keep = []
    plan = [("hipMalloc", {})]*a.contexts + [
        ("vmm_1GiB", {"alloc_vmm_mb": 1024, "alloc_vmm_align_mb": 1024})]*a.vmm
    for k, (kind, opts) in enumerate(plan):
        eng = Engine(0, lib_path=lib)
        for key, v in opts.items():
            eng.set_option(key, v)
        g = ra.GeometricTrace(system, engine=eng)
        keep.append(g)
        rec = {"context": k, "kind": kind}
        for tile in [0] + a.tiles:
            eng.set_option("tile_rays", tile)
            g.rays_given(y, u)
            g.propagate(clip=True)
            res = {}
            # lds_pad (not resident_lds): every layout, SoA included, runs
            # the SAME laboratory kernel (48 B per ray read, no tile notes)
            for lds in (65536, 32768, 16384):
                eng.set_option("lds_pad", lds)
                for _ in range(6):
                    eng.trace(1, 0, True)
                res["trace_%d" % lds] = float(np.median(
                    [block_ms(eng, 4) for _ in range(6)]))
                pm = [eng.probe(8)[0] for _ in range(8)]
                res["store_%d" % lds] = float(np.median(pm[2:]))
            eng.set_option("lds_pad", 0)
            rec["tile_%d" % tile] = res
        out(**rec)


def cmd_placed(a):
    """The shipped library: arithmetic self-test, then contexts with and
    without the measured placement, each timed per resident setting and with
    the range shortcuts off / on (same process, alternating blocks); the
    image rows of both arithmetics are compared bit for bit."""
    ra, P, _build, Engine = _imports()
    from bench import workload_rays
    from rayopt_amd._lib import RT_Y, RT_U, RT_T
    eng0 = Engine(0)
    for span in (100, 300, 1000):
        t0 = time.time()
        bad = [eng0.selftest_arith(seed, 1 << 26, span) for seed in range(4)]
        out(selftest_span=span, draws=4 << 26, mismatches=bad,
            seconds=round(time.time() - t0, 2))
    eng0.close()
    n = a.rays
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    y, u = workload_rays(n, 0)
    keep = []
    for k in range(a.contexts):
        placed = k % 2 == 0 or a.all_placed
        eng = Engine(0)
        eng.set_option("placement", 1 if placed else 0)
        t0 = time.time()
        eng.upload_system(ra.pack.pack_system(
            system, system.wavelengths[0],
            system.refractive_index(system.wavelengths[0], 0))[0])
        eng.reserve(n)
        eng.sync()
        t_reserve = time.time() - t0
        g = ra.GeometricTrace(system, engine=eng)
        keep.append(g)
        g.rays_given(y, u)
        g.propagate(clip=True)
        eng.sync()
        rec = {"context": k, "placement": eng.placement(),
               "reserve_seconds": round(t_reserve, 4)}
        rec["first_block_ms"] = block_ms(eng, 4)
        for _ in range(3):
            block_ms(eng, 10)
        res = {}
        for lds in (-1, 65536, 32768, 0):
            eng.set_option("resident_lds", lds)
            res[str(lds)] = steady(eng, .5)
        eng.set_option("resident_lds", -1)
        rec["ms_by_resident_lds"] = res
        ab = {"0": [], "1": []}
        rows = {}
        for rep in range(3):
            for v in (0, 1):
                eng.set_option("range_shortcuts", v)
                ab[str(v)].append(steady(eng, .6))
                if rep == 0:
                    L = len(system)
                    rows[v] = [eng.download(w, L - 1, L) for w in (RT_Y, RT_U)]
                    rows[v].append(eng.download(RT_T, 1, L))
        rec["ms_range_shortcuts_off_on"] = [float(np.median(ab["0"])),
                                            float(np.median(ab["1"]))]
        rec["bit_identical"] = all(
            np.array_equal(p.view(np.uint64), q.view(np.uint64)) or
            np.array_equal(np.where(np.isnan(p), 0, p).view(np.uint64),
                           np.where(np.isnan(q), 0, q).view(np.uint64))
            for p, q in zip(rows[0], rows[1]))
        out(**rec)
    # allocation churn: contexts created and destroyed
    for k in range(3):
        eng = Engine(0)
        g = ra.GeometricTrace(system, engine=eng)
        g.rays_given(y[:4_000_000], u[:4_000_000])
        g.propagate(clip=True)
        pl = eng.placement()
        ms = steady(eng, .3)
        eng.close()
        out(churn=k, placement=pl, ms_4e6_rays=ms)


def cmd_resident(a):
    """Resident workgroups per CU (unused dynamic LDS) per kind of trace,
    arrays in measured placement (shipped library)."""
    ra, P, _build, Engine = _imports()
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import digest_cases as dc
    from bench import workload_rays, FIELD_FRACTIONS, BUNDLE_RADIUS
    caps = [65536, 53248, 40960, 36864, 32768, 28672, 24576, 20480, 16384, 0]

    def sweep(name, g, propagate):
        eng = g.engine
        propagate()
        for _ in range(20):
            propagate()
        res = {}
        for lds in caps:
            eng.set_option("resident_lds", lds)
            propagate()
            t_end = time.time() + .35
            ms = []
            while time.time() < t_end:
                eng.event_record(0)
                for _ in range(8):
                    propagate()
                eng.event_record(1)
                ms.append(eng.event_elapsed(0, 1)/8)
            res[str(lds)] = float(np.median(ms[len(ms)//3:]))
        eng.set_option("resident_lds", -1)
        propagate()
        auto = block_ms(eng, 1) if False else None
        ms = []
        for _ in range(6):
            eng.event_record(0)
            for _ in range(8):
                propagate()
            eng.event_record(1)
            ms.append(eng.event_elapsed(0, 1)/8)
        out(kind=name, placement=eng.placement(), auto_ms=float(np.median(ms)),
            ms_by_resident_lds=res)

    n = a.rays
    s3 = ra.system_from_yaml(P.DOUBLE_GAUSS)
    y, u = workload_rays(n, 0)
    g = ra.GeometricTrace(s3, engine=Engine(0))
    g.rays_given(y, u)
    sweep("C3 host-seeded clip", g, lambda: g.propagate(clip=True))
    sweep("C3 host-seeded unclipped", g, lambda: g.propagate(clip=False))
    sweep("C3 image row only", g,
          lambda: g.propagate(clip=True, keep=[0, -1]))
    g.engine.set_option("alias_i", 0)
    sweep("C3 every i row stored (80 B/op)", g, lambda: g.propagate(clip=True))
    g.engine.set_option("alias_i", 1)
    nf = len(FIELD_FRACTIONS)
    m = n//nf//64*64
    pts = dc.disc_points(m, 91)
    g.rays_fields(np.c_[np.zeros(nf), FIELD_FRACTIONS], pts,
                  P.DOUBLE_GAUSS_PUPIL_Z, BUNDLE_RADIUS)
    sweep("C3 built on the device", g, lambda: g.propagate(clip=True))
    del g
    s2 = ra.system_from_yaml(P.COOKE % dict(
        air="air", sk16="SCHOTT-SK|N-SK16", f2="SCHOTT-F|N-F2"))
    ls = [587.56e-9, 656.27e-9, 486.13e-9]
    y2, u2 = dc.bundle(10**6, 5.5, 5., 0)
    g2 = ra.GeometricTrace(s2, engine=Engine(0))
    g2.rays_given(y2, u2, l=ls)
    sweep("C2 3 x 10^6 rays", g2, lambda: g2.propagate(clip=True))
    del g2
    s4 = ra.system_from_yaml(P.ASPHERE_PHONE)
    y4, u4 = dc.bundle(n, .6, 10., 4)
    y4[:, 1] -= .5*np.tan(np.radians(10.))
    for label, opts in (("C4 default", {}), ("C4 exact", {"exact_asphere": 1})):
        g4 = ra.GeometricTrace(s4, engine=Engine(0), **opts)
        g4.rays_given(y4, u4, s4.wavelengths[0])
        sweep(label, g4, lambda: g4.propagate(clip=True))
        del g4


def cmd_c4check(a):
    """Why does bench.py's C4 record (shared engine, after C3 and C2) time
    slower than a fresh context?  The same flow, with a resident sweep."""
    ra, P, _build, Engine = _imports()
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import digest_cases as dc
    from bench import workload_rays, kernel_ms_of
    s4 = ra.system_from_yaml(P.ASPHERE_PHONE)
    y4, u4 = dc.bundle(10_000_000, .6, 10., 4)
    y4[:, 1] -= .5*np.tan(np.radians(10.))
    l4 = s4.wavelengths[0]

    def sweep(tag, g):
        eng = g.engine
        res = {"bench_way": kernel_ms_of(g, True)}
        for lds in (32768, 28672, 24576, 0):
            eng.set_option("resident_lds", lds)
            res[str(lds)] = steady(eng, .4)
        eng.set_option("resident_lds", -1)
        res["auto_steady"] = steady(eng, .4)
        res["bench_way_again"] = kernel_ms_of(g, True)
        out(tag=tag, placement=eng.placement(), ms=res)

    g = ra.GeometricTrace(s4, engine=Engine(0))
    g.rays_given(y4, u4, l4)
    sweep("fresh context", g)
    del g
    s3 = ra.system_from_yaml(P.DOUBLE_GAUSS)
    y, u = workload_rays(10_000_000, 0)
    g3 = ra.GeometricTrace(s3, device=0)
    g3.rays_given(y, u)
    g3.propagate(clip=True)
    del g3
    g = ra.GeometricTrace(s4, device=0)
    g.rays_given(y4, u4, l4)
    sweep("shared engine after C3 (10 pieces of 13 x 10^7)", g)
    small = ra.GeometricTrace(s4, device=0)
    small.rays_given(y4[:100_000], u4[:100_000], l4)
    small.propagate(clip=True)
    sweep("... after another trace used the engine (re-seeded)", g)


def cmd_nsweep(a):
    """Launch time and fraction of the HBM spec over the number of rays: C3
    shape (five field bundles built on the device, 16 B/ray read, and
    host-seeded up to 10^7) and C2 shape (three wavelength groups, one
    launch).  One fresh context per size: its own placement."""
    ra, P, _build, Engine = _imports()
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import digest_cases as dc
    from bench import workload_rays, FIELD_FRACTIONS, BUNDLE_RADIUS

    def measure(eng, propagate):
        propagate()
        for _ in range(10):
            propagate()
        res = {}
        for lds in (-1, 65536, 32768, 0):
            eng.set_option("resident_lds", lds)
            propagate()
            t_end = time.time() + .3
            ms = []
            while time.time() < t_end:
                eng.event_record(0)
                for _ in range(5):
                    propagate()
                eng.event_record(1)
                ms.append(eng.event_elapsed(0, 1)/5)
            res[str(lds)] = float(np.median(ms[len(ms)//3:]))
        eng.set_option("resident_lds", -1)
        return res

    s3 = ra.system_from_yaml(P.DOUBLE_GAUSS)
    nf = len(FIELD_FRACTIONS)
    for n in a.sizes:
        m = n//nf//64*64
        eng = Engine(0)
        g = ra.GeometricTrace(s3, engine=eng)
        g.rays_fields(np.c_[np.zeros(nf), FIELD_FRACTIONS],
                      dc.disc_points(m, 91), P.DOUBLE_GAUSS_PUPIL_Z,
                      BUNDLE_RADIUS)
        res = measure(eng, lambda: g.propagate(clip=True))
        alg = m*nf*(56*12 + 16)
        out(shape="C3 built on the device", rays=m*nf,
            blocks=eng.blocks(), placement=eng.placement(), ms=res,
            frac_auto=alg/(res["-1"]*1e-3)/8e12)
        if n <= 10_000_000:
            y, u = workload_rays(n, 0)
            g.rays_given(y, u)
            res = measure(eng, lambda: g.propagate(clip=True))
            uni, tiles = eng.input_uniform()
            rb = sum(8*(n - 64*q) + 8*q for q in uni) + 4*tiles
            out(shape="C3 host-seeded", rays=n, placement=eng.placement(),
                ms=res, frac_auto=(n*56*12 + rb)/(res["-1"]*1e-3)/8e12)
        del g
        eng.close()
    s2 = ra.system_from_yaml(P.COOKE % dict(
        air="air", sk16="SCHOTT-SK|N-SK16", f2="SCHOTT-F|N-F2"))
    ls = [587.56e-9, 656.27e-9, 486.13e-9]
    for n in a.sizes:
        if n > 30_000_000:
            continue
        per = n//3//64*64
        y2, u2 = dc.bundle(per, 5.5, 5., 0)
        eng = Engine(0)
        g2 = ra.GeometricTrace(s2, engine=eng)
        g2.rays_given(y2, u2, l=ls)
        res = measure(eng, lambda: g2.propagate(clip=True))
        uni, tiles = eng.input_uniform()
        rb = sum(8*(3*per - 64*q) + 8*q for q in uni) + 4*tiles
        out(shape="C2 three wavelength groups", rays=3*per,
            placement=eng.placement(), ms=res,
            frac_auto=(3*per*56*8 + rb)/(res["-1"]*1e-3)/8e12)
        del g2
        eng.close()


KINDS_NOTE = ("C3 host-seeded clip", "C3 image row only", "C2 3 x 10^6 rays",
              "C4 default", "C4 exact")


def cmd_kinds(a):
    """A fixed schedule of trace launches per kind of trace, for a
    `rocprofv3 --pmc` pass (counters per kind: `kinds-summary`).  Without a
    profiler it prints the launch times and the gfx clock per kind."""
    ra, P, _build, Engine = _imports()
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import digest_cases as dc
    from bench import workload_rays, Telemetry
    count = [0]
    tele = Telemetry(0, period=0.01) if a.telemetry else None
    recs = []

    def run(name, g, propagate):
        propagate()
        count[0] += 1
        for _ in range(a.warm):
            propagate()
        count[0] += a.warm
        first = count[0]
        if tele is not None:
            tele.mark("%d:begin" % len(recs))
        eng = g.engine
        t_end = time.time() + a.seconds
        ms = []
        while True:
            eng.event_record(0)
            for _ in range(a.launches):
                propagate()
            eng.event_record(1)
            ms.append(eng.event_elapsed(0, 1)/a.launches)
            count[0] += a.launches
            if time.time() >= t_end:
                break
        if tele is not None:
            tele.mark("%d:end" % len(recs))
        recs.append({"kind": name, "first": first, "count": count[0] - first,
                     "ms": float(np.median(ms)),
                     "placement_mixed": eng.placement()["fast"]})

    n = a.rays
    s3 = ra.system_from_yaml(P.DOUBLE_GAUSS)
    y, u = workload_rays(n, 0)
    g = ra.GeometricTrace(s3, engine=Engine(0))
    g.rays_given(y, u)
    run("C3 host-seeded clip", g, lambda: g.propagate(clip=True))
    run("C3 image row only", g, lambda: g.propagate(clip=True, keep=[0, -1]))
    del g
    s2 = ra.system_from_yaml(P.COOKE % dict(
        air="air", sk16="SCHOTT-SK|N-SK16", f2="SCHOTT-F|N-F2"))
    y2, u2 = dc.bundle(10**6, 5.5, 5., 0)
    g2 = ra.GeometricTrace(s2, engine=Engine(0))
    g2.rays_given(y2, u2, l=[587.56e-9, 656.27e-9, 486.13e-9])
    run("C2 3 x 10^6 rays", g2, lambda: g2.propagate(clip=True))
    del g2
    s4 = ra.system_from_yaml(P.ASPHERE_PHONE)
    y4, u4 = dc.bundle(n, .6, 10., 4)
    y4[:, 1] -= .5*np.tan(np.radians(10.))
    for label, opts in (("C4 default", {}), ("C4 exact", {"exact_asphere": 1})):
        g4 = ra.GeometricTrace(s4, engine=Engine(0), **opts)
        g4.rays_given(y4, u4, s4.wavelengths[0])
        run(label, g4, lambda: g4.propagate(clip=True))
        del g4
    t = tele.stop() if tele is not None else None
    for k, r in enumerate(recs):
        w = (t or {}).get(str(k)) or {}
        r["gfxclk_mhz"] = (w.get("gfxclk_mhz") or [None]*3)[1]
        r["socket_power_w"] = (w.get("socket_power_w") or [None]*3)[1]
        r["power_limited_fraction"] = w.get("power_limited_fraction")
        out(**r)


def cmd_sizes(a):
    """A fixed schedule of C3 traces (built on the device) per batch size,
    for `rocprofv3 --pmc` passes: what grows with N?  (`kinds-summary` reads
    the result; the kind is the size.)"""
    ra, P, _build, Engine = _imports()
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import digest_cases as dc
    from bench import FIELD_FRACTIONS, BUNDLE_RADIUS
    s3 = ra.system_from_yaml(P.DOUBLE_GAUSS)
    nf = len(FIELD_FRACTIONS)
    count = 0
    for n in a.sizes:
        if a.piece_mib:
            os.environ["RT_MI355_PIECE_MIB"] = str(a.piece_mib)
        m = n//nf//64*64
        eng = Engine(0)
        g = ra.GeometricTrace(s3, engine=eng)
        g.rays_fields(np.c_[np.zeros(nf), FIELD_FRACTIONS],
                      dc.disc_points(m, 91), P.DOUBLE_GAUSS_PUPIL_Z,
                      BUNDLE_RADIUS)
        for lds in a.caps:
            eng.set_option("resident_lds", lds)
            g.propagate(clip=True)
            for _ in range(a.warm):
                g.propagate(clip=True)
            count += 1 + a.warm
            if a.settle:        # (not under a profiler: a fixed schedule)
                steady(eng, a.settle)
            first = count
            ms = block_ms_fn(eng, lambda: g.propagate(clip=True), a.launches)
            count += a.launches
            out(kind="%d rays, resident_lds %d" % (m*nf, lds), rays=m*nf,
                resident_lds=lds, first=first, count=a.launches, ms=ms,
                per_1e7=ms*1e7/(m*nf), placement=eng.placement())
        del g
        eng.close()


def cmd_variants(a):
    """The laboratory kernel's variants again, now that the arrays lie in
    mixed memory (the balance between the store streams and the FP64 side has
    moved since they were rejected): rays per lane, non-temporal stores, XCD
    dealing, workgroup size -- all against the laboratory kernel's own
    default (48 B per ray read, no tile notes), same process, same arrays."""
    ra, P, _build, Engine = _imports()
    from bench import workload_rays
    n = a.rays
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    y, u = workload_rays(n, 0)
    eng = Engine(0, lib_path=lab_lib())
    g = ra.GeometricTrace(system, engine=eng)
    g.rays_given(y, u)
    g.propagate(clip=True)
    base = np.array(g.y[-1])
    steady(eng, 1.)
    defaults = dict(rays_per_thread=1, nontemporal=0, xcd_remap=0, block=256,
                    lds_pad=32768)
    variants = [{}, dict(lds_pad=40960), dict(lds_pad=49152),
                dict(rays_per_thread=2), dict(rays_per_thread=2, lds_pad=65536),
                dict(rays_per_thread=2, lds_pad=40960),
                dict(nontemporal=1 - a.nt), dict(xcd_remap=1),
                dict(block=128, lds_pad=16384), dict(block=512, lds_pad=65536),
                dict(rays_per_thread=4, lds_pad=65536),
                dict(rays_per_thread=2, xcd_remap=1), {}]
    defaults["nontemporal"] = a.nt
    for v in variants:
        opts = dict(defaults, **v)
        for k, val in opts.items():
            eng.set_option(k, val)
        g.propagate(clip=True)
        same = bool(np.array_equal(np.array(g.y[-1]), base, equal_nan=True))
        ms = [steady(eng, .5) for _ in range(2)]
        out(variant=v or "laboratory default", ms=ms, bit_identical=same,
            placement_mixed=eng.placement()["fast"])
    for k, val in defaults.items():
        eng.set_option(k, val)
    eng.set_option("lds_pad", 0)
    g.propagate(clip=True)
    out(variant="shipped kernel (tile notes, 16.6 B/ray read)",
        ms=[steady(eng, .5) for _ in range(2)])


def cmd_libab(a):
    """Same-process A/B of two BUILDS of the library (e.g. a -D variant):
    placed contexts of each, alternating steady blocks per kind of trace, the
    image rows compared, the consumers on the freshly traced rows timed."""
    ra, P, _build, Engine = _imports()
    from bench import workload_rays
    libs = {"A": a.lib_a or _build.LIB, "B": a.lib_b}
    n = a.rays
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    L = len(system)
    y, u = workload_rays(n, 0)
    if a.per_ray_directions:    # nothing but z = 0 uniform across a tile
        rng = np.random.default_rng(3)
        u[:, 0] += 1e-7*rng.standard_normal(n)
        u[:, 1] += 1e-7*rng.standard_normal(n)
        u[:, 2] = np.sqrt(1. - u[:, 0]**2 - u[:, 1]**2)
    traces = {}
    for tag, path in libs.items():
        g = ra.GeometricTrace(system, engine=Engine(0, lib_path=path))
        g.rays_given(y, u)
        g.propagate(clip=True)
        traces[tag] = g
    rows = {t: np.array(g.y[-1]) for t, g in traces.items()}
    same = bool(np.array_equal(rows["A"], rows["B"], equal_nan=True))
    for kind, kw in (("clip", dict(clip=True)), ("unclipped", dict(clip=False)),
                     ("image row only", dict(clip=True, keep=[0, -1]))):
        res = {"A": [], "B": []}
        for rep in range(a.reps):
            for tag in ("A", "B") if rep % 2 == 0 else ("B", "A"):
                g = traces[tag]
                g.propagate(**kw)
                t_end = time.time() + .6
                ms = []
                while time.time() < t_end:
                    g.engine.event_record(0)
                    for _ in range(10):
                        g.propagate(**kw)
                    g.engine.event_record(1)
                    ms.append(g.engine.event_elapsed(0, 1)/10)
                res[tag].append(float(np.median(ms[len(ms)//3:])))
        out(kind=kind, libs={t: os.path.basename(p) for t, p in libs.items()},
            image_rows_bit_identical=same,
            ms_A=float(np.median(res["A"])), ms_B=float(np.median(res["B"])),
            all_A=res["A"], all_B=res["B"],
            placement={t: traces[t].engine.placement()["per_class"]
                       for t in traces})
    # consumers right after a trace (are the rows still in the Infinity Cache?)
    for tag, g in traces.items():
        eng = g.engine
        cons = {}
        for name, fn in (("rms", lambda: g.rms()),
                         ("refocus_shift", lambda: eng.refocus_shift(L - 1)),
                         ("row_rmax", lambda: eng.row_rmax(L - 1))):
            t = []
            for _ in range(12):
                g.propagate(clip=True)
                eng.sync()
                t0 = time.perf_counter()
                fn()
                t.append((time.perf_counter() - t0)*1e3)
            cons[name] = float(np.median(t))
        out(consumers_first_call_after_a_trace_ms=cons, lib=tag)


def cmd_compact(a):
    """Where the compacting kernel (ballots + LDS exchange, opt-in) pays: an
    over-filled C3 batch (most rays vignetted at the first elements) traced
    for its image row only -- dead rays are wasted FP64 issue there -- and
    with every row stored (where it cannot pay)."""
    ra, P, _build, Engine = _imports()
    from rayopt_amd.bundles import multi_field_bundle
    from bench import FIELD_FRACTIONS
    n = a.rays
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    fields = [f*P.DOUBLE_GAUSS_FIELD_DEG for f in FIELD_FRACTIONS]
    for scale in a.scales:
        y, u = multi_field_bundle(n, 17.*scale, fields, seed=5,
                                  z_pupil=P.DOUBLE_GAUSS_PUPIL_Z)
        eng = Engine(0)
        g = ra.GeometricTrace(system, engine=eng)
        g.rays_given(y, u)
        g.propagate(clip=True)
        alive = float(np.isfinite(np.asarray(g.u[-1])[:, 0]).mean())
        ref = np.array(g.y[-1])
        for keep, label in (([0, -1], "image row only"), (None, "every row")):
            res = {}
            for compact in (0, 2):
                eng.set_option("compact", compact)
                g.propagate(clip=True, keep=keep)
                same = bool(np.array_equal(np.array(g.y[-1]), ref,
                                           equal_nan=True))
                res["compact=%d" % compact] = steady(eng, .5, clip=True) \
                    if keep is None else None
                if keep is not None:
                    t_end = time.time() + .5
                    ms = []
                    while time.time() < t_end:
                        eng.event_record(0)
                        for _ in range(10):
                            g.propagate(clip=True, keep=keep)
                        eng.event_record(1)
                        ms.append(eng.event_elapsed(0, 1)/10)
                    res["compact=%d" % compact] = float(np.median(ms))
                res["same_bits_%d" % compact] = same
            eng.set_option("compact", 0)
            out(bundle_radius_scale=scale, alive_at_image=alive, rows=label,
                ms=res)
        del g
        eng.close()


def cmd_noinput(a):
    """Upper bound of what hiding the launch-row reads could buy: the
    laboratory kernel with non-temporal stores, its six input components
    read per lane (48 B per ray) against "uniform_fix" masks that take them
    from the wavefront's first column through the scalar path (the results
    are wrong for those components; the arithmetic and the stores are the
    same)."""
    ra, P, _build, Engine = _imports()
    from bench import workload_rays
    n = a.rays
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    y, u = workload_rays(n, 0)
    eng = Engine(0, lib_path=lab_lib())
    g = ra.GeometricTrace(system, engine=eng)
    g.rays_given(y, u)
    g.propagate(clip=True)
    steady(eng, 1.)
    eng.set_option("nontemporal", 1)
    for lds in (32768, 65536):
        eng.set_option("lds_pad", lds)
        for mask, what in ((0, "48 B per ray read"),
                           (0b111100, "y0 y1 read (16 B per ray)"),
                           (0b111111, "nothing read per lane")):
            eng.set_option("uniform_fix", mask)
            g.propagate(clip=True)
            out(lds_pad=lds, uniform_fix=mask, what=what,
                ms=[steady(eng, .5) for _ in range(2)])
    eng.set_option("uniform_fix", 0)


def cmd_optab(a):
    """Same-context A/B of one engine option (same arrays, alternating steady
    blocks), for collimated bundles and bundles with per-ray directions; the
    rows compared bit for bit."""
    ra, P, _build, Engine = _imports()
    from bench import workload_rays
    n = a.rays
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    L = len(system)
    for shape in ("collimated", "per-ray directions"):
        y, u = workload_rays(n, 0)
        if shape != "collimated":
            rng = np.random.default_rng(3)
            u[:, 0] += 1e-7*rng.standard_normal(n)
            u[:, 1] += 1e-7*rng.standard_normal(n)
            u[:, 2] = np.sqrt(1. - u[:, 0]**2 - u[:, 1]**2)
        eng = Engine(0)
        g = ra.GeometricTrace(system, engine=eng)
        g.rays_given(y, u)
        g.propagate(clip=True)
        steady(eng, .6)
        for kw_name, kw in (("clip", dict(clip=True)),
                            ("unclipped", dict(clip=False))):
            res = {v: [] for v in a.values}
            rows = {}
            for rep in range(a.reps):
                for v in (a.values if rep % 2 == 0 else a.values[::-1]):
                    eng.set_option(a.option, v)
                    g.propagate(**kw)
                    if rep == 0:
                        rows[v] = [np.array(eng.download(w, 1, L))
                                   for w in (0, 1, 3)]
                    t_end = time.time() + .5
                    ms = []
                    while time.time() < t_end:
                        eng.event_record(0)
                        for _ in range(10):
                            g.propagate(**kw)
                        eng.event_record(1)
                        ms.append(eng.event_elapsed(0, 1)/10)
                    res[v].append(float(np.median(ms[len(ms)//3:])))
            same = all(np.array_equal(p, q, equal_nan=True)
                       for v in a.values[1:]
                       for p, q in zip(rows[a.values[0]], rows[v]))
            out(bundles=shape, trace=kw_name, option=a.option,
                ms={str(v): float(np.median(t)) for v, t in res.items()},
                all={str(v): t for v, t in res.items()}, bit_identical=same,
                placement=eng.placement()["per_class"])
        del g
        eng.close()


def cmd_floor(a):
    """The store floor under the trace: the trace kernel's own store pattern
    without arithmetic and without reads (rt_probe 8), ordinary and
    non-temporal stores, in placed arrays, per cap on the resident
    workgroups; next to it the trace itself and the FP64 side alone."""
    ra, P, _build, Engine = _imports()
    from bench import workload_rays
    n = a.rays
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    y, u = workload_rays(n, 0)
    eng = Engine(0, lib_path=lab_lib())
    g = ra.GeometricTrace(system, engine=eng)
    g.rays_given(y, u)
    g.propagate(clip=True)
    steady(eng, 1.)
    res = {"trace_ms": steady(eng, .6)}
    g.propagate(clip=True, keep=[0, -1])
    t_end = time.time() + .5
    ms = []
    while time.time() < t_end:
        eng.event_record(0)
        for _ in range(10):
            g.propagate(clip=True, keep=[0, -1])
        eng.event_record(1)
        ms.append(eng.event_elapsed(0, 1)/10)
    res["image_row_only_ms"] = float(np.median(ms))
    g.propagate(clip=True)
    for store in (0, 1):
        eng.set_option("probe_store", store)
        for lds in (65536, 32768, 0):
            eng.set_option("lds_pad", lds)
            pm = [eng.probe(8)[0] for _ in range(40)]
            res["pattern_%s_lds_%d" % ("nt" if store else "plain", lds)] = \
                float(np.median(pm[10:]))
    eng.set_option("lds_pad", 0)
    eng.set_option("probe_store", 0)
    out(placement=eng.placement()["per_class"], **res)


def cmd_spacing(a):
    """Is it the distance between the rows or the size of the batch that
    slows traces above 10^7 rays?  A batch of N rays traced whole, and in
    pieces of 10^7 rays (rt_trace_chunk: the same row spacing, a launch the
    size of the headline's)."""
    ra, P, _build, Engine = _imports()
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import digest_cases as dc
    from bench import FIELD_FRACTIONS, BUNDLE_RADIUS
    s3 = ra.system_from_yaml(P.DOUBLE_GAUSS)
    nf = len(FIELD_FRACTIONS)
    for n in a.sizes:
        m = n//nf//64*64
        eng = Engine(0)
        g = ra.GeometricTrace(s3, engine=eng)
        g.rays_fields(np.c_[np.zeros(nf), FIELD_FRACTIONS],
                      dc.disc_points(m, 91), P.DOUBLE_GAUSS_PUPIL_Z,
                      BUNDLE_RADIUS)
        g.propagate(clip=True)
        whole = steady(eng, .5)
        q = max(1, round(m*nf/10_000_000))
        L = len(s3)

        def pieces():
            for k in range(q):
                eng.trace_chunk(1, L, True, k, q)
        for _ in range(5):
            pieces()
        t_end = time.time() + .5
        ms = []
        while time.time() < t_end:
            eng.event_record(0)
            for _ in range(3):
                pieces()
            eng.event_record(1)
            ms.append(eng.event_elapsed(0, 1)/3)
        out(rays=m*nf, whole_ms=whole, per_1e7=whole*1e7/(m*nf), pieces=q,
            in_pieces_ms=float(np.median(ms)),
            in_pieces_per_1e7=float(np.median(ms))*1e7/(m*nf),
            placement=eng.placement())
        del g
        eng.close()


def block_ms_fn(eng, fn, launches):
    eng.event_record(0)
    for _ in range(launches):
        fn()
    eng.event_record(1)
    return eng.event_elapsed(0, 1)/launches


def cmd_superblock(a):
    """The knee above 1.1*10^7 rays: contexts below it traced in turn stay
    fast (`alternate`), a big batch traced in pieces does not (`spacing`) --
    so it is the address span the 84 concurrent row streams cover.  The
    laboratory's tile-major layout with tiles of 2^21..2^23 rays keeps that
    span per tile below 8 GiB whatever the batch: SoA against it, same
    laboratory kernel, same contexts."""
    ra, P, _build, Engine = _imports()
    from bench import workload_rays
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    lib = lab_lib()
    for n in a.sizes:
        y, u = workload_rays(n, 0)
        eng = Engine(0, lib_path=lib)
        g = ra.GeometricTrace(system, engine=eng)
        rec = {"rays": n, "row_pad_doubles": a.pad if a.planes else 0,
               "inside_a_tile": "[Y|U|I|T][element][3][TR] (SoA)"
               if a.planes else "[element][10 components][TR]",
               "kernel": "laboratory (48 B per ray read)"
               if a.lab else "the shipped rt_trace_kernel (tile notes, "
               "non-temporal stores) of the laboratory build"}
        for tile in [0] + a.tiles:
            eng.set_option("tile_planes", 1 if a.planes else 0)
            eng.set_option("tile_shipped_kernel", 0 if a.lab else 1)
            eng.set_option("tile_pad", a.pad if a.planes else 0)
            eng.set_option("tile_rays", 0)
            eng.set_option("tile_rays", tile)
            g.rays_given(y, u)
            g.propagate(clip=True)
            eng.set_option("lds_pad" if a.lab else "resident_lds", a.lds)
            steady(eng, .4)
            ms = steady(eng, .6)
            per_block = None
            if tile and a.per_block and n % tile == 0:
                # one launch per block (rt_trace_chunk on tile boundaries)
                q, L = n//tile, len(system)

                def blocks():
                    for k in range(q):
                        eng.trace_chunk(1, L, True, k, q)
                for _ in range(5):
                    blocks()
                t_end, mb = time.time() + .6, []
                while time.time() < t_end:
                    eng.event_record(0)
                    for _ in range(3):
                        blocks()
                    eng.event_record(1)
                    mb.append(eng.event_elapsed(0, 1)/3)
                per_block = float(np.median(mb))
            slots = -(-n//tile)*tile if tile else -(-n//64)*64
            eng.set_option("lds_pad" if a.lab else "resident_lds",
                           0 if a.lab else -1)
            rec["tile_%d" % tile] = {
                "ms": ms, "slots_traced": slots,
                "per_1e7_slots": ms*1e7/slots,
                "one_launch_per_block_per_1e7":
                    per_block*1e7/slots if per_block else None,
                "span_GiB_of_concurrent_streams":
                    (13*10*tile*8 if tile else 13*10*n*8)/2**30,
                "placement": eng.placement()["per_class"]}
        out(**rec)
        del g
        eng.close()


def cmd_planes(a):
    """Is it the ADDRESS SPAN of the concurrently written rows that makes
    the knee?  The planes in the order Y | U | T | I instead of Y | U | I | T
    (the i rows are served from u rows and never written: a 3 L ld gap
    inside the written span); same context, same rays, shipped kernel."""
    ra, P, _build, Engine = _imports()
    from bench import workload_rays
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    lib = lab_lib()
    for n in a.sizes:
        y, u = workload_rays(n, 0)
        eng = Engine(0, lib_path=lib)
        g = ra.GeometricTrace(system, engine=eng)
        rec = {"rays": n}
        for rep in range(2):
            for order in (0, 1):
                eng.set_option("t_before_i", order)
                g.rays_given(y, u)
                g.propagate(clip=True)
                steady(eng, .4)
                ms = steady(eng, .6)
                rec.setdefault("Y|U|T|I" if order else "Y|U|I|T", []).append(
                    ms*1e7/n)
        rec["placement"] = eng.placement()["per_class"]
        rec["resident"] = eng.placement()["fast"]
        out(**rec)
        del g
        eng.close()


def cmd_alternate(a):
    """Is the knee above 1.1*10^7 rays the reach of the address translation
    ACROSS launches?  K contexts of --rays rays each (every one below the
    knee) traced in turn A B C A B C ...: if a trace finds the translations
    of its arrays evicted by the traces in between, each runs like a slice
    of one big batch."""
    ra, P, _build, Engine = _imports()
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import digest_cases as dc
    from bench import FIELD_FRACTIONS, BUNDLE_RADIUS
    s3 = ra.system_from_yaml(P.DOUBLE_GAUSS)
    nf = len(FIELD_FRACTIONS)
    m = int(a.rays)//nf//64*64
    gs = []
    for k in range(a.contexts):
        g = ra.GeometricTrace(s3, engine=Engine(0))
        g.rays_fields(np.c_[np.zeros(nf), FIELD_FRACTIONS],
                      dc.disc_points(m, 91), P.DOUBLE_GAUSS_PUPIL_Z,
                      BUNDLE_RADIUS)
        g.propagate(clip=True)
        gs.append(g)
    steady(gs[0].engine, 1.)
    alone = [steady(g.engine, .5) for g in gs]
    for group in range(1, a.contexts + 1):
        use = gs[:group]
        for _ in range(20):
            for g in use:
                g.engine.trace(1, 0, True)
        ms = [[] for _ in use]
        for _ in range(40):
            for i, g in enumerate(use):
                g.engine.trace(1, 0, True)
                g.engine.sync()
                ms[i].append(g.engine.kernel_ms())
        out(rays_per_context=m*nf, contexts_in_turn=group,
            written_GiB_in_turn=group*m*nf*728/2**30,
            alone_ms=alone[:group],
            in_turn_ms=[float(np.median(x)) for x in ms],
            placement=[g.engine.placement()["per_class"] for g in use])


def cmd_consumers(a):
    """The reductions on the resident headline batch, 20 calls each -- for a
    `rocprofv3 --kernel-trace --stats` pass whose rows stand beside the
    `consumers` records of bench.py."""
    ra, P, _build, Engine = _imports()
    from bench import workload_rays, FIELD_FRACTIONS
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    n = int(a.rays)
    y, u = workload_rays(n, 0)
    g = ra.GeometricTrace(system)
    g.rays_given(y, u)
    g.propagate(clip=True)
    eng, L, nf = g.engine, len(system), len(FIELD_FRACTIONS)
    calls = {"rms": lambda: g.rms(),
             "refocus_shift": lambda: eng.refocus_shift(L - 1),
             "spot_stats": lambda: eng.spot_stats(L - 1, n//nf, nf),
             "row_rmax": lambda: eng.row_rmax(L - 1)}
    for name, fn in calls.items():
        fn()
        eng.sync()
        t0 = time.perf_counter()
        for _ in range(a.calls):
            fn()
        out(call=name, rays=n, calls=a.calls,
            wall_ms_per_call=(time.perf_counter() - t0)/a.calls*1e3)


def cmd_hostpath(a):
    """Where the host-side time of the calls that return big arrays goes:
    rows down to fresh / reused numpy arrays, one ray's column, the pieces
    of opd_rays."""
    ra, P, _build, Engine = _imports()
    from rayopt_amd._lib import RT_Y, RT_U, RT_T
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    n = int(a.rays)
    y, u = ra.bundles.disc_bundle(n, 17., 5., 1, P.DOUBLE_GAUSS_PUPIL_Z)
    g = ra.GeometricTrace(system)
    g.rays_given(y, u)
    g.propagate(clip=False)
    eng, L = g.engine, len(system)
    eng.sync()

    def wall(fn, reps=5):
        fn()
        t = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            t.append((time.perf_counter() - t0)*1e3)
        return float(np.median(t)), float(min(t))
    mb = 24*n/1e6
    keep = np.empty((1, 3, n))
    rec = {"rays": n, "row_MB": mb}
    rec["download_row_fresh_ms"] = wall(lambda: eng.download(RT_Y, L - 1, L))
    rec["download_row_reused_ms"] = wall(
        lambda: eng.download(RT_Y, L - 1, L, out=keep))
    rec["np_empty_and_fill_ms"] = wall(lambda: np.empty((1, 3, n)).fill(0.))
    rec["np_copy_of_resident_ms"] = wall(lambda: keep.copy())
    rec["download_ray_ms"] = wall(lambda: eng.download_ray(RT_T, 5), 20)
    rec["opd_rays_ms"] = wall(lambda: g.opd_rays(radius=100.))
    rec["opd_kernel_ms"] = eng.kernel_ms()
    rec["upload_rays_ms"] = wall(lambda: g.rays_given(y, u))
    # what a pool of pinned host buffers would give: allocation cost once,
    # then direct DMA into memory that is neither faulted nor staged
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    ptr = ctypes.c_void_p()
    nbytes = 24*n
    t0 = time.perf_counter()
    assert hip.hipHostMalloc(ctypes.byref(ptr), ctypes.c_size_t(nbytes), 0) == 0
    rec["hipHostMalloc_ms"] = (time.perf_counter() - t0)*1e3
    src = eng.device_ptr(RT_Y, L - 1)

    def direct():
        assert hip.hipMemcpy(ptr, ctypes.c_void_p(src),
                             ctypes.c_size_t(nbytes), 2) == 0
    rec["direct_D2H_into_pinned_ms"] = wall(direct)
    view = np.ctypeslib.as_array(
        (ctypes.c_double*(3*n)).from_address(ptr.value))
    rec["numpy_sum_over_pinned_ms"] = wall(lambda: view.sum())
    rec["numpy_sum_over_pageable_ms"] = wall(lambda: keep.sum())
    t0 = time.perf_counter()
    assert hip.hipHostFree(ptr) == 0
    rec["hipHostFree_ms"] = (time.perf_counter() - t0)*1e3
    out(**rec)


def cmd_kinds_summary(a):
    """Counters per kind from a `rocprofv3 --pmc` pass of `kinds`, joined with
    the launch times and clocks of a plain run of the same command:
    VALU issue = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x gfx clock x launch
    time), the ceiling of a kernel that is bound by FP64 issue."""
    plain = {}
    for l in open(a.plain):
        if l.startswith("{"):
            r = json.loads(l)
            plain[r["kind"]] = r
    table = collections.OrderedDict()
    for sched in sorted(glob.glob(os.path.join(a.dir, "pass*.jsonl"))):
        tag = os.path.basename(sched)[:-6]
        recs = [json.loads(l) for l in open(sched) if l.startswith("{")]
        files = glob.glob(os.path.join(a.dir, tag, "**",
                                       "*counter_collection.csv"),
                          recursive=True)
        rows = [r for f in files for r in csv.DictReader(open(f))
                if "rt_trace" in r["Kernel_Name"]]
        by_counter = collections.defaultdict(dict)
        for r in rows:
            by_counter[r["Counter_Name"]][int(r["Dispatch_Id"])] = \
                float(r["Counter_Value"])
        for name, d in by_counter.items():
            ids = sorted(d)
            for rec in recs:
                sel = ids[rec["first"]:rec["first"] + rec["count"]]
                if sel:
                    table.setdefault(rec["kind"], {})[name] = float(
                        np.mean([d[i] for i in sel]))
    for kind, c in table.items():
        p = plain.get(kind, {})
        rec = {"kind": kind, "counters_per_launch": c, "ms": p.get("ms"),
               "gfxclk_mhz": p.get("gfxclk_mhz"),
               "socket_power_w": p.get("socket_power_w"),
               "power_limited_fraction": p.get("power_limited_fraction")}
        if p.get("ms") and p.get("gfxclk_mhz") and "SQ_INSTS_VALU" in c:
            rec["valu_issue_frac"] = c["SQ_INSTS_VALU"]*4/(
                1024*p["gfxclk_mhz"]*1e6*p["ms"]*1e-3)
        if "SQ_ACTIVE_INST_VALU" in c and "SQ_BUSY_CYCLES" in c:
            rec["active_inst_valu_over_busy_cycles"] = \
                c["SQ_ACTIVE_INST_VALU"]/c["SQ_BUSY_CYCLES"]
        out(**rec)


def cmd_pmc_summary(a):
    """For each pass directory: per (context, setting) mean of every counter
    over the trace-kernel dispatches of that block."""
    table = collections.OrderedDict()
    for sched in sorted(glob.glob(os.path.join(a.dir, "pass*.jsonl"))):
        tag = os.path.basename(sched)[:-6]
        recs = [json.loads(l) for l in open(sched) if l.startswith("{")]
        files = glob.glob(os.path.join(a.dir, tag, "**",
                                       "*counter_collection.csv"),
                          recursive=True)
        if not files or not recs:
            continue
        rows = [r for f in files for r in csv.DictReader(open(f))
                if "rt_trace_kernel" in r["Kernel_Name"]]
        by_counter = collections.defaultdict(dict)
        for r in rows:
            by_counter[r["Counter_Name"]][int(r["Dispatch_Id"])] = \
                float(r["Counter_Value"])
        for name, d in by_counter.items():
            ids = sorted(d)
            for rec in recs:
                for b in rec["blocks"]:
                    sel = ids[b["first"]:b["first"] + b["count"]]
                    if not sel:
                        continue
                    key = (tag, rec["context"], rec["kind"],
                           b["resident_lds"], b["first"])
                    table.setdefault(key, {"ms_under_pmc": b["ms"]})[name] = \
                        float(np.mean([d[i] for i in sel]))
    for key, vals in table.items():
        out(**{"pass": key[0], "context": key[1], "kind": key[2],
               "resident_lds": key[3], **vals})


def main():
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    p = sub.add_parser("placement")
    p.add_argument("--contexts", type=int, default=5)
    p.add_argument("--vmm", type=int, default=3)
    p.add_argument("--launches", type=int, default=24)
    p.add_argument("--warm", type=int, default=12)
    p.add_argument("--rays", type=int, default=10_000_000)
    p.add_argument("--probe", type=int, default=1)
    p.add_argument("--serial", type=int, default=0)
    p.add_argument("--vmm-mb", type=int, default=1024)
    p.set_defaults(fn=cmd_placement)
    p = sub.add_parser("layouts")
    p.add_argument("--contexts", type=int, default=6)
    p.add_argument("--vmm", type=int, default=3)
    p.add_argument("--rays", type=int, default=10_000_000)
    p.add_argument("--tiles", type=int, nargs="*", default=[64, 256, 2048])
    p.set_defaults(fn=cmd_layouts)
    p = sub.add_parser("placed")
    p.add_argument("--contexts", type=int, default=6)
    p.add_argument("--rays", type=int, default=10_000_000)
    p.add_argument("--all-placed", type=int, default=0)
    p.set_defaults(fn=cmd_placed)
    p = sub.add_parser("resident")
    p.add_argument("--rays", type=int, default=10_000_000)
    p.set_defaults(fn=cmd_resident)
    p = sub.add_parser("nsweep")
    p.add_argument("--sizes", type=int, nargs="*", default=[
        100_000, 300_000, 1_000_000, 2_000_000, 3_000_000, 5_000_000,
        10_000_000, 20_000_000, 50_000_000, 100_000_000])
    p.set_defaults(fn=cmd_nsweep)
    p = sub.add_parser("kinds")
    p.add_argument("--rays", type=int, default=10_000_000)
    p.add_argument("--launches", type=int, default=8)
    p.add_argument("--warm", type=int, default=8)
    p.add_argument("--seconds", type=float, default=0.)
    p.add_argument("--telemetry", type=int, default=0)
    p.set_defaults(fn=cmd_kinds)
    p = sub.add_parser("variants")
    p.add_argument("--rays", type=int, default=10_000_000)
    p.add_argument("--nt", type=int, default=1,
                   help="the baseline's row stores: 1 non-temporal")
    p.set_defaults(fn=cmd_variants)
    p = sub.add_parser("libab")
    p.add_argument("lib_b")
    p.add_argument("--lib-a", default=None)
    p.add_argument("--rays", type=int, default=10_000_000)
    p.add_argument("--reps", type=int, default=4)
    p.add_argument("--per-ray-directions", type=int, default=0)
    p.set_defaults(fn=cmd_libab)
    p = sub.add_parser("compact")
    p.add_argument("--rays", type=int, default=10_000_000)
    p.add_argument("--scales", type=float, nargs="*", default=[1., 1.6, 2.5])
    p.set_defaults(fn=cmd_compact)
    p = sub.add_parser("noinput")
    p.add_argument("--rays", type=int, default=10_000_000)
    p.set_defaults(fn=cmd_noinput)
    p = sub.add_parser("optab")
    p.add_argument("option")
    p.add_argument("values", type=int, nargs="+")
    p.add_argument("--rays", type=int, default=10_000_000)
    p.add_argument("--reps", type=int, default=4)
    p.set_defaults(fn=cmd_optab)
    p = sub.add_parser("floor")
    p.add_argument("--rays", type=int, default=10_000_000)
    p.set_defaults(fn=cmd_floor)
    p = sub.add_parser("spacing")
    p.add_argument("--sizes", type=int, nargs="*",
                   default=[10_000_000, 20_000_000, 50_000_000])
    p.set_defaults(fn=cmd_spacing)
    p = sub.add_parser("sizes")
    p.add_argument("--sizes", type=int, nargs="*",
                   default=[10_000_000, 20_000_000, 50_000_000])
    p.add_argument("--caps", type=int, nargs="*", default=[32768, 65536])
    p.add_argument("--launches", type=int, default=6)
    p.add_argument("--warm", type=int, default=4)
    p.add_argument("--piece-mib", type=int, default=0)
    p.add_argument("--settle", type=float, default=0.,
                   help="seconds of untimed launches before the timed block "
                        "(a context's first launches run at idle clocks: "
                        "without it the times are NOT comparable -- session "
                        "13 / 14 fell for that)")
    p.set_defaults(fn=cmd_sizes)
    p = sub.add_parser("kinds-summary")
    p.add_argument("dir")
    p.add_argument("plain")
    p.set_defaults(fn=cmd_kinds_summary)
    p = sub.add_parser("c4check")
    p.set_defaults(fn=cmd_c4check)
    p = sub.add_parser("superblock")
    p.add_argument("--sizes", type=lambda v: int(float(v)), nargs="+",
                   default=[10_000_000, 20_000_000])
    p.add_argument("--tiles", type=int, nargs="+",
                   default=[1 << 21, 1 << 22, 1 << 23])
    p.add_argument("--lds", type=int, default=32768)
    p.add_argument("--lab", action="store_true",
                   help="the laboratory kernel instead of the shipped one")
    p.add_argument("--planes", action="store_true",
                   help="inside a tile the planes of SoA (super-blocked SoA)")
    p.add_argument("--per-block", action="store_true",
                   help="also: one launch per block (whole-tile batches)")
    p.add_argument("--pad", type=int, default=0,
                   help="with --planes: doubles between the rows of a tile "
                        "beyond its rays (rows not a power of two apart)")
    p.set_defaults(fn=cmd_superblock)
    p = sub.add_parser("planes")
    p.add_argument("--sizes", type=lambda v: int(float(v)), nargs="+",
                   default=[10_000_000, 12_500_000, 15_000_000, 20_000_000])
    p.set_defaults(fn=cmd_planes)
    p = sub.add_parser("alternate")
    p.add_argument("--rays", type=float, default=5e6)
    p.add_argument("--contexts", type=int, default=4)
    p.set_defaults(fn=cmd_alternate)
    p = sub.add_parser("consumers")
    p.add_argument("--rays", type=float, default=1e7)
    p.add_argument("--calls", type=int, default=20)
    p.set_defaults(fn=cmd_consumers)
    p = sub.add_parser("hostpath")
    p.add_argument("--rays", type=float, default=1e7)
    p.set_defaults(fn=cmd_hostpath)
    p = sub.add_parser("pmc-summary")
    p.add_argument("dir")
    p.set_defaults(fn=cmd_pmc_summary)
    a = ap.parse_args()
    a.fn(a)


if __name__ == "__main__":
    main()
