"""The legs of bench.py besides the contract's timed loop: CPU baselines (the
reference itself and the ports, test infrastructure used as a yardstick),
telemetry, the counter passes (HBM traffic, VALU instructions) of a child run
under rocprofv3, one record per BASELINE config, the device-side consumers,
the host path (PCIe-inclusive) and the N > 1 configs[4] leg.  bench.py holds
the contract; everything here is reported beside it and never fatal."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
BENCH = os.path.join(ROOT, "bench.py")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# seeded, trigonometry-free bundle builders shared with the digest tests
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

HBM_PEAK_GBS = 8000.        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_ACHIEVABLE_GBS = 6290.  # same guide: measured float4 copy
HBM_NOMINAL_MHZ = 2000.     # uclk at which the 8 TB/s figure holds (amdsmi:
                            # MEM clock min = max = 2000 MHz on MI355X)
FIELD_FRACTIONS = (0, .35, .5, .7, 1.)
BUNDLE_RADIUS = 17.


def log(*a):
    print(*a, file=sys.stderr, flush=True)


_T0 = time.perf_counter()
LAPS = []       # (what just finished, seconds since the interpreter got here)


def lap(label):
    """Where the wall time of this command goes (stderr + `wall_s`)."""
    t = time.perf_counter() - _T0
    LAPS.append((label, round(t, 2)))
    log("[bench %6.2f s] %s" % (t, label))


def workload_rays(n, rank):
    from rayopt_amd import prescriptions as P
    from rayopt_amd.bundles import multi_field_bundle
    fields = [f*P.DOUBLE_GAUSS_FIELD_DEG for f in FIELD_FRACTIONS]
    return multi_field_bundle(n, BUNDLE_RADIUS, fields, seed=1000*rank,
                              z_pupil=P.DOUBLE_GAUSS_PUPIL_Z)


class Job:
    """One rank's share of the benchmark: its trace, its engine and the host
    group it synchronises with."""

    exchange = True
    chunks = 1

    def __init__(self, args, group, g, counts, d_dst):
        self.args, self.group, self.g = args, group, g
        self.eng = g.engine
        self.dist = group is not None
        self.counts, self.d_dst = counts, d_dst
        self.L = len(g.system)

    def gather(self):
        from rayopt_amd._lib import RT_Y
        if self.exchange:
            self.eng.gather_final(RT_Y, self.L - 1, self.counts, 0,
                                  self.d_dst)

    def gather_chunk(self, k, chunks):
        from rayopt_amd._lib import RT_Y
        if self.exchange:
            self.eng.gather_chunk(RT_Y, self.L - 1, self.counts, 0,
                                  self.d_dst, k, chunks)

    def fence(self):
        self.eng.sync()
        if self.dist:
            if self.exchange:
                self.eng.comm_sync()
            self.group.barrier()

    def timed(self, step, steps, warmup, final_gather, last_step=None):
        """W untimed + exactly K timed calls of `step`, bracketed by device
        sync + barrier on both sides.  Returns (wall s, HIP-event ms over the
        K steps on the trace stream, ms of the last kernel).  With
        ``final_gather`` the job's one exchange follows the last step inside
        the timed region; ``last_step`` (if given) IS the K-th step, traced
        in chunks whose gathers overlap the following chunks."""
        eng = self.eng
        for _ in range(warmup):
            step()
        self.fence()
        t0 = time.perf_counter()
        eng.event_record(0)
        chunked = final_gather and last_step is not None
        for _ in range(steps - 1 if chunked else steps):
            step()
        if chunked:
            last_step()         # K-th step + the exchange, pipelined
        eng.event_record(1)
        if final_gather and not chunked:
            self.gather()       # the job's one exchange
        self.fence()
        return (time.perf_counter() - t0, eng.event_elapsed(0, 1),
                eng.kernel_ms())


# --------------------------------------------------------------------------
# CPU baselines (N = 1, rank 0 only; test infrastructure used as a yardstick)
# --------------------------------------------------------------------------

_SHARD = {}


def _shard_worker(k):
    from oracle import trace_numpy as tn
    table, y, u, clip, bounds = (_SHARD[key] for key in
                                 ("table", "y", "u", "clip", "bounds"))
    lo, hi = bounds[k]
    Y, U, I, T = tn.propagate(table, y[lo:hi], u[lo:hi], clip=clip)
    return float(np.nansum(Y[-1]))      # touch the result


def cpu_port_on_processes(system, y, u, clip, procs):
    """The numpy port on `procs` forked processes over contiguous shards of
    the whole batch (must run before this process touches the GPU)."""
    import multiprocessing as mp
    from rayopt_amd.pack import pack_system
    from rayopt_amd.distributed import shard_bounds
    l = system.wavelengths[0]
    table, _ = pack_system(system, l, system.refractive_index(l, 0))
    _SHARD.update(table=table, y=y, u=u, clip=clip,
                  bounds=shard_bounds(len(y), procs))
    ctx = mp.get_context("fork")
    with ctx.Pool(procs) as pool:
        pool.map(_shard_worker, range(procs))          # warm the workers
        t0 = time.perf_counter()
        pool.map(_shard_worker, range(procs))
        dt = time.perf_counter() - t0
    _SHARD.clear()
    S = len(system) - 1
    return {"value": len(y)*S/dt, "unit": "ray-surface-ops/s",
            "cores": procs, "kind": "port",
            "sample": "the whole %d-ray batch on %d forked processes (one "
                      "per host core), contiguous shards, one propagate() of "
                      "the numpy port each (%.2f s)" % (len(y), procs, dt)}


def host_cpu():
    """'model name, N logical cores' of this host."""
    model = "unknown CPU"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return "%s, %d logical cores" % (model, os.cpu_count() or 0)


def cpu_one_core(table, system, y, u, clip, S, g, L, sample, l):
    """One propagate() of rayopt itself (oracle/_ref) -- and of the numpy
    port beside it -- on one core of this host; doubles as a parity check of
    the bench run itself: the image row the GPU computed in the timed loop
    against the reference's, bit for bit."""
    from oracle import trace_numpy as tn
    from oracle import refshim
    from rayopt_amd import prescriptions as P
    m = min(sample, y.shape[0])
    mp = min(m, 1_000_000)          # the port: a bounded slice of the sample
    ys, us = y[:m], u[:m]
    tn.propagate(table, ys[:100000], us[:100000], clip=clip)   # warm
    t0 = time.perf_counter()
    Y, U, I, T = tn.propagate(table, ys[:mp], us[:mp], clip=clip)
    dt = time.perf_counter() - t0
    got = np.asarray(g.y[L - 1])[:m]
    ref = Y[-1]
    assert np.array_equal(np.isnan(got[:mp]), np.isnan(ref))
    fin = np.isfinite(ref)
    assert (np.abs(got[:mp][fin] - ref[fin]) <=
            1e-10*np.maximum(np.abs(ref[fin]), 1.)).all()
    port = {
        "value": mp*S/dt, "unit": "ray-surface-ops/s", "cores": 1,
        "kind": "port",
        "sample": "first %d rays of the same workload, one propagate() of "
                  "the numpy port (%.1f s); host has %d cores" % (
                      mp, dt, os.cpu_count()),
        "host": host_cpu(),
        "image_row_bit_identical_to_gpu": bool(
            np.array_equal(got[:mp], ref, equal_nan=True)),
    }
    del Y, U, I, T
    if not refshim.available():     # no archive travelled: the port stands in
        port["note"] = ("oracle/_ref is missing (python -m oracle.make_ref "
                        "in the build container): the numpy port stands in "
                        "for the reference")
        return port
    out, t = reference_one_process(P.DOUBLE_GAUSS, ys, us, l, clip, m)
    image = t.y[-1]
    # the reference's own reduction on its result (rayopt/geometric_trace.py:
    # 171-183), for the `consumers` record
    t0 = time.perf_counter()
    with np.errstate(all="ignore"):
        ref_rms = float(t.rms())
    out["reference_rms_seconds"] = time.perf_counter() - t0
    out["reference_rms"] = ref_rms
    assert np.array_equal(np.isnan(got), np.isnan(image))
    out["host"] = host_cpu()
    out["image_row_bit_identical_to_gpu"] = bool(
        np.array_equal(got, image, equal_nan=True))
    out["port_value"] = port["value"]
    out["port_sample"] = port["sample"]
    return out


def cpu_c_oracle(table, y, u, clip, S, g, L, sample=2_000_000,
                 every_team=False):
    """The independent plain-C oracle (oracle/trace_c.c, OpenMP over rays):
    what a compiled multi-threaded CPU implementation of the same path
    reaches on this box, per team size.  Doubles as a second parity check."""
    from oracle import build_c
    build_c.build()
    m = min(sample, y.shape[0])
    ys, us = np.ascontiguousarray(y[:m]), np.ascontiguousarray(u[:m])
    build_c.propagate(table, ys[:100000], us[:100000], clip=clip)      # warm
    import ctypes
    try:
        gomp = ctypes.CDLL("libgomp.so.1")
    except OSError:
        gomp = None
    teams = sorted({min(os.cpu_count(), t) for t in (
        (16, 64, os.cpu_count()) if every_team else (64,))})
    out = build_c.propagate(table, ys, us, clip=clip)   # touch output pages
    by_team = {}
    for team in (teams if gomp is not None else teams[-1:]):
        if gomp is not None:
            gomp.omp_set_num_threads(team)
        best = None
        for _ in range(3 if every_team else 2):
            t0 = time.perf_counter()
            out = build_c.propagate(table, ys, us, clip=clip, out=out)
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        by_team[team] = m*S/best
    Y = out[0]
    got = np.asarray(g.y[L - 1])[:m]
    same = np.array_equal(got, Y[-1], equal_nan=True)
    cores = max(by_team, key=by_team.get)
    return {
        "value": by_team[cores],
        "range": [min(by_team.values()), max(by_team.values())],
        "by_team_size": {str(k): v for k, v in by_team.items()},
        "unit": "ray-surface-ops/s",
        "cores": cores,
        "kind": "port",
        "sample": "first %d rays, best of %d propagate() of the C port with "
                  "OpenMP (team sizes %s; --extras: 16 / 64 / all), same "
                  "output arrays; host has %d cores; boxes of the pool "
                  "differ by up to x1.8 on this figure" % (
                      m, 3 if every_team else 2, teams, os.cpu_count()),
        "image_row_bit_identical_to_gpu": bool(same),
    }


# --------------------------------------------------------------------------
# telemetry: clocks / power / temperature around the timed loop
# --------------------------------------------------------------------------

def telemetry_child(device, period):
    """Body of the sampling child (``bench.py --telemetry-child``): amdsmi
    metrics every ``period`` s until "stop" arrives on stdin; "mark <label>"
    lines stamp the sample stream.  A process of its own, so that sampling
    never competes with the launch loop for the interpreter."""
    import select
    out = {"samples": [], "marks": [], "error": None}
    try:
        import amdsmi
        amdsmi.amdsmi_init()
        handles = amdsmi.amdsmi_get_processor_handles()
        h = handles[device if device < len(handles) else 0]
        out["handles"] = len(handles)
    except Exception as err:
        out["error"] = repr(err)[:200]
        h = None
    sys.stdout.write("ready\n")
    sys.stdout.flush()

    def num(v):
        return float(v) if isinstance(v, (int, float)) else None
    running, pending = True, b""
    while running:
        r, _, _ = select.select([0], [], [], period)
        if r:
            # raw reads: lines that arrive together must not hide in a
            # buffered reader where select() cannot see them
            chunk = os.read(0, 65536)
            if not chunk:
                running = False
            pending += chunk
            while b"\n" in pending:
                line, pending = pending.split(b"\n", 1)
                line = line.decode().strip()
                if line == "stop":
                    running = False
                elif line.startswith("mark "):
                    out["marks"].append((line[5:].strip(), time.time()))
        if h is None:
            continue
        try:
            m = amdsmi.amdsmi_get_gpu_metrics_info(h)
            gfx = [num(v) for v in (m.get("current_gfxclks") or [])]
            gfx = [v for v in gfx if v]
            out["samples"].append((
                time.time(),
                sum(gfx)/len(gfx) if gfx else num(m.get("current_gfxclk")),
                num(m.get("current_uclk")),
                num(m.get("current_socket_power")),
                num(m.get("temperature_hotspot")),
                num(m.get("temperature_mem")),
                num(m.get("average_gfx_activity")),
                num(m.get("ppt_residency_acc")),
                num(m.get("socket_thm_residency_acc")),
                num(m.get("hbm_thm_residency_acc")),
                num(m.get("prochot_residency_acc")),
                num(m.get("accumulation_counter"))))
        except Exception as err:
            out["error"] = repr(err)[:200]
            h = None
    sys.stdout.write(json.dumps(out) + "\n")
    sys.stdout.flush()


class Telemetry:
    """Parent side of the sampling child."""
    FIELDS = ("gfxclk_mhz", "hbm_uclk_mhz", "socket_power_w", "hotspot_c",
              "hbm_c", "gfx_activity_pct", "ppt_residency_acc",
              "socket_thm_residency_acc", "hbm_thm_residency_acc",
              "prochot_residency_acc", "accumulation_counter")
    COUNTERS = FIELDS[6:]   # running totals: reported as the window's gain

    def __init__(self, device=0, period=0.004):
        import subprocess
        self.proc = None
        try:
            self.proc = subprocess.Popen(
                [sys.executable, BENCH, "--telemetry-child", str(device),
                 str(period)],
                stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                stderr=subprocess.DEVNULL, text=True)
            if self.proc.stdout.readline().strip() != "ready":
                raise RuntimeError("telemetry child did not start")
        except Exception as err:
            log("[bench] telemetry unavailable: %r" % (err,))
            self.proc = None

    def mark(self, label):
        if self.proc is not None:
            try:
                self.proc.stdin.write("mark %s\n" % label)
                self.proc.stdin.flush()
            except OSError:
                self.proc = None

    def stop(self, raw=False):
        """{window: {field: [min, mean, max]}} for the windows between marks
        "<name>:begin" and "<name>:end", plus the first and last sample
        (``raw``: also the sample rows themselves)."""
        if self.proc is None:
            return None
        try:
            self.proc.stdin.write("stop\n")
            self.proc.stdin.flush()
            data = json.loads(self.proc.stdout.readline())
            self.proc.wait(timeout=10)
        except Exception as err:
            return {"error": repr(err)[:200]}
        samples, marks = data["samples"], dict(
            (k, t) for k, t in data["marks"])

        def window(t0, t1):
            rows = [r for r in samples if t0 <= r[0] <= t1]
            out = {"samples": len(rows), "seconds": t1 - t0}
            # the window's neighbours bracket it: counters are differenced
            # across them, and a window shorter than the sampling period
            # still gets the state it ran in
            before = [r for r in samples if r[0] < t0][-1:]
            after = [r for r in samples if r[0] > t1][:1]
            rows = before + rows + after
            for k, name in enumerate(self.FIELDS, 1):
                v = [r[k] for r in rows if r[k] is not None]
                if not v:
                    continue
                if name in self.COUNTERS:
                    out[name + "_gain"] = v[-1] - v[0]
                else:
                    out[name] = [min(v), sum(v)/len(v), max(v)]
            if "ppt_residency_acc_gain" in out and \
                    out.get("accumulation_counter_gain"):
                # share of the window the power limiter was active
                out["power_limited_fraction"] = \
                    out["ppt_residency_acc_gain"] / \
                    out["accumulation_counter_gain"]
            return out
        out = {"source": "amdsmi_get_gpu_metrics_info in a child process",
               "error": data.get("error"),
               "samples": len(samples)}
        if raw:
            out["rows"] = samples
            out["fields"] = ("t",) + self.FIELDS
        if samples:
            out["first_sample"] = dict(zip(self.FIELDS, samples[0][1:]))
            out["last_sample"] = dict(zip(self.FIELDS, samples[-1][1:]))
        for name in sorted({k.split(":")[0] for k in marks}):
            if name + ":begin" in marks and name + ":end" in marks:
                out[name] = window(marks[name + ":begin"],
                                   marks[name + ":end"])
        return out


# --------------------------------------------------------------------------
# the reference itself on this host (oracle/_ref, test infrastructure)
# --------------------------------------------------------------------------

_REF = {}


def _ref_shard_worker(k):
    ro, text, y, u, l, clip, bounds = (_REF[key] for key in (
        "ro", "text", "y", "u", "l", "clip", "bounds"))
    lo, hi = bounds[k]
    system = ro.system_from_yaml(text)
    t = ro.GeometricTrace(system)
    t.rays_given(y[lo:hi], u[lo:hi], l)
    with np.errstate(all="ignore"):
        t.propagate(clip=clip)
    return float(np.nansum(t.y[-1]))


def reference_on_processes(text, y, u, l, clip, procs):
    """rayopt's own propagate() on ``procs`` forked processes over contiguous
    shards of the batch (forks: before this process opens the GPU)."""
    import multiprocessing as mp
    import warnings
    from oracle import refshim
    from rayopt_amd.distributed import shard_bounds
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ro = refshim.load()
    _REF.update(ro=ro, text=text, y=y, u=u, l=l, clip=clip,
                bounds=shard_bounds(len(y), procs))
    ctx = mp.get_context("fork")
    with ctx.Pool(procs) as pool:
        pool.map(_ref_shard_worker, range(procs))          # warm
        t0 = time.perf_counter()
        pool.map(_ref_shard_worker, range(procs))
        dt = time.perf_counter() - t0
    S = len(ro.system_from_yaml(text)) - 1
    _REF.clear()
    return {"value": len(y)*S/dt, "unit": "ray-surface-ops/s",
            "cores": procs, "kind": "reference",
            "sample": "the whole %d-ray batch on %d forked processes (one "
                      "per host core), contiguous shards, one rayopt."
                      "GeometricTrace.propagate() each (%.2f s)" % (
                          len(y), procs, dt)}


def reference_one_process(text, y, u, l, clip, sample):
    """One rayopt.GeometricTrace.propagate() of the first ``sample`` rays,
    one process.  Returns (record, trace) -- the trace for parity checks."""
    import warnings
    from oracle import refshim
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ro = refshim.load()
        system = ro.system_from_yaml(text)
    m = min(sample, len(y))
    warm = ro.GeometricTrace(system)
    warm.rays_given(y[:max(1, m//20)], u[:max(1, m//20)], l)
    with np.errstate(all="ignore"):
        warm.propagate(clip=clip)
    t = ro.GeometricTrace(system)
    t.rays_given(y[:m], u[:m], l)
    t0 = time.perf_counter()
    with np.errstate(all="ignore"):
        t.propagate(clip=clip)
    dt = time.perf_counter() - t0
    S = len(system) - 1
    return {"value": m*S/dt, "unit": "ray-surface-ops/s", "cores": 1,
            "kind": "reference", "rays": m, "seconds": dt,
            "sample": "first %d rays of the workload, one rayopt."
                      "GeometricTrace.propagate() (rayopt/geometric_trace.py"
                      ":72-80, unmodified, imported from %s), one process "
                      "(%.1f s); host has %d cores" % (
                          m, "oracle/_ref" if refshim.carried() else
                          refshim.REFERENCE_ROOT, dt, os.cpu_count())}, t


def traffic_from_profile():
    """HBM bytes per launch from the committed PMC profile, if one exists
    for this workload (profiles/traffic.json: roofline.traffic_detail of a
    committed bench line whose counter passes ran); otherwise null."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def profiled_from_outside(env):
    """True when this process already runs under a profiler (rocprofv3 / the
    rocprofiler-sdk tool library): a counter session nested inside another
    one is not attempted."""
    keys = ("ROCP_TOOL_LIBRARIES", "ROCPROF_OUTPUT_PATH",
            "ROCPROFILER_REGISTER_FORCE_LOAD", "ROCPROF_KERNEL_TRACE")
    return any(env.get(k) for k in keys) or \
        "rocprofiler-sdk-tool" in env.get("LD_PRELOAD", "")


# what the child run launches, in this order, per kernel name: the parent
# maps the counter rows (sorted by dispatch) back onto it
PMC_SCHEDULE = {
    "rt_trace_kernel": [("headline", 4), ("image_row_only", 3),
                        ("C4 default", 3), ("C4 exact", 3)],
    "rt_trace_gen_kernel": [("generated", 4)],
}
PMC_C4_RAYS = 2_000_000     # instructions per ray are what is counted


def pmc_child(n, clip):
    """``bench.py --pmc-child n clip``: the workloads whose counters the line
    reports, a few launches each and nothing else (run under rocprofv3 by
    :func:`pmc_live`): the headline batch (host-seeded, five collimated
    bundles), the same with only the image row kept, the same bundles built
    on the device, C4 on both arithmetics (2*10^6 rays: the counts per ray
    are what the line uses)."""
    import rayopt_amd as ra
    from rayopt_amd import prescriptions as P
    import digest_cases as dc
    n, clip = int(n), bool(int(clip))
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    y, u = workload_rays(n, 0)
    g = ra.GeometricTrace(system, device=0)
    g.rays_given(y, u)
    for _ in range(4):
        g.propagate(clip=clip)
    for _ in range(3):
        g.propagate(clip=clip, keep=[0, -1])
    g.engine.sync()
    del g
    nf = len(FIELD_FRACTIONS)
    h = ra.GeometricTrace(system, device=0)
    h.rays_fields(np.c_[np.zeros(nf), FIELD_FRACTIONS],
                  dc.disc_points(n//nf//64*64, 77), P.DOUBLE_GAUSS_PUPIL_Z,
                  BUNDLE_RADIUS)
    for _ in range(4):
        h.propagate(clip=clip)
    h.engine.sync()
    del h
    s4 = ra.system_from_yaml(P.ASPHERE_PHONE)
    y, u = dc.bundle(PMC_C4_RAYS, .6, 10., 4)
    y[:, 1] -= .5*np.tan(np.radians(10.))
    for opts in ({}, {"exact_asphere": 1}):
        k = ra.GeometricTrace(s4, device=0, **opts)
        k.rays_given(y, u, s4.wavelengths[0])
        for _ in range(3):
            k.propagate(clip=True)
        k.engine.sync()
        del k


def pmc_live(n, clip, timeout=120.):
    """Counters of THIS box, now: `rocprofv3 --kernel-trace --pmc ...` passes
    of :func:`pmc_child` -- FETCH_SIZE and WRITE_SIZE in separate passes
    (they do not fit one: MI355X_MICROARCH.md, TCC budget), SQ_INSTS_VALU +
    SQ_ACTIVE_INST_VALU riding with WRITE_SIZE where the tool takes them
    together, else in a third pass.  Returns {label: {counter: mean per
    launch}} with the first launch of every label left out (it carries what
    only a first launch does: row 0 of a generated batch, a table upload),
    or raises."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        raise RuntimeError("rocprofv3 not found")
    env = {k: v for k, v in os.environ.items()
           if not k.startswith(("ROCP", "ROCPROF"))}
    # (plain allocations in the child: what is counted -- bytes, instructions
    # -- does not depend on where the arrays live, and the placement's pair
    # tests are hundreds of launches a counter pass serialises)
    env.update(TMPDIR="/tmp", RT_BENCH_CHILD="1", RT_MI355_PLACEMENT="0")
    work = tempfile.mkdtemp(prefix="rt_bench_pmc_", dir="/tmp")
    cmd = [sys.executable, BENCH, "--pmc-child", str(n), str(int(clip))]
    sq = ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU"]
    result, passes = {}, 0

    def one_pass(counters):
        nonlocal passes
        out = os.path.join(work, "_".join(counters))
        res = subprocess.run(
            [exe, "--kernel-trace", "--pmc"] + counters +
            ["--output-format", "csv", "-d", out, "--"] + cmd, cwd="/tmp",
            env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
            text=True, timeout=timeout)
        if res.returncode != 0:
            raise RuntimeError("rocprofv3 --pmc %s: rc %d: %s" % (
                " ".join(counters), res.returncode, res.stderr[-300:]))
        passes += 1
        rows = []
        for path in glob.glob(os.path.join(
                out, "**", "*counter_collection.csv"), recursive=True):
            with open(path) as f:
                rows += list(csv.DictReader(f))
        for counter in counters:
            for kernel, schedule in PMC_SCHEDULE.items():
                mine = sorted(
                    (int(r.get("Dispatch_Id", 0)), float(r["Counter_Value"]))
                    for r in rows if r.get("Counter_Name") == counter and
                    kernel in r.get("Kernel_Name", ""))
                if len(mine) != sum(k for _, k in schedule):   # noqa: E501
                    raise RuntimeError(
                        "%s: %d rows of %s, the schedule has %d" % (
                            counter, len(mine), kernel,
                            sum(k for _, k in schedule)))
                at = 0
                for label, k in schedule:
                    vals = [v for _, v in mine[at + 1:at + k]]
                    result.setdefault(label, {})[counter] = \
                        sum(vals)/len(vals)
                    result[label]["launches"] = len(vals)
                    at += k
    try:
        one_pass(["FETCH_SIZE"])
        try:
            one_pass(["WRITE_SIZE"] + sq)
        except RuntimeError as err:
            log("[bench] WRITE_SIZE + SQ counters in one pass: %s; separate "
                "passes" % (err,))
            one_pass(["WRITE_SIZE"])
            one_pass(sq)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    for rec in result.values():
        # gfx950 corrections of the guide: both counters are KiB; FETCH_SIZE
        # tallies 128-B requests at 64 B
        rec["fetch_bytes_corrected_x2"] = rec.pop("FETCH_SIZE")*1024*2
        rec["write_bytes"] = rec.pop("WRITE_SIZE")*1024
        rec["hbm_bytes_per_launch"] = rec["fetch_bytes_corrected_x2"] + \
            rec["write_bytes"]
    result["_passes"] = passes
    return result


def counters_for_the_line(out, args, n, clip):
    """roofline.traffic (+ the generated batch's) from this run's counter
    passes, or from the committed profile where rocprofv3 cannot run; returns
    the per-workload counters (None: nothing measured in this run)."""
    r = out["roofline"]
    if args.counters == "live":
        try:
            t0 = time.perf_counter()
            c = pmc_live(n, clip)
            took = time.perf_counter() - t0
            lap("live counters (%d rocprofv3 --pmc child runs)" % c["_passes"])
            h = c["headline"]
            r["traffic"] = h["hbm_bytes_per_launch"]
            r["traffic_detail"] = {k: h[k] for k in (
                "fetch_bytes_corrected_x2", "write_bytes", "launches")}
            r["traffic_source"] = (
                "measured in this run: rocprofv3 --kernel-trace --pmc "
                "FETCH_SIZE / WRITE_SIZE (separate passes, %d launches of "
                "this workload each in a child process, %.0f s), FETCH_SIZE "
                "x2 per MI355X_MICROARCH.md (gfx950)" % (h["launches"], took))
            if "SQ_INSTS_VALU" in h:
                S = out["config"]["surfaces"]
                r["valu"] = {
                    "wave_instructions_per_launch": h["SQ_INSTS_VALU"],
                    "per_ray_surface_op": h["SQ_INSTS_VALU"]*64/(n*S),
                    "busy_quad_cycles_per_launch": h["SQ_ACTIVE_INST_VALU"],
                    "source": "this run (the same child passes)"}
            gb = out.get("generated_batch")
            if gb is not None and "generated" in c:
                gb["traffic"] = c["generated"]["hbm_bytes_per_launch"]
                gb["traffic_detail"] = {k: c["generated"][k] for k in (
                    "fetch_bytes_corrected_x2", "write_bytes")}
            return c
        except Exception as err:
            log("[bench] live counters failed: %r" % (err,))
    prof = traffic_from_profile()
    if prof and prof.get("rays") == n and prof.get("clip") == clip:
        r["traffic"] = prof.get("hbm_bytes_per_launch")
        r["traffic_source"] = (
            "profiles/traffic.json (rocprofv3 --pmc passes of this command "
            "on the GPU box, %s; not re-measured in this run)"
            % prof.get("profile", "committed"))
    return None


def valu_roofline(counters, label, rays_counted, rec, n, surfaces, ms,
                  clock_mhz):
    """The FP64-issue ceiling: VALU wave-instructions per launch (SQ counters
    of this run's child passes, scaled to n rays) x 4 cycles / (1024 SIMDs x
    the gfx clock observed during THIS measurement x launch time).  `busy`
    uses SQ_ACTIVE_INST_VALU (quad-cycles the VALUs were executing,
    quarter-rate v_rcp / v_rsq included) instead of the count."""
    c = (counters or {}).get(label)
    if not c or "SQ_INSTS_VALU" not in c or not clock_mhz:
        return
    scale = n/rays_counted
    cyc = 1024*clock_mhz*1e6*ms*1e-3
    rec["valu"] = {
        "wave_instructions_per_launch": c["SQ_INSTS_VALU"]*scale,
        "per_ray_surface_op": c["SQ_INSTS_VALU"]*scale*64/(n*surfaces),
        "gfxclk_mhz_observed": clock_mhz,
        "valu_issue_frac": c["SQ_INSTS_VALU"]*scale*4/cyc,
        "valu_busy_frac": c["SQ_ACTIVE_INST_VALU"]*scale*4/cyc,
        "source": "this run: rocprofv3 --pmc SQ_INSTS_VALU "
                  "SQ_ACTIVE_INST_VALU over %d launches of this workload on "
                  "%d rays in a child process, this leg's clock and launch "
                  "time" % (c["launches"], rays_counted)}


def input_bytes(eng, n):
    """Bytes of the launch rows (row 0 of Y and U) a trace from element 1 has
    to read: 8 per ray and component, except where a component is one bit
    pattern across a 64-ray tile -- the direction of a collimated bundle, z = 0
    of rays starting on a plane: noted by the seed kernel, fetched once per
    tile (8 B) -- plus the 4-byte note per tile; and u2 is not read at all
    where it is, bit for bit, the completion of u0 and u1 (rt_input_completed:
    the trace rebuilds it)."""
    uniform, tiles = eng.input_uniform()
    done = eng.input_completed()    # u2 rebuilt from u0, u1: not read
    if not any(uniform) and not done:
        return 48*n, [0]*6
    return (sum(8*(n - 64*u) + 8*u for u in uniform) + 4*tiles - 8*64*done,
            [u/tiles for u in uniform])


def algorithmic_bytes(tables, n, clip, generated=False, alias=True,
                      pupil_reuse=1, read_bytes=None):
    """HBM bytes one launch has to move for ``n`` rays through the packed
    table(s): per ray-surface op 56 written (y 24, u 24, t 8), + 24 where i
    must be materialised (element j or j-1 tilted), - 24 where an unclipped
    trace leaves u[j] = i[j] (no bend); per ray 48 read (host-seeded rows) or
    16 (pupil coordinates of a device-generated batch)."""
    from rayopt_amd._lib import F_ROTATED, F_REFRACT
    flags = np.atleast_2d(tables["flags"])
    rot = ((flags & F_ROTATED) != 0).any(0)
    bends = ((flags & F_REFRACT) != 0).any(0)
    L = flags.shape[1]
    stored_i = sum(1 for j in range(1, L) if not alias or rot[j] or rot[j - 1])
    skipped_u = sum(1 for j in range(1, L) if alias and not clip
                    and not bends[j])
    per_op = 56*(L - 1) + 24*stored_i - 24*skipped_u
    if read_bytes is None:
        read_bytes = n*(16 if generated else 48)
    return n*per_op + read_bytes, per_op/(L - 1)


def cpu_all_cores(args, system, y, u, clip):
    """The reference (or, without its archive, the numpy port) on many forked
    processes -- before this process opens the GPU."""
    from rayopt_amd import prescriptions as P
    procs = (min(64, os.cpu_count()) if args.cpu_procs == -1 else
             os.cpu_count() if args.cpu_procs < 0 else args.cpu_procs)
    out = None
    if procs > 1:
        try:
            from oracle import refshim
            if refshim.available():
                out = reference_on_processes(
                    P.DOUBLE_GAUSS, y, u, system.wavelengths[0], clip, procs)
                if args.extras:
                    out["port_value"] = cpu_port_on_processes(
                        system, y, u, clip, procs)["value"]
            else:
                out = cpu_port_on_processes(system, y, u, clip, procs)
        except Exception as err:      # a reported extra, never fatal
            out = {"error": repr(err)[:200]}
    lap("reference on %d forked processes" % procs if out else
        "(no many-process CPU leg)")
    return out


def cpu_legs(out, args, table, system, y, u, clip, S, g, L, cpu_all):
    out["cpu_baseline"] = cpu_one_core(table, system, y, u, clip, S, g, L,
                                       args.cpu_sample, g.l)
    if cpu_all is not None:
        out["cpu_baseline_all_cores"] = cpu_all
    try:
        out["cpu_baseline_c"] = cpu_c_oracle(table, y, u, clip, S, g, L,
                                             every_team=args.extras)
    except Exception as err:      # a reported extra, never fatal
        out["cpu_baseline_c"] = {"error": repr(err)[:200]}
    lap("cpu_baseline: reference on one core, C port")


def legs_before_the_loop(args, job, g, mode, step, mark, plain, side_legs):
    """Timed loops of their own that run before the headline's (so that the
    headline loop is the last thing the device did when its clocks are
    read): image row only, the bare engine call; --extras: unclipped, every
    i row stored."""
    eng, clip = g.engine, mode["clip"]
    out = {}
    if plain and (args.extras or side_legs):
        # keep only the image row (merit-function use): the kernel leaves
        # the HBM roofline for the FP64 one
        def step_image():
            g.propagate(clip=clip, keep=[0, -1])
        job.timed(step_image, 300, 10, False)    # its own settled state
        mark("imgrow:begin")
        # (~0.5 s: amdsmi's clock is a moving average of about that length)
        k = max(args.steps, 900)
        e_img, ev_img, _ = job.timed(step_image, k, 0, False)
        mark("imgrow:end")
        out["image_only"] = (e_img/k*args.steps, ev_img/k)
    if args.extras and plain and clip:
        # the reference's default: propagate(clip=False); the u rows of the
        # elements that do not bend the ray are not written either
        mode["clip"] = False
        e_nc, ev_nc, _ = job.timed(step, args.steps, args.warmup, False)
        out["unclipped"] = (e_nc, ev_nc/args.steps)
        mode["clip"] = clip
    if plain and (args.extras or side_legs):
        # SURVEY 8(d)'s literal case: every row of y, u, i, t materialised
        # (rayopt/geometric_trace.py:41-47), 80 B per ray-surface op
        eng.set_option("alias_i", 0)
        e_full, ev_full, _ = job.timed(step, args.steps, args.warmup, False)
        out["full_i"] = (e_full, ev_full/args.steps)
        eng.set_option("alias_i", 1)
    if not args.no_api_leg and not args.no_engine_leg and not job.dist:
        g.propagate(clip=clip)

        def step_engine():      # the bare C-ABI call, table already there
            eng.trace(1, 0, clip)
        e_eng, ev_eng, _ = job.timed(step_engine, args.steps, args.warmup,
                                     False)
        out["engine_leg"] = (e_eng, ev_eng/args.steps)
    return out


def record_legs_before(out, before, args, n, S, L, total_rays, elapsed,
                       read_bytes, table, clip, alias_on):
    from rayopt_amd._lib import F_REFRACT
    if "engine_leg" in before:
        e = before["engine_leg"][0]
        out["propagate_api"] = {
            "engine_trace_ms_per_step": e*1e3/args.steps,
            "propagate_ms_per_step": elapsed*1e3/args.steps,
            "ratio": elapsed/e,
            "note": "`value` times the public GeometricTrace.propagate() "
                    "(re-pack + table hand-over + launch); engine_trace = "
                    "the bare rt_trace call in the same timed loop"}

    def rec(e, k, nbytes, note):
        return {"value": total_rays*S*args.steps/e, "kernel_ms": k,
                "algorithmic_bytes_per_launch": nbytes,
                "achieved": nbytes/(k*1e-3)/1e9,
                "frac": nbytes/(k*1e-3)/1e9/HBM_PEAK_GBS, "note": note}
    if "full_i" in before:
        out["full_i"] = rec(*before["full_i"], n*80*S + read_bytes,
                            "every row of i materialised (alias_i=0): 80 B "
                            "per op")
    if "unclipped" in before:
        bends = (table["flags"] & F_REFRACT) != 0
        nb = read_bytes + n*(56*S - 24*sum(
            1 for j in range(1, L) if alias_on and not bends[j]))
        out["unclipped"] = rec(*before["unclipped"], nb,
                               "propagate(clip=False), the reference's "
                               "default: u rows of stop and image are i "
                               "rows bit for bit and are not written")
    if "image_only" in before:
        e_img, k_img = before["image_only"]
        out["image_row_only"] = {
            "value": total_rays*S*args.steps/e_img, "kernel_ms": k_img,
            "note": "propagate(keep=[0, -1]): all %d surfaces traced, only "
                    "the image row stored (80 B/ray); FP64-VALU bound" % S}


def record_telemetry(out, t, counters, n):
    """Clocks / power around the timed loop; the FP64-issue roofline of the
    image-row-only leg from this run's SQ counters and that window's clock."""
    t = t or {"samples": 0, "error": "the telemetry child did not start"}
    r = out["roofline"]
    uclk = ((t.get("loop") or {}).get("hbm_uclk_mhz") or [None]*3)[1]
    if uclk:
        r["hbm_uclk_mhz_observed"] = uclk
        r["frac_at_observed_hbm_clock"] = \
            r["achieved"]/(HBM_PEAK_GBS*uclk/HBM_NOMINAL_MHZ)
    out["telemetry"] = t
    w = t.get("imgrow") or {}
    clk = (w.get("gfxclk_mhz") or [None]*3)[1]
    io = out.get("image_row_only")
    if io is not None and clk:
        valu_roofline(counters, "image_row_only", n, io, n,
                      out["config"]["surfaces"], io["kernel_ms"], clk)
        if "valu" in io:
            io.update(
                bound="fp64 valu issue",
                power_limited_fraction=w.get("power_limited_fraction"),
                flop_equivalents_per_s=io["value"]*190.,
                flop_equivalent_note="SURVEY 8(d): not an HBM leg (~190 "
                "flop-equivalents per ray-surface op: 70 flop + 3 sqrt + 4 "
                "div); the ceiling is 1024 SIMDs x gfx clock / 4 cycles per "
                "FP64 wave-instruction")


def _hip():
    """libamdhip64 through ctypes: the pinned copies that bound the host
    path (measurement only)."""
    import ctypes
    lib = ctypes.CDLL("libamdhip64.so")
    lib.hipHostMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p),
                                  ctypes.c_size_t, ctypes.c_uint]
    lib.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p),
                              ctypes.c_size_t]
    lib.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p,
                              ctypes.c_size_t, ctypes.c_int]
    lib.hipFree.argtypes = [ctypes.c_void_p]
    lib.hipHostFree.argtypes = [ctypes.c_void_p]
    return lib


def end_to_end(g, y, u, clip, kernel_ms):
    """The drop-in's real use, PCIe inclusive (never `value`): host rays in
    (rayopt/geometric_trace.py:49-70), one trace, the image row out
    (rayopt/analysis.py:237-245) -- and, beside each copy, what a plain
    hipMemcpy between PINNED host memory and the device reaches for the same
    bytes on this box: the ceiling of any staging scheme."""
    import ctypes
    n, L = len(y), len(g.system)
    rec = {"rays": n, "note": "wall clock of the public calls on pageable "
           "numpy arrays; the copies are staged through two pinned buffers "
           "(csrc/rt_engine.hip: rt_h2d / rt_d2h; the way out by a copy "
           "kernel, not the DMA)"}
    t = []
    for _ in range(3):
        t0 = time.perf_counter()
        g.rays_given(y, u)
        t.append(time.perf_counter() - t0)
    rec["h2d_ms"] = min(t)*1e3
    rec["h2d_bytes"] = 48*n
    rec["h2d_GBps"] = 48*n/min(t)/1e9
    t0 = time.perf_counter()
    g.propagate(clip=clip)
    g.engine.sync()
    rec["trace_ms_first_call_after_upload"] = (time.perf_counter() - t0)*1e3
    rec["trace_ms"] = kernel_ms
    t = []
    for k in range(3):
        g.propagate(clip=clip)
        g.engine.sync()
        t0 = time.perf_counter()
        row = np.asarray(g.y[L - 1])
        t.append(time.perf_counter() - t0)
    rec["d2h_image_row_ms_first"] = t[0]*1e3    # fresh host pages
    rec["d2h_image_row_ms"] = min(t[1:])*1e3
    rec["d2h_bytes"] = row.nbytes
    rec["d2h_GBps"] = row.nbytes/min(t[1:])/1e9
    # what the reference's consumers of the image row read is x and y
    # (`t.y[-1, :, :2]`: rayopt/geometric_trace.py:172, analysis.py:237-283):
    # two thirds of the row cross PCIe (rt_download_xy)
    t = []
    for k in range(3):
        g.propagate(clip=clip)
        g.engine.sync()
        t0 = time.perf_counter()
        xy = g.y[L - 1, :, :2]
        t.append(time.perf_counter() - t0)
    rec["d2h_image_xy_ms"] = min(t)*1e3
    rec["d2h_xy_bytes"] = 16*n
    rec["d2h_xy_GBps"] = 16*n/min(t)/1e9
    assert xy.shape == (n, 2)
    rec["end_to_end_ms"] = rec["h2d_ms"] + kernel_ms + rec["d2h_image_xy_ms"]
    rec["end_to_end_full_row_ms"] = rec["h2d_ms"] + kernel_ms + \
        rec["d2h_image_row_ms"]
    try:
        hip = _hip()
        nb = 48*n
        hp, dp = ctypes.c_void_p(), ctypes.c_void_p()
        assert hip.hipHostMalloc(ctypes.byref(hp), nb, 0) == 0
        assert hip.hipMalloc(ctypes.byref(dp), nb) == 0
        ctypes.memset(hp, 1, nb)
        best = {}
        for kind, (dst, src, size) in (("h2d", (dp, hp, nb)),
                                       ("d2h", (hp, dp, 24*n))):
            t = []
            for _ in range(4):
                t0 = time.perf_counter()
                assert hip.hipMemcpy(dst, src, size,
                                     1 if kind == "h2d" else 2) == 0
                t.append(time.perf_counter() - t0)
            best[kind] = size/min(t[1:])/1e9
        hip.hipFree(dp)
        hip.hipHostFree(hp)
        rec["pinned_hipMemcpy_ceiling_GBps"] = best
        rec["h2d_fraction_of_ceiling"] = rec["h2d_GBps"]/best["h2d"]
        rec["d2h_fraction_of_ceiling"] = rec["d2h_GBps"]/best["d2h"]
        rec["d2h_xy_fraction_of_ceiling"] = rec["d2h_xy_GBps"]/best["d2h"]
    except Exception as err:      # a reported extra, never fatal
        rec["pinned_hipMemcpy_ceiling_GBps"] = {"error": repr(err)[:200]}
    return rec


def side_legs(out, ra, g, system, device, args, n, S, clip, kernel_ms,
              alg_bytes, achieved, counters, y, u, upload_s):
    """One record per BASELINE config, and the host path."""
    try:
        out["configs"] = [{
            "config": "C3 double-Gauss, %d rays in 5 field bundles (the "
                      "headline line above)" % n,
            "rays": n, "surfaces": S, "clip": clip, "kernel_ms": kernel_ms,
            "value": n*S/(kernel_ms*1e-3),
            "algorithmic_bytes_per_launch": alg_bytes,
            "achieved": achieved, "frac": achieved/HBM_PEAK_GBS}] + \
            run_configs(ra, device, args, counters)
    except Exception as err:      # reported extras, never fatal
        out["configs"] = {"error": repr(err)[:300]}
    lap("configs C1 C2 C3' C4 C4x C5 (with their parity subsamples)")
    try:
        out["end_to_end"] = end_to_end(g, y, u, clip, kernel_ms)
        out["end_to_end"]["first_rays_given_s_with_allocation"] = upload_s
    except Exception as err:
        out["end_to_end"] = {"error": repr(err)[:300]}
    lap("host path (end to end)")


def kernel_ms_of(g, clip, settle_s=.3, per_block=10, dwell_s=.35, mark=None):
    """Launch time of one propagate() of the resident batch, measured like
    the headline's: after a settle phase of back-to-back launches (clocks and
    power filter in their loaded state), HIP events around blocks of
    back-to-back launches for ``dwell_s`` seconds; the median block / launches
    per block.  ``mark(label)``: called at the begin and the end of the dwell
    phase (the telemetry window: amdsmi's clocks are ~0.5 s moving averages,
    a shorter window would report the state before it)."""
    eng = g.engine
    g.propagate(clip=clip)
    eng.sync()
    if g.kernel_ms() < .1:
        # launch bound (C1): the kernel's own events, one launch at a time
        t = []
        for _ in range(40):
            g.propagate(clip=clip)
            t.append(g.kernel_ms())
        return float(np.median(t[10:]))
    per = max(1, min(per_block, int(40./max(g.kernel_ms(), 1e-3))))
    t_end = time.perf_counter() + settle_s
    while time.perf_counter() < t_end:
        for _ in range(per):
            g.propagate(clip=clip)
        eng.sync()
    if mark is not None:
        mark("begin")
    t = []
    t_end = time.perf_counter() + dwell_s
    while time.perf_counter() < t_end or len(t) < 5:
        eng.event_record(0)
        for _ in range(per):
            g.propagate(clip=clip)
        eng.event_record(1)
        t.append(eng.event_elapsed(0, 1)/per)
    if mark is not None:
        mark("end")
    return float(np.median(t))


def subsample_parity(ra, device, system, y, u, l, clip, options, m=100_000):
    """The first ``m`` rays traced on their own with the same options against
    the plain-C oracle (a ray's result does not depend on its batch): every
    value of y, u, i, t of every row.  Exact arithmetic: bit identity;
    default asphere arithmetic: worst error relative to the row scale and
    whether the NaN masks are the same."""
    from oracle import build_c
    from rayopt_amd.pack import pack_system
    y, u = np.ascontiguousarray(y[:m]), np.ascontiguousarray(u[:m])
    g = ra.GeometricTrace(system, device=device, **options)
    g.rays_given(y, u, l)
    g.propagate(clip=clip)
    ls = list(np.atleast_1d(l if l is not None else system.wavelengths[0]))
    same, masks, worst = True, True, 0.
    per = len(y)
    for k, lk in enumerate(ls):
        table, _ = pack_system(system, lk, system.refractive_index(lk, 0))
        want = build_c.propagate(table, y, u, clip=clip)
        for rows, ref in zip((g.y, g.u, g.i, g.t), want):
            got = np.asarray(rows[1:])[:, k*per:(k + 1)*per]
            same = same and np.array_equal(got, ref, equal_nan=True)
            masks = masks and np.array_equal(np.isnan(got), np.isnan(ref))
            with np.errstate(all="ignore"):
                for a, b in zip(got, ref):
                    fin = np.isfinite(a) & np.isfinite(b)
                    if fin.any():
                        scale = np.abs(b[fin]).max()
                        worst = max(worst, float(
                            (np.abs(a[fin] - b[fin]) /
                             np.maximum(np.abs(b[fin]), scale)).max()))
    return {"rays": per, "bit_identical_to_c_oracle": bool(same),
            "nan_masks_equal": bool(masks), "max_rel_err": worst}


def parity_sample_of_a_large_batch(g, system, count):
    """Every (n / count)-th ray of a batch too large to bring down, gathered
    on the device across all surfaces (rt_download_rays) and compared with
    the C oracle started from the launch rays the device itself built: all
    blocks and all bundles are sampled."""
    from rayopt_amd._lib import RT_Y, RT_U, RT_I, RT_T
    from rayopt_amd.pack import pack_system
    from oracle import build_c
    build_c.build()
    n = g.nrays
    stride = max(1, n//count)
    count = min(count, (n - 1 - 137 % n)//stride + 1)
    cols = {w: g.engine.download_rays(w, 137 % n, stride, count)
            for w in (RT_Y, RT_U, RT_I, RT_T)}
    table, _ = pack_system(system, g.l, g.n[0])
    want = build_c.propagate(table, np.ascontiguousarray(cols[RT_Y][0]),
                             np.ascontiguousarray(cols[RT_U][0]), clip=True)
    same, nan_ok, err = True, True, 0.
    for w, ref in zip((RT_Y, RT_U, RT_I, RT_T), want):
        got = cols[w][1:]
        same = same and bool(np.array_equal(got, ref, equal_nan=True))
        nan_ok = nan_ok and bool(np.array_equal(np.isnan(got),
                                                np.isnan(ref)))
        fin = np.isfinite(ref) & np.isfinite(got)
        if fin.any():
            err = max(err, float((np.abs(got[fin] - ref[fin]) /
                                  np.maximum(np.abs(ref[fin]), 1.)).max()))
    nb = g.engine.blocks()[0]
    return {"rays": int(count), "stride": int(stride),
            "blocks_of_the_batch": int(nb),
            "bit_identical_to_c_oracle": same, "nan_masks_equal": nan_ok,
            "max_rel_err": err,
            "finite_fraction_at_image": float(
                np.isfinite(cols[RT_U][-1][:, 0]).mean())}


CONFIG_KEYS = ("C1", "C2", "C3p", "C4", "C4x", "C5")
ONLY_KEYS = CONFIG_KEYS + ("C3", "gen")   # + the headline batch, host-seeded
                                          # and built on the device


def config_trace(ra, device, key, args):
    """The workload of one BASELINE config, resident: dict with ``g`` (rays
    set, not traced), ``system``, the host rays ``y, u`` (None for the batch
    built on the device), wavelength(s) ``l``, ``n`` rays, the engine
    ``options``, ``generated``, ``name``, ``kind``, ``note`` and the
    reference's prescription ``text``."""
    from rayopt_amd import prescriptions as P
    import digest_cases as dc
    big = 10_000_000 if not args.rays or args.rays >= 10**6 else args.rays
    c = {"key": key, "options": {}, "generated": False, "note": "",
         "kind": None, "y": None, "u": None}
    if key == "C1":
        # singlet, 10^4 rays, one wavelength (launch bound: 160 wavefronts)
        c["system"] = ra.system_from_yaml(P.SINGLET)
        c["text"] = P.SINGLET
        c["y"], c["u"] = dc.bundle(10**4, 8., 0., 0)
        c["l"] = c["system"].wavelengths[0]
        c["name"] = "C1 singlet, 10^4 rays"
        c["note"] = ("160 wavefronts on 256 CUs: bound by launch latency, "
                     "not HBM")
    elif key == "C2":
        # Cooke triplet, 10^6 rays x 3 wavelengths as ONE launch (ray groups)
        c["system"] = ra.system_from_yaml(P.COOKE % dict(
            air="air", sk16="SCHOTT-SK|N-SK16", f2="SCHOTT-F|N-F2"))
        c["text"] = None
        c["y"], c["u"] = dc.bundle(10**6, 5.5, 5., 0)
        c["l"] = [587.56e-9, 656.27e-9, 486.13e-9]
        c["name"] = "C2 Cooke triplet, 10^6 rays x 3 wavelengths, one launch"
        c["kind"] = "C2 3 x 10^6 rays"
    elif key == "C3p":
        # C3 with launch directions that differ from ray to ray (a bundle as
        # rays_given gets it from a caller's own generator): only z = 0 is
        # uniform across a 64-ray tile, the notes save 8 of 48 B per ray
        c["system"] = ra.system_from_yaml(P.DOUBLE_GAUSS)
        c["text"] = P.DOUBLE_GAUSS
        y, u = workload_rays(big, 7)
        rng = np.random.default_rng(3)
        u[:, 0] += 1e-7*rng.standard_normal(big)
        u[:, 1] += 1e-7*rng.standard_normal(big)
        u[:, 2] = np.sqrt(1. - u[:, 0]**2 - u[:, 1]**2)
        c["y"], c["u"] = y, u
        c["l"] = c["system"].wavelengths[0]
        c["name"] = ("C3 double-Gauss, %d rays, per-ray launch directions "
                     "(no uniform direction to fetch once per wavefront)"
                     % big)
        c["note"] = ("the headline's bundles are collimated: their direction "
                     "is read once per 64-ray tile; this is what a bundle "
                     "with individual directions costs")
        c["kind"] = "C3 host-seeded clip"
    elif key in ("C4", "C4x"):
        # aspheric phone lens, 10^7 rays: default and exact arithmetic
        c["system"] = ra.system_from_yaml(P.ASPHERE_PHONE)
        c["text"] = P.ASPHERE_PHONE
        y, u = dc.bundle(big, .6, 10., 4)
        y[:, 1] -= .5*np.tan(np.radians(10.))
        c["y"], c["u"] = y, u
        c["l"] = c["system"].wavelengths[0]
        if key == "C4x":
            c["options"] = {"exact_asphere": 1}
        c["name"] = "C4 aspheric phone lens, %d rays, %s" % (
            big, "exact_asphere=True (the reference's bits)" if key == "C4x"
            else "default (FMA / rcp / rsq Newton, 1e-8 contract)")
        c["kind"] = "C4 exact" if key == "C4x" else "C4 default"
    elif key == "C3":
        # the headline batch itself (--only-config: a leg's profile)
        c["system"] = ra.system_from_yaml(P.DOUBLE_GAUSS)
        c["text"] = P.DOUBLE_GAUSS
        c["y"], c["u"] = workload_rays(big, 0)
        c["l"] = c["system"].wavelengths[0]
        c["name"] = "C3 double-Gauss, %d rays in 5 field bundles" % big
    elif key in ("C5", "gen"):
        # double-Gauss, 10^8 rays built on the device (104 GB), ONE GPU
        # ("gen": the headline's 10^7 rays built on the device)
        c["system"] = ra.system_from_yaml(P.DOUBLE_GAUSS)
        c["text"] = P.DOUBLE_GAUSS
        c["l"] = c["system"].wavelengths[0]
        c["generated"] = True
        c["kind"] = "C5 generated"
        c["note"] = ("the 8-GPU form shards these rays and gathers y[L-1] "
                     "over RCCL (bench.py --gpus 8 --total-rays 100000000)")
    else:
        raise ValueError("config %r: one of %s" % (key, ONLY_KEYS))
    g = ra.GeometricTrace(c["system"], device=device, **c["options"])
    if c["generated"]:
        nf = len(FIELD_FRACTIONS)
        m = (big if key == "gen" else
             args.configs5_rays or 100_000_000)//nf//64*64
        c["pupil_points"] = m
        g.rays_fields(np.c_[np.zeros(nf), FIELD_FRACTIONS],
                      dc.disc_points(m, 91), P.DOUBLE_GAUSS_PUPIL_Z,
                      BUNDLE_RADIUS)
        c["n"] = m*nf
        c["name"] = ("%s: double-Gauss, %d rays built on the device" % (
            "C5 on one GPU" if key == "C5" else "generated batch", c["n"]))
    else:
        g.rays_given(c["y"], c["u"], c["l"])
        c["n"] = len(c["y"])*len(np.atleast_1d(c["l"]))
    c["g"] = g
    return c


def only_config(ra, device, key, args):
    """``bench.py --only-config KEY``: ONE config leg and nothing else in the
    process -- what a `rocprofv3 --kernel-trace --stats` run of a leg needs
    (profiles/r06_final/legs/): every launch of the trace kernel in the
    summary is this workload's.  Prints its own small JSON line."""
    c = config_trace(ra, device, key, args)
    g = c["g"]
    g.propagate(clip=True)
    ms = kernel_ms_of(g, True)
    pl = g.engine.placement()
    return {"only_config": key, "config": c["name"], "rays": c["n"],
            "surfaces": len(c["system"]) - 1, "kernel_ms": ms,
            "kernel": "rt_trace_gen_kernel" if c["generated"]
            else "rt_trace_kernel",
            "store_pattern_GBps_per_piece_set":
                pl["store_pattern_GBps_per_piece_set"],
            "per_class": pl["per_class"]}


def run_configs(ra, device, args, live=None):
    """One record per BASELINE config (C3 is the headline itself)."""
    from rayopt_amd import prescriptions as P
    from rayopt_amd.pack import pack_system
    from oracle import refshim
    import digest_cases as dc
    out = []

    def reference_rate(text, y, u, l, clip, m):
        if not refshim.available():
            return None
        rec, _ = reference_one_process(text, y, u, l, clip, m)
        return {k: rec[k] for k in ("value", "rays", "seconds", "kind")}

    tele = Telemetry(device, period=0.01)
    windows = []
    # which workload of this run's counter passes a record's kernel is, and
    # the rays it was counted on (bench_legs.PMC_SCHEDULE)
    counted = {"C4 default": ("C4 default", PMC_C4_RAYS),
               "C4 exact": ("C4 exact", PMC_C4_RAYS),
               "C3 host-seeded clip": ("headline", args.rays),
               "C5 generated": ("generated", args.rays)}

    def record(name, system, g, n, l, clip, generated, parity, ref, note="",
               kind=None):
        ls = np.atleast_1d(l)
        tables = np.stack([pack_system(system, lk,
                                       system.refractive_index(lk, 0))[0]
                           for lk in ls])
        k = len(windows)
        ms = kernel_ms_of(g, clip, mark=(
            (lambda what: tele.mark("%d:%s" % (k, what)))
            if tele is not None else None))
        windows.append(kind)
        rb, uni = (None, None) if generated else input_bytes(g.engine, n)
        alg, per_op = algorithmic_bytes(tables, n, clip, generated,
                                        read_bytes=rb)
        S = len(system) - 1
        rec = {"config": name, "rays": n, "surfaces": S, "clip": clip,
               "kernel_ms": ms, "value": n*S/(ms*1e-3),
               "algorithmic_bytes_per_launch": alg,
               "bytes_per_ray_surface_op": per_op,
               "input_bytes_per_ray": (rb if rb is not None else 16*n)/n,
               "frac_if_48B_per_ray_were_read":
                   (alg - (rb if rb is not None else 16*n) + 48*n) /
                   (ms*1e-3)/1e9/HBM_PEAK_GBS,
               "achieved": alg/(ms*1e-3)/1e9,
               "frac": alg/(ms*1e-3)/1e9/HBM_PEAK_GBS,
               "parity_subsample": parity,
               "cpu_reference": ref}
        pl = g.engine.placement()
        rec["placement"] = {k: pl[k] for k in (
            "pieces", "piece_mib", "per_class", "fast", "store_pattern_GBps",
            "store_pattern_GBps_per_piece_set", "created", "ballast_blocks",
            "search_ms", "search_cut_short")}
        rec["_kind"] = kind
        if kind and kind.startswith("C4"):
            # how many of the lanes a wavefront drags through the asphere
            # iteration are still iterating (rt_newton_census: the batch
            # marched again by a kernel that counts and stores nothing)
            try:
                c = g.engine.newton_census(clip)
                rec["newton_census"] = c
                rec["newton_lane_utilisation"] = c["lane_utilisation"]
            except Exception as err:      # a reported extra, never fatal
                rec["newton_census"] = {"error": repr(err)[:200]}
        if note:
            rec["note"] = note
        out.append(rec)
        log("[configs] %s: %.4f ms, frac %.3f" % (name, ms, rec["frac"]))

    ref4 = None
    for key in CONFIG_KEYS:
        if key == "C5" and args.no_configs5:
            continue
        c = config_trace(ra, device, key, args)
        g, system, y, u, l = c["g"], c["system"], c["y"], c["u"], c["l"]
        if key == "C5":
            g.propagate(clip=True)
            parity = parity_sample_of_a_large_batch(g, system, 10_000)
            ref = None
        else:
            m = {"C2": 64*1500}.get(key, 100_000)
            parity = subsample_parity(ra, device, system, y, u, l, True,
                                      c["options"], m)
            ref = None
            if key == "C1":
                ref = reference_rate(c["text"], y, u, l, True, 10**4)
            elif key == "C2" and refshim.available():
                rates = [reference_rate(P.cooke(lk), y, u, lk, True, 100_000)
                         for lk in l]
                ref = {"value": sum(r["rays"] for r in rates) *
                       (len(system) - 1)/sum(r["seconds"] for r in rates),
                       "kind": "reference", "rays": rates[0]["rays"],
                       "note": "three traces, one per wavelength, as the "
                               "reference has to run them"}
            elif key in ("C4", "C4x"):
                if ref4 is None:
                    ref4 = reference_rate(c["text"], y, u, l, True, 3000)
                    if ref4 is not None:
                        ref4["note"] = (
                            "per-ray scipy.optimize.newton in a Python loop "
                            "(rayopt/elements.py:333-349): timed on 3000 "
                            "rays and extrapolated, BASELINE.md 3.4")
                ref = ref4
        record(c["name"], system, g, c["n"], l, True, c["generated"], parity,
               ref, c["note"], kind=c["kind"])
        if key != "C5":
            del g, c, y, u
            continue
        s5, m, nf = system, c["pupil_points"], len(FIELD_FRACTIONS)
        out[-1]["finite_fraction_at_image_sampled"] = \
            out[-1]["parity_subsample"].pop("finite_fraction_at_image")
        if args.extras:
            # the same launch with 30 ms of idle device before it (round 5:
            # is the one batch slowed by sitting at the power cap launch
            # after launch?  No: after an idle gap it is 20 % SLOWER -- the
            # clocks have to come back -- and ten batches in turn run at the
            # cap just the same)
            t = []
            for _ in range(12):
                g.engine.sync()
                time.sleep(.03)
                g.propagate(clip=True)
                g.engine.sync()
                t.append(g.kernel_ms())
            out[-1]["launches_30_ms_apart"] = {
                "kernel_ms": float(np.median(t[2:])),
                "kernel_ms_min_max": [min(t[2:]), max(t[2:])]}
            log("[configs] C5, launches 30 ms apart: %.4f ms"
                % out[-1]["launches_30_ms_apart"]["kernel_ms"])
        del g, c
        # the same rays as TEN batches of a tenth each, ten contexts traced
        # in turn: the cross-check of the layout in blocks (csrc/rt_lay.h) --
        # as ONE block this batch took 12.0 ms, the ten batches 10.3; in
        # blocks 10.25 against 9.9, and with the bundles taken in turns
        # (rt_gen_wg: the pupil points stay cached) the two agree
        # 
        try:
            if not args.extras:
                raise StopIteration
            parts = 10
            mk = m//parts//64*64
            gs = []
            for i in range(parts):
                gk = ra.GeometricTrace(s5, device=device)
                gk.rays_fields(np.c_[np.zeros(nf), FIELD_FRACTIONS],
                               dc.disc_points(mk, 92 + i),
                               P.DOUBLE_GAUSS_PUPIL_Z, BUNDLE_RADIUS)
                gk.propagate(clip=True)
                gs.append(gk)

            def turn():
                for gk in gs:
                    gk.engine.trace(1, 0, True)
                for gk in gs:
                    gk.engine.sync()
            t_end = time.time() + .5
            while time.time() < t_end:
                turn()
            if tele is not None:
                tele.mark("ten:begin")
            t0 = time.perf_counter()
            for _ in range(20):
                turn()
            ms = (time.perf_counter() - t0)/20*1e3
            if tele is not None:
                tele.mark("ten:end")
            tables = np.stack([pack_system(
                s5, s5.wavelengths[0],
                s5.refractive_index(s5.wavelengths[0], 0))[0]])
            alg = parts*algorithmic_bytes(tables, mk*nf, True, True)[0]
            out[-1]["as_ten_batches_in_turn"] = {
                "rays": parts*mk*nf, "batches": parts,
                "ms_per_turn_wall": ms,
                "value": parts*mk*nf*(len(s5) - 1)/(ms*1e-3),
                "algorithmic_bytes_per_turn": alg,
                "frac": alg/(ms*1e-3)/1e9/HBM_PEAK_GBS,
                "note": "wall clock around 20 turns of ten propagate() "
                        "launches (one context each) and their syncs"}
            log("[configs] C5 as ten batches of %d rays in turn: %.4f ms, "
                "frac %.3f" % (mk*nf, ms,
                               out[-1]["as_ten_batches_in_turn"]["frac"]))
            del gs
        except StopIteration:
            pass
        except Exception as err:      # a reported extra, never fatal
            out[-1]["as_ten_batches_in_turn"] = {"error": repr(err)[:200]}
    # what bounds each config: the store streams (HBM) or FP64 issue
    t = tele.stop() if tele is not None else None
    w = (t or {}).get("ten") or {}
    if w and "as_ten_batches_in_turn" in out[-1]:
        # (both forms hold the socket at its power cap: the clocks do not
        # tell them apart)
        out[-1]["as_ten_batches_in_turn"]["telemetry"] = {
            "gfxclk_mhz": (w.get("gfxclk_mhz") or [None]*3)[1],
            "socket_power_w": (w.get("socket_power_w") or [None]*3)[1],
            "power_limited_fraction": w.get("power_limited_fraction")}
    for k, rec in enumerate(out):
        kind = rec.pop("_kind", None)
        w = (t or {}).get(str(k)) or {}
        clock = (w.get("gfxclk_mhz") or [None]*3)[1]
        if w:
            rec["telemetry"] = {
                "gfxclk_mhz": clock,
                "socket_power_w": (w.get("socket_power_w") or [None]*3)[1],
                "power_limited_fraction": w.get("power_limited_fraction")}
        if kind in counted:
            valu_roofline(live, counted[kind][0], counted[kind][1], rec,
                          rec["rays"], rec["surfaces"], rec["kernel_ms"],
                          clock)
        v = rec.get("valu")
        if rec["kernel_ms"] < .05:
            rec["bound"] = "launch latency"
        elif v and v["valu_busy_frac"] > rec["frac"]:
            rec["bound"] = "fp64 valu issue"
        else:
            rec["bound"] = "hbm"
    return out


def run_consumers(ra, g, system, n, nf, cpu):
    """The device-side consumers on the resident headline
    batch: streaming reductions over one or two rows.  Wall time per call
    (each returns a scalar or a small array to the host, i.e. includes its own
    synchronisation), algorithmic bytes read, fraction of the 8 TB/s spec."""
    eng, L = g.engine, len(system)
    g.propagate(clip=True)
    eng.sync()

    def timed(fn, reps=20):
        fn()
        eng.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        return (time.perf_counter() - t0)/reps*1e3

    def device_ms(fn, reps=10):
        # the call's kernels alone, between HIP events on the engine's stream
        # (option "consumer_events"; the wall time above includes the launch
        # and the host's wait for the scalar)
        eng.set_option("consumer_events", 1)
        try:
            ms = []
            for _ in range(reps):
                fn()
                ms.append(eng.kernel_ms())
        finally:
            eng.set_option("consumer_events", 0)
        return float(np.mean(ms))

    def rec(name, ms, nbytes, replaces, note="", fn=None):
        r = {"call": name, "ms": ms, "bytes_read": nbytes,
             "GBps": nbytes/(ms*1e-3)/1e9,
             "frac": nbytes/(ms*1e-3)/1e9/HBM_PEAK_GBS,
             "replaces": replaces}
        if fn is not None:
            r["kernel_ms"] = device_ms(fn)
            r["kernel_frac"] = nbytes/(r["kernel_ms"]*1e-3)/1e9/HBM_PEAK_GBS
        if note:
            r["note"] = note
        return r
    out = []
    def both(name, fn, nbytes, replaces):
        # the shipped one-pass reduction, and the two passes it replaced
        r = rec(name, timed(fn), nbytes, replaces, fn=fn)
        eng.set_option("consumers_one_pass", 0)
        try:
            r["two_pass_ms"] = timed(fn)
        finally:
            eng.set_option("consumers_one_pass", 1)
        return r
    out.append(both("rms (one pass over y0, y1, shifted by ray 0)",
                    lambda: g.rms(), 16*n,
                    "rayopt/geometric_trace.py:171-183"))
    out.append(both("refocus_shift (one pass over y0 y1 i0 i1 i2)",
                    lambda: eng.refocus_shift(L - 1), 40*n,
                    "rayopt/geometric_trace.py:82-97 (the sums; the "
                    "re-propagate of :98-99 is one more trace)"))
    out.append(rec("spot_stats, %d field bundles (two passes over y0, y1)"
                   % nf, timed(lambda: eng.spot_stats(L - 1, n//nf, nf)),
                   32*n, "per-field rms of rayopt/analysis.py spot diagrams",
                   fn=lambda: eng.spot_stats(L - 1, n//nf, nf)))
    out.append(rec("row_rmax (one pass over y0, y1)",
                   timed(lambda: eng.row_rmax(L - 1)), 16*n,
                   "rayopt/geometric_trace.py:185-193 resize()",
                   fn=lambda: eng.row_rmax(L - 1)))
    try:
        # the three calls above in ONE pass over y0, y1 (rt_row_stats): per
        # bundle count, centroid, rms about the mean and about the reference
        # ray, the largest distance from the axis
        fn = lambda: eng.row_stats(L - 1, n//nf, nf, 0)     # noqa: E731
        r = rec("row_stats, %d field bundles (ONE pass over y0, y1: count, "
                "centroid, rms about mean and about the reference ray, r "
                "max)" % nf, timed(fn), 16*n,
                "rayopt/geometric_trace.py:171-193 + analysis.py:250-283: "
                "replaces rms + spot_stats + row_rmax", fn=fn)
        three = [c for c in out if c["call"].split(" ")[0].rstrip(",") in
                 ("rms", "spot_stats", "row_rmax")]
        r["replaces_ms"] = sum(c["ms"] for c in three)
        r["replaces_bytes_read"] = sum(c["bytes_read"] for c in three)
        out.append(r)
    except Exception as err:                  # a reported extra, never fatal
        out.append({"call": "row_stats", "error": repr(err)[:200]})
    try:
        nrows = L - 1
        ms = timed(lambda: g.opd_rays(radius=100.), reps=4)
        out.append(rec(
            "opd_rays (t rows 0..%d, y/u of the last element, y[0]; x y t "
            "per ray written AND copied to the host)" % (nrows - 1), ms,
            (8*nrows + 72)*n, "rayopt/geometric_trace.py:101-131",
            "the call returns three host arrays: 24 B/ray cross PCIe inside "
            "the timed region, which is what bounds it"))
    except Exception as err:                  # a reported extra, never fatal
        out.append({"call": "opd_rays", "error": repr(err)[:200]})
    try:
        # the same path differences reduced on the device: mean / rms / P-V
        # per field bundle, nothing per ray over PCIe (rt_opd_stats).  The
        # kernel's own time is between the events rt_opd_stats records.
        nrows = L - 1
        stats = g.opd_stats(radius=100., bundles=nf)
        ms = timed(lambda: g.opd_stats(radius=100., bundles=nf), reps=10)
        kms = []
        for _ in range(5):
            g.opd_stats(radius=100., bundles=nf)
            kms.append(eng.kernel_ms())
        r = rec("opd_stats, %d field bundles (t rows 0..%d, y/u of the last "
                "element, y[0] read once; mean, rms and P-V of the OPD per "
                "bundle come back)" % (nf, nrows - 1), ms, (8*nrows + 72)*n,
                "rayopt/geometric_trace.py:101-131 up to the resampling")
        r["kernel_ms"] = float(np.mean(kms))
        r["kernel_frac"] = r["bytes_read"]/(r["kernel_ms"]*1e-3)/1e9 / \
            HBM_PEAK_GBS
        r["opd_rms_waves_per_bundle"] = stats[:, 3].tolist()
        r["opd_pv_waves_per_bundle"] = stats[:, 6].tolist()
        out.append(r)
    except Exception as err:
        out.append({"call": "opd_stats", "error": repr(err)[:200]})
    try:
        from rayopt_amd.aiming import FieldAimer
        from rayopt_amd import prescriptions as P
        s2 = ra.system_from_yaml(P.cooke().replace("radius: 20.",
                                                   "radius: 0.364"))
        s2.update()
        fields = np.c_[np.zeros(2000), np.linspace(0., 1., 2000)]
        aimer = FieldAimer(s2, s2.wavelengths[0], eng, aim=None)
        ms = timed(lambda: aimer.pupil(fields), reps=5)
        out.append({"call": "aim_pupil, 2000 fields of the Cooke triplet "
                            "(chief + four marginal root finds each)",
                    "ms": ms, "fields_per_s": 2000/(ms*1e-3),
                    "replaces": "rayopt/system.py:507-593 (~130 serial "
                                "one-ray traces per field)",
                    "bound": "latency: 8000 lanes, one per root find"})
        g.rays_given  # (the aimer used its own small batch on this engine)
    except Exception as err:
        out.append({"call": "aim_pupil", "error": repr(err)[:200]})
    if cpu and cpu.get("reference_rms_seconds"):
        out[0]["cpu_reference"] = {
            "seconds": cpu["reference_rms_seconds"], "rays": cpu["rays"],
            "rays_per_s": cpu["rays"]/cpu["reference_rms_seconds"],
            "device_rays_per_s": n/(out[0]["ms"]*1e-3)}
    return out


def small_batch_latency(ra, system, device, n=10_000, reps=300):
    """Wall time of one propagate() on a small batch: the launch-bound regime
    of aiming iterations and merit evaluations, where the host path (re-pack,
    table hand-over) decides."""
    from rayopt_amd import prescriptions as P
    y, u = ra.bundles.disc_bundle(n, BUNDLE_RADIUS, 5., 1,
                                  P.DOUBLE_GAUSS_PUPIL_Z)
    g = ra.GeometricTrace(system, device=device)
    g.rays_given(y, u)
    for _ in range(50):
        g.propagate(clip=True)
    g.engine.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.propagate(clip=True)
    g.engine.sync()
    wall = (time.perf_counter() - t0)/reps
    return {"small_batch_rays": n, "small_batch_propagate_us": wall*1e6,
            "small_batch_kernel_us": g.kernel_ms()*1e3}


def settle(g, seconds, clip):
    """Untimed launches until the device runs at its sustained clocks: the
    host work of a setup phase (ray generation, uploads) lets them drop, and
    the first ~50 launches after it are ~10 % slower."""
    if seconds <= 0:
        return
    eng = g.engine
    g.propagate(clip=clip)
    t_end = time.perf_counter() + seconds
    while time.perf_counter() < t_end:
        for _ in range(10):
            eng.trace(1, 0, clip)
        eng.sync()


def run_generated(ra, system, device, n, clip, args):
    """The same workload with the bundles built on the device (five field
    points x n/5 pupil points, `rays_fields`: the counterpart of the
    reference's rays_point entry) instead of handed over with rays_given.
    Every timed step is the public propagate() on the resident batch; a
    re-trace of a generated batch builds its launch rays again in registers
    rather than read row 0, so the launch writes 56 B per ray-surface op and
    reads 16 B per ray: the pupil coordinates (a pupil point is shared by
    the five fields, but its five uses are a fifth of the launch apart, so
    the L2 sees it five times -- what the fetch counter confirms)."""
    from rayopt_amd import prescriptions as P
    nf = len(FIELD_FRACTIONS)
    m = n//nf//64*64
    rng = np.random.default_rng(7000)
    r, phi = np.sqrt(rng.random(m)), 2*np.pi*rng.random(m)
    yp = np.c_[r*np.cos(phi), r*np.sin(phi)]
    fields = np.c_[np.zeros(nf), FIELD_FRACTIONS]
    g = ra.GeometricTrace(system, device=device)
    g.rays_fields(fields, yp, P.DOUBLE_GAUSS_PUPIL_Z, BUNDLE_RADIUS)
    job = Job(args, None, g, None, 0)
    g.propagate(clip=clip)      # the first trace writes row 0 as well
    settle(g, args.settle, clip)

    def step():
        g.propagate(clip=clip)
    elapsed, ev_ms, _ = job.timed(step, args.steps, args.warmup, False)
    S = len(system) - 1
    rays = m*nf
    kernel_ms = ev_ms/args.steps
    alg = rays*(56*S + 16)
    ulast = np.asarray(g.u[S])
    return {
        "workload": "the same five field bundles built on the device "
                    "(rays_fields, %d rays), one step = one "
                    "GeometricTrace.propagate() re-tracing the resident "
                    "batch" % rays,
        "rays": rays,
        "value": rays*S*args.steps/elapsed,
        "ms_per_step": elapsed*1e3/args.steps,
        "kernel_ms": kernel_ms,
        "algorithmic_bytes_per_launch": alg,
        "achieved": alg/(kernel_ms*1e-3)/1e9,
        "frac": alg/(kernel_ms*1e-3)/1e9/HBM_PEAK_GBS,
        "finite_fraction_at_image": float(np.isfinite(ulast[:, 0]).mean()),
    }


def run_configs4(ra, system, g, job, group, world, rank, args, clip,
                 total=100_000_000):
    """BASELINE configs[4]: 10^8 rays in total, sharded over the N GPUs; the
    rays are built on the device (five field bundles per rank, pupil points
    seeded per rank), results stay in HBM, one RCCL gather of y[L-1] to rank
    0 after the last step inside the timed region."""
    from rayopt_amd import distributed as D
    from rayopt_amd import prescriptions as P
    counts = D.shard_counts(total, world)
    nf = len(FIELD_FRACTIONS)
    m = int(counts[rank])//nf//64*64       # pupil points per field bundle
    box = group.gather(m*nf)
    counts = group.broadcast(np.array(box, dtype=np.int64)
                             if rank == 0 else None)
    rng = np.random.default_rng(7000 + rank)
    r, phi = np.sqrt(rng.random(m)), 2*np.pi*rng.random(m)
    yp = np.c_[r*np.cos(phi), r*np.sin(phi)]
    fields = np.c_[np.zeros(nf), FIELD_FRACTIONS]
    eng = g.engine
    g.rays_fields(fields, yp, P.DOUBLE_GAUSS_PUPIL_Z, BUNDLE_RADIUS)
    L = len(system)
    S = L - 1
    job.counts = counts
    if rank == 0:
        job.d_dst = eng.scratch(int(counts.sum())*3*8)

    def step():
        g.propagate(clip=clip)

    def last_step():
        g.propagate(clip=clip, chunks=job.chunks,
                    after_chunk=job.gather_chunk)
    settle(g, args.settle, clip)
    elapsed, ev_ms, _ = job.timed(
        step, args.steps, args.warmup, True,
        last_step if (job.chunks > 1 and job.exchange) else None)
    exposed = None
    if job.exchange:
        tot_ms, exp_ms = eng.gather_ms()
        exposed = [group.allreduce_max(tot_ms), group.allreduce_max(exp_ms)]
    job.fence()
    t0 = time.perf_counter()
    job.gather()
    job.fence()
    gather_ms = group.allreduce_max((time.perf_counter() - t0)*1e3)
    elapsed = group.allreduce_max(elapsed)
    per_rank = group.gather(ev_ms/args.steps)
    if rank != 0:
        return None
    tot = int(counts.sum())
    return {
        "gather_pipelined_ms": exposed[0] if exposed else None,
        "gather_exposed_ms": exposed[1] if exposed else None,
        "gather_chunks": job.chunks,
        "workload": "BASELINE configs[4]: double-Gauss, %d rays in total "
                    "over %d GPUs (%d per GPU), built on the device, RCCL "
                    "gather of y[L-1] to rank 0 after the last step inside "
                    "the timed region" % (tot, world, int(counts[0])),
        "total_rays": tot,
        "rays_per_gpu": int(counts[0]),
        "ms_per_step": elapsed*1e3/args.steps,
        "value": tot*S*args.steps/elapsed,
        "gather_ms": gather_ms,
        "kernel_ms_per_rank": per_rank,
    }



# --------------------------------------------------------------------------
# the driver's line: the contract's keys and one short record per leg; the
# full records go to a side file the line names
# --------------------------------------------------------------------------

CORE_LINE_LIMIT = 8192      # bytes; tests/test_bench_contract.py asserts it


def _r(v, digits=5):
    """Floats of the sub-records to `digits` significant digits."""
    if isinstance(v, float):
        return float("%.*g" % (digits, v)) if np.isfinite(v) else None
    if isinstance(v, (np.floating, np.integer, np.bool_)):
        return _r(v.item(), digits)
    if isinstance(v, dict):
        return {k: _r(x, digits) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_r(x, digits) for x in v]
    return v


def _pick(rec, *keys):
    return {k: rec[k] for k in keys if isinstance(rec, dict) and k in rec}


def _parity_ok(par, default_arith):
    """One boolean per config: bit identity with the C oracle (exact
    arithmetic) or <= 1e-8 with equal NaN masks (default asphere)."""
    if not par:
        return None
    if default_arith:
        return bool(par["nan_masks_equal"] and par["max_rel_err"] <= 1e-8)
    return bool(par["bit_identical_to_c_oracle"])


def write_detail(out, path=None):
    """The full records as one JSON file; returns the path written (relative
    to the repository where it lies inside it) or None."""
    path = path or os.environ.get("RT_BENCH_DETAIL") or os.path.join(
        ROOT, "gpurun_out", "bench_detail.json")
    for p in (path, os.path.join("/tmp", "rt_bench_detail_%d.json"
                                 % os.getpid())):
        try:
            os.makedirs(os.path.dirname(p), exist_ok=True)
            with open(p, "w") as f:
                json.dump(out, f, allow_nan=False)
            return os.path.relpath(p, ROOT) if p.startswith(ROOT + os.sep) \
                else p
        except (OSError, ValueError) as err:
            log("[bench] detail file %s: %r" % (p, err))
    return None


def core_line(out, detail):
    """What the driver reads: the contract's keys, `roofline`, `cpu_baseline`
    and per leg only what a reader needs to recompute its fraction.  Placement
    records, telemetry samples, VALU detail, wall-clock laps and the prose live
    in the detail file (``detail``)."""
    core = _pick(out, "metric", "value", "unit", "n_gpus", "steps", "warmup",
                 "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data", "config")
    r = out["roofline"]
    core["roofline"] = _r(_pick(
        r, "bound", "achieved", "peak", "unit", "frac", "traffic", "kernel",
        "kernel_ms", "algorithmic_bytes_per_launch",
        "bytes_per_ray_surface_op", "input_bytes_per_ray",
        "frac_if_80B_per_op_and_48B_per_ray_were_moved",
        "frac_at_observed_hbm_clock"), 7)
    core["roofline"].update(peak=r["peak"], frac=r["frac"],
                            achieved=r["achieved"],
                            kernel_ms=r["kernel_ms"])   # (unrounded)
    src = r.get("traffic_source") or ""
    core["roofline"]["traffic_source"] = (
        "rocprofv3 --pmc in this run" if src.startswith("measured in this")
        else "profiles/traffic.json" if src else None)
    pl = r.get("placement") or {}
    core["roofline"]["placement"] = _r(_pick(
        pl, "pieces", "per_class", "fast", "store_pattern_GBps",
        "workgroups_per_cu_cap"))
    if "valu" in r:
        core["roofline"]["valu_per_ray_surface_op"] = _r(
            r["valu"]["per_ray_surface_op"])
    if "full_i" in out:
        # SURVEY 8(d)'s own byte count: all of y, u, i, t stored (80 B/op)
        core["full_i"] = _r(_pick(out["full_i"], "kernel_ms",
                                  "algorithmic_bytes_per_launch", "achieved",
                                  "frac"), 7)
        core["full_i"]["bytes_per_ray_surface_op"] = 80.
    if "unclipped" in out:
        core["unclipped"] = _r(_pick(out["unclipped"], "kernel_ms", "frac"))
    for key in ("cpu_baseline", "cpu_baseline_all_cores", "cpu_baseline_c"):
        c = out.get(key)
        if c is None:
            continue
        core[key] = _r(_pick(c, "value", "unit", "cores", "kind", "error",
                             "image_row_bit_identical_to_gpu"))
        if key == "cpu_baseline":
            core[key]["sample"] = c.get("sample", "")[:220]
            core[key]["host"] = c.get("host")
    gb = out.get("generated_batch")
    if gb is not None:
        core["generated_batch"] = _r(_pick(
            gb, "rays", "kernel_ms", "algorithmic_bytes_per_launch", "frac",
            "traffic"))
    io = out.get("image_row_only")
    if io is not None:
        core["image_row_only"] = _r(_pick(io, "kernel_ms", "bound"))
        if "valu" in io:
            core["image_row_only"].update(_r(_pick(
                io["valu"], "valu_issue_frac", "valu_busy_frac")))
    api = out.get("propagate_api")
    if api is not None:
        core["propagate_api"] = _r(_pick(api, "engine_trace_ms_per_step",
                                         "propagate_ms_per_step"))
    cfgs = out.get("configs")
    if isinstance(cfgs, list):
        core["configs"] = []
        for c in cfgs:
            rec = _r(_pick(c, "rays", "surfaces", "kernel_ms",
                           "algorithmic_bytes_per_launch", "frac", "bound"))
            rec["config"] = c["config"].split(",")[0].split(":")[0][:40]
            if "exact_asphere" in c["config"]:
                rec["config"] += " exact"
            if "per-ray launch" in c["config"]:
                rec["config"] += " per-ray directions"
            if "parity_subsample" in c:
                rec["parity_ok"] = _parity_ok(
                    c["parity_subsample"], "default (" in c["config"])
            if "valu" in c:
                rec.update(_r(_pick(c["valu"], "valu_issue_frac",
                                    "valu_busy_frac"), 4))
            for k in ("newton_lane_utilisation",):
                if k in c:
                    rec[k] = _r(c[k], 4)
            sp = (c.get("placement") or {}).get("store_pattern_GBps")
            if sp:
                rec["store_pattern_GBps"] = _r(sp, 4)
            core["configs"].append(rec)
    elif cfgs is not None:
        core["configs"] = cfgs
    e = out.get("end_to_end")
    if e is not None:
        core["end_to_end"] = _r(_pick(
            e, "rays", "h2d_ms", "trace_ms", "d2h_image_xy_ms",
            "d2h_xy_bytes", "d2h_image_row_ms", "d2h_bytes", "end_to_end_ms",
            "end_to_end_full_row_ms", "h2d_fraction_of_ceiling",
            "d2h_fraction_of_ceiling", "d2h_xy_fraction_of_ceiling",
            "error"))
    cons = out.get("consumers")
    if isinstance(cons, list):
        core["consumers"] = []
        for c in cons:
            rec = _r(_pick(c, "ms", "kernel_ms", "bytes_read", "frac",
                           "kernel_frac", "fields_per_s", "replaces_ms",
                           "error"), 4)
            rec["call"] = c["call"].split(" ")[0].rstrip(",")
            core["consumers"].append(rec)
    elif cons is not None:
        core["consumers"] = cons
    t = (out.get("telemetry") or {}).get("loop") or {}
    if t:
        core["telemetry"] = _r({
            "gfxclk_mhz": (t.get("gfxclk_mhz") or [None]*3)[1],
            "hbm_uclk_mhz": (t.get("hbm_uclk_mhz") or [None]*3)[1],
            "socket_power_w": (t.get("socket_power_w") or [None]*3)[1],
            "power_limited_fraction": t.get("power_limited_fraction")}, 4)
    for key in ("gather_ms", "transport", "kernel_ms_per_rank",
                "gather_pipelined_ms", "gather_exposed_ms", "gather_chunks",
                "plain_loop_ms_per_step", "exchange_cost_ratio", "test_mode",
                "note", "configs4"):
        if key in out:
            core[key] = _r(out[key], 6)
    laps = (out.get("wall_s") or {}).get("since_start")
    if laps:
        core["wall_s"] = laps[-1][1]
    core["detail"] = detail
    return core


def leg_summaries(out):
    """One line per leg on stderr -- the headline included -- so that the
    driver's tail of the run carries them whatever happens to stdout."""
    r = out["roofline"]
    log("[summary] headline %s: %.4f ms/step, kernel %.4f ms, %.4g %s, "
        "frac %.4f of %g GB/s (%.0f B/op + %.2f B/ray)" % (
            out["config"]["workload"].split(":")[0], out["ms_per_step"],
            r["kernel_ms"], out["value"], out["unit"], r["frac"], r["peak"],
            r["bytes_per_ray_surface_op"], r["input_bytes_per_ray"]))
    for key in ("full_i", "unclipped", "generated_batch", "image_row_only"):
        c = out.get(key)
        if c:
            log("[summary] %s: kernel %.4f ms%s" % (
                key, c["kernel_ms"],
                ", frac %.4f" % c["frac"] if "frac" in c else ""))
    for c in (out.get("configs") if isinstance(out.get("configs"), list)
              else []):
        pl = c.get("placement") or {}
        log("[summary] %s: %.4f ms, frac %.4f, bound %s, store pattern per "
            "piece set %s" % (c["config"][:60], c["kernel_ms"],
                              c.get("frac", float("nan")), c.get("bound"),
                              pl.get("store_pattern_GBps_per_piece_set")))
    e = out.get("end_to_end") or {}
    if "end_to_end_ms" in e:
        log("[summary] end_to_end: h2d %.2f + trace %.3f + d2h (x, y of the "
            "image row) %.2f = %.2f ms; with all of the row (%.2f) %.2f ms"
            % (e["h2d_ms"], e["trace_ms"], e["d2h_image_xy_ms"],
               e["end_to_end_ms"], e["d2h_image_row_ms"],
               e["end_to_end_full_row_ms"]))
    for c in (out.get("consumers") if isinstance(out.get("consumers"), list)
              else []):
        if "ms" in c:
            log("[summary] consumer %s: %.4f ms (kernel %s), frac %s" % (
                c["call"].split(" ")[0], c["ms"], c.get("kernel_ms"),
                c.get("frac")))
    c = out.get("cpu_baseline")
    if c:
        log("[summary] cpu_baseline (%s, %d core): %.4g %s" % (
            c.get("kind"), c.get("cores", 0), c.get("value", float("nan")),
            c.get("unit")))
