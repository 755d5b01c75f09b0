#!/usr/bin/env python
"""Headline benchmark: ray-surface-ops/s of GeometricTrace.propagate().

    python bench.py --gpus N --steps K --warmup W

A "step" is one call of the public ``GeometricTrace.propagate(clip=True)`` --
re-packing the System, handing the table over and one fused pass of the hot
path over all S = len(system)-1 elements -- on one batch of synthetic rays
that is already resident in HBM.  Workload at every N: BASELINE.json
configs[2] -- the double-Gauss (L=13, S=12, spherical + stop), 10^7 rays per
GPU in five field bundles, clip=True (weak scaling: each rank traces its own
10^7-ray shard, different seeds).

N > 1 is one process per GPU.  ``python bench.py --gpus N`` starts the N
workers itself; started by a per-GPU launcher (``python -m
torch.distributed.run --nproc-per-node N ... bench.py --gpus N``: RANK /
LOCAL_RANK / WORLD_SIZE in the environment) it uses the ranks it is given.
Either way the host side is PyTorch-free: rendezvous, barrier and the
max-over-ranks of the timing go over rayopt_amd.distributed.HostGroup (TCP on
127.0.0.1), the device exchange is the engine's own RCCL gather.  The trace
needs no communication; the one exchange of the job -- the RCCL gather of the
last-surface intercepts y[L-1] of all ranks to rank 0 over xGMI -- runs once,
after the last step, INSIDE the timed region (results otherwise stay sharded
in HBM exactly as they stay in HBM at N=1).  ``gather_ms`` reports it alone;
--gather-every-step makes every step a complete job (trace + gather,
pipelined), which is bound by the root's xGMI ingest (24 B/ray over <= 7
links), not by the engine.  For N > 1 the line also carries ``configs4``:
BASELINE configs[4], 10^8 rays in total sharded over the N GPUs (1.25*10^7
per GPU at N=8), rays built on the device, same timed-loop rules.

Setup (untimed, before the W warm-up steps): rays are generated and uploaded
and the kernel is launched for --settle seconds (default 0.3 s) so the device
reaches its sustained clocks -- short runs otherwise measure the clock ramp
(first launches ~15 % slower).  Then exactly W untimed and K timed steps.

Rank 0 prints ONE JSON line.  Besides the contract's keys it carries

  roofline      achieved/peak HBM GB/s of the trace kernel; achieved =
                algorithmic bytes per launch / average launch duration from
                HIP events on the kernel's own stream.  Algorithmic bytes:
                48 B read per ray + per ray-surface op 56 B written (y 24,
                u 24, t 8) + 24 B for i where it has to be materialised.
                i[j] is bit-identical to u[j-1] unless element j or j-1 is
                tilted (rayopt/system.py:461,464), so by default the engine
                serves those rows of `i` from `u` instead of writing them
                again; the double-Gauss has no tilted element -> 56 B.
                ``traffic`` = HBM bytes per launch from PMC counters,
                measured in this run at N = 1 (two rocprofv3 --pmc passes of
                a 3-launch child run of this command; --traffic), or, where
                rocprofv3 cannot run, taken from the committed profile;
                ``traffic_source`` says which.
  propagate_api the public call against the bare engine call (Engine.trace
                in the same timed loop); with --extras also the wall time of
                one propagate() on a 10^4-ray batch, where the host path
                decides
  generated_batch   the same bundles built on the device (rays_fields): a
                re-trace rebuilds its launch rays in registers instead of
                reading row 0 -- own timed loop, value / kernel_ms / achieved
  full_i / unclipped / image_row_only   (--extras) other store modes
  cpu_baseline  rayopt's own GeometricTrace.propagate()
                (rayopt/geometric_trace.py:72-80, imported unmodified from
                oracle/_ref/, which oracle/make_ref.py packs in the build
                container and which travels with the snapshot) timed on this
                host, ONE process, on a bounded sample of the same workload:
                kind "reference"; the numpy port (oracle/trace_numpy.py) is
                timed beside it (``port_value``) and stands in (kind "port")
                only where the reference archive is missing
  cpu_baseline_all_cores   the same on every host core (one forked process
                per core over contiguous ray shards)
  configs       every BASELINE config on this GPU, one record each: C1
                (singlet 10^4), C2 (Cooke 10^6 rays x 3 wavelengths, ONE
                launch), C3 (= the headline), C4 (asphere phone lens 10^7
                rays: default arithmetic and exact_asphere), C5 on one GPU
                (double-Gauss 10^8 rays built on the device) -- kernel_ms,
                algorithmic bytes, frac, parity of a 10^5-ray subsample
                against the C oracle, and the reference's own rate on a
                small sample of the same config
  telemetry     gfx clock, HBM clock, socket power, temperatures sampled by
                a child process (amdsmi) around the timed loop, and ``frac``
                against the HBM clock actually observed
  cpu_baseline_c the independent plain-C port (oracle/trace_c.c) with OpenMP:
                the compiled multi-threaded CPU figure, as a range over team
                sizes (boxes of the pool differ by x1.8)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
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


def cpu_c_oracle(table, y, u, clip, S, g, L, sample=2_000_000):
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
    teams = sorted({min(os.cpu_count(), t) for t in (16, 64, os.cpu_count())})
    out = build_c.propagate(table, ys, us, clip=clip)   # touch output pages
    by_team = {}
    for team in (teams if gomp is not None else teams[-1:]):
        if gomp is not None:
            gomp.omp_set_num_threads(team)
        best = None
        for _ in range(3):
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
        "sample": "first %d rays, best of 3 propagate() of the C port with "
                  "OpenMP per team size, same output arrays; host has %d "
                  "cores; boxes of the pool differ by up to x1.8 on this "
                  "figure -- read it as a range" % (m, os.cpu_count()),
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
                [sys.executable, os.path.abspath(__file__),
                 "--telemetry-child", str(device), str(period)],
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
    for this workload (profiles/traffic.json, written by
    scripts/pmc_traffic.py on the GPU box); otherwise null."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def _profiled_from_outside(env):
    """True when this process already runs under a profiler (rocprofv3 / the
    rocprofiler-sdk tool library): a counter session nested inside another
    one is not attempted."""
    keys = ("ROCP_TOOL_LIBRARIES", "ROCPROF_OUTPUT_PATH",
            "ROCPROFILER_REGISTER_FORCE_LOAD", "ROCPROF_KERNEL_TRACE")
    return any(env.get(k) for k in keys) or \
        "rocprofiler-sdk-tool" in env.get("LD_PRELOAD", "")


def traffic_live(n, clip, timeout=120.):
    """HBM bytes per launch of rt_trace_kernel on THIS box, now: two
    `rocprofv3 --kernel-trace --pmc <counter>` passes (FETCH_SIZE and
    WRITE_SIZE do not fit one pass: MI355X_MICROARCH.md, TCC budget) of a
    short run of this very command -- same workload, same kernel, 3 launches
    -- with the guide's gfx950 correction (FETCH_SIZE tallies 128-B requests
    at 64 B: doubled; both counters are KiB).  Returns (bytes, detail) or
    raises; the caller falls back to the committed profile."""
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
    env.update(TMPDIR="/tmp", RT_BENCH_CHILD="1")
    work = tempfile.mkdtemp(prefix="rt_bench_pmc_", dir="/tmp")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--rays", str(n),
           "--steps", "2", "--warmup", "1", "--settle", "0", "--cpu-sample",
           "0", "--cpu-procs", "0", "--no-engine-leg", "--traffic", "off"]
    if not clip:
        cmd.append("--no-clip")
    kib, launches, gen = {}, {}, {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(work, counter)
            res = subprocess.run(
                [exe, "--kernel-trace", "--pmc", counter, "--output-format",
                 "csv", "-d", out, "--"] + cmd, cwd="/tmp", env=env,
                stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                timeout=timeout)
            if res.returncode != 0:
                raise RuntimeError("rocprofv3 --pmc %s: rc %d: %s" % (
                    counter, res.returncode, res.stderr[-300:]))
            vals, regen = [], []
            for path in glob.glob(os.path.join(
                    out, "**", "*counter_collection.csv"), recursive=True):
                with open(path) as f:
                    rows = [row for row in csv.DictReader(f)
                            if row.get("Counter_Name") == counter]
                rows.sort(key=lambda row: int(row.get("Dispatch_Id", 0)))
                for row in rows:
                    name = row.get("Kernel_Name", "")
                    if "rt_trace_kernel" in name:
                        vals.append(float(row["Counter_Value"]))
                    elif "rt_trace_gen_kernel" in name:
                        regen.append(float(row["Counter_Value"]))
            if not vals:
                raise RuntimeError("no %s rows for rt_trace_kernel" % counter)
            kib[counter] = sum(vals)/len(vals)
            launches[counter] = len(vals)
            # the generated batch: its first launch writes row 0 as well;
            # the re-traces are what the leg times
            if len(regen) > 1:
                gen[counter] = sum(regen[1:])/len(regen[1:])
    finally:
        shutil.rmtree(work, ignore_errors=True)
    fetch = kib["FETCH_SIZE"]*1024*2
    write = kib["WRITE_SIZE"]*1024
    detail = {"fetch_bytes_corrected_x2": fetch, "write_bytes": write,
              "launches": [launches["FETCH_SIZE"], launches["WRITE_SIZE"]]}
    if len(gen) == 2:
        detail["generated_batch"] = {
            "fetch_bytes_corrected_x2": gen["FETCH_SIZE"]*1024*2,
            "write_bytes": gen["WRITE_SIZE"]*1024,
            "hbm_bytes_per_launch": gen["FETCH_SIZE"]*1024*2 +
            gen["WRITE_SIZE"]*1024}
    return fetch + write, detail


# --------------------------------------------------------------------------
# timed loops
# --------------------------------------------------------------------------

class Job:
    """One rank's share of the benchmark: its trace, its engine and the host
    group it synchronises with."""

    def __init__(self, args, group, g, counts, d_dst):
        self.args, self.group, self.g = args, group, g
        self.eng = g.engine
        self.dist = group is not None
        self.counts, self.d_dst = counts, d_dst
        self.L = len(g.system)

    exchange = True
    chunks = 1

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


def main():
    # the contract is ONE JSON line on stdout: native libraries (RCCL prints a
    # version banner) must not get at it, so fd 1 is pointed at stderr for the
    # whole run and the line is written to the saved descriptor at the end
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rays", type=int, default=10_000_000,
                    help="rays per GPU")
    ap.add_argument("--total-rays", type=int, default=0,
                    help="rays of the whole job, sharded over the GPUs "
                         "(overrides --rays; 100000000 at --gpus 8 is "
                         "BASELINE configs[4])")
    ap.add_argument("--no-clip", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=3_000_000,
                    help="rays of the workload the reference is timed on, "
                         "one process (0: skip every CPU leg); the default "
                         "is ~4 s of rayopt on one core")
    ap.add_argument("--cpu-procs", type=int, default=-1,
                    help="processes of the many-cores leg of the reference "
                         "(forked before the GPU is touched); -1 = one per "
                         "host core but at most 64 (forking and warming 256 "
                         "interpreters costs more wall time than the "
                         "measurement), -2 = one per host core, 0 = skip")
    ap.add_argument("--settle", type=float, default=0.3,
                    help="seconds of untimed launches during setup so the "
                         "device reaches its sustained clocks (boxes of the "
                         "pool take ~50 launches); 0 disables")
    ap.add_argument("--extras", action="store_true",
                    help="also time the full_i (80 B/op), unclipped and "
                         "image-row-only modes (separate timed loops, "
                         "reported as extra objects)")
    ap.add_argument("--traffic", choices=("live", "profile", "off"),
                    default="live",
                    help="roofline.traffic: 'live' = two rocprofv3 --pmc "
                         "passes of a short run of this command on this box "
                         "(N = 1 only; falls back to 'profile' if rocprofv3 "
                         "cannot run), 'profile' = the committed "
                         "profiles/traffic.json, 'off' = null")
    ap.add_argument("--no-engine-leg", action="store_true",
                    help="skip the bare-engine comparison leg only")
    ap.add_argument("--no-api-leg", action="store_true",
                    help="skip the propagate_api comparison legs")
    ap.add_argument("--gather-every-step", action="store_true",
                    help="N>1: gather y[L-1] to rank 0 in every step")
    ap.add_argument("--no-configs4", action="store_true",
                    help="N>1: skip the BASELINE configs[4] leg (10^8 rays "
                         "in total)")
    ap.add_argument("--gather-chunks", type=int, default=4,
                    help="N>1: the last step is traced in this many pieces "
                         "and the gather of piece k overlaps the trace of "
                         "piece k+1 (1: trace, then gather)")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the per-config records (C1, C2, C4, C5 on one "
                         "GPU)")
    ap.add_argument("--no-configs5", action="store_true",
                    help="skip the 10^8-ray one-GPU batch (104 GB) of the "
                         "configs leg")
    ap.add_argument("--configs5-rays", type=int, default=0,
                    help="rays of that batch (default 10^8)")
    ap.add_argument("--option", action="append", default=[],
                    help="kernel variant key=value (rt_set_option)")
    args = ap.parse_args()

    from rayopt_amd import distributed as D
    launched = "WORLD_SIZE" in os.environ
    if not launched and args.gpus > 1:
        # plain `python bench.py --gpus N`: one worker per GPU, this process
        # only waits (rank 0's JSON line goes to the inherited stdout)
        raise SystemExit(D.spawn_workers(
            args.gpus,
            check_devices=not os.environ.get("RT_BENCH_SHARE_DEVICE")))
    world, rank, local_rank = D.world_info()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # RT_BENCH_SHARE_DEVICE=1 (tests only): every rank opens device 0 and the
    # RCCL exchange is left out -- RCCL refuses two ranks on one device -- so
    # that the host side of the N>1 path (spawn, host group, per-rank
    # bookkeeping, the configs[4] leg) can run on a one-GPU box.  The line
    # it prints is marked "test_mode" and is not a measurement.
    share = bool(os.environ.get("RT_BENCH_SHARE_DEVICE"))
    # ... unless RT_TRANSPORT_LIBRARY names a stand-in for librccl.so (the
    # shared-memory transport of tests/stubs): then the engine's gather runs
    # as it is, nranks > 1 branch included, and rank 0 checks every shard
    stand_in = os.environ.get("RT_TRANSPORT_LIBRARY", "")
    if share:
        local_rank = 0

    def check_device():
        # opens the HIP runtime: only after the forked CPU leg (N = 1)
        nonlocal local_rank
        have = D.visible_devices()
        if have == 1 and world > 1 and local_rank and any(
                os.environ.get(k) for k in ("HIP_VISIBLE_DEVICES",
                                            "ROCR_VISIBLE_DEVICES",
                                            "CUDA_VISIBLE_DEVICES")):
            # the launcher masks the devices per rank: ours is device 0 (two
            # ranks that really share one GPU are refused by RCCL below)
            local_rank = 0
        if local_rank >= have:
            raise SystemExit("--gpus %d: %d devices needed, %d visible"
                             % (args.gpus, max(world, local_rank + 1), have))
    if world > 1:
        check_device()

    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(line):
        sys.stdout.flush()
        os.write(real_stdout, (line + "\n").encode())

    # RT_BENCH_FORCE_DIST=1 exercises the whole multi-process path (host
    # group, RCCL communicator, pipelined gather) with a single rank
    dist_mode = world > 1 or bool(os.environ.get("RT_BENCH_FORCE_DIST"))
    group = D.HostGroup(world, rank) if dist_mode else None

    import rayopt_amd as ra
    from rayopt_amd import prescriptions as P
    from rayopt_amd.pack import pack_system

    clip = not args.no_clip
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    L = len(system)
    S = L - 1
    if args.total_rays:
        counts = D.shard_counts(args.total_rays, world)
    else:
        counts = np.full(world, args.rays, dtype=np.int64)
    n = int(counts[rank])

    lap("imports, system")
    t0 = time.perf_counter()
    y, u = workload_rays(n, rank)
    lap("workload rays on the host")
    cpu_all = None
    procs = (min(64, os.cpu_count()) if args.cpu_procs == -1 else
             os.cpu_count() if args.cpu_procs < 0 else args.cpu_procs)
    if procs > 1 and args.cpu_sample > 0 and rank == 0 and not dist_mode:
        try:            # forks: before this process opens the GPU
            from oracle import refshim
            if refshim.available():
                cpu_all = reference_on_processes(
                    P.DOUBLE_GAUSS, y, u, system.wavelengths[0], clip, procs)
                if args.extras:
                    cpu_all["port_value"] = cpu_port_on_processes(
                        system, y, u, clip, procs)["value"]
            else:
                cpu_all = cpu_port_on_processes(system, y, u, clip, procs)
        except Exception as err:      # a reported extra, never fatal
            cpu_all = {"error": repr(err)[:200]}
    lap("reference on %d forked processes" % procs if cpu_all else
        "(no many-process CPU leg)")
    if world == 1:
        check_device()
    g = ra.GeometricTrace(system, device=local_rank)
    eng = g.engine
    for kv in args.option:
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    g.rays_given(y, u)          # rays resident in HBM from here on
    log("[rank %d] %d rays generated + uploaded in %.2f s" % (
        rank, n, time.perf_counter() - t0))
    lap("context, arrays placed, rays uploaded")

    # RCCL communicator for the gather of the final intercepts (only where
    # there is an exchange); no fallback: without it the job fails
    d_dst = 0
    if dist_mode:
        if not share or stand_in:
            D.init_engine_comm(eng, group)
        if rank == 0:
            d_dst = eng.scratch(int(counts.sum())*3*8)
    job = Job(args, group, g, counts, d_dst)
    job.exchange = not share or bool(stand_in)
    # what the transport itself says it is (not what the launcher claimed):
    # ranks in the communicator, RCCL version, link type to every peer device
    comm_info = None
    if dist_mode and job.exchange:
        mine = eng.comm_info()
        assert mine["nranks_seen"] == world and mine["rank_seen"] == rank, mine
        every = group.gather(np.array([mine["nranks_seen"],
                                       mine["rank_seen"]], dtype=np.int64))
        if rank == 0:
            comm_info = dict(mine, every_rank_saw=[e.tolist() for e in every])

    mode = {"clip": clip}

    def step():                 # the public call
        g.propagate(clip=mode["clip"])
        if dist_mode and args.gather_every_step:
            job.gather()

    def step_engine():          # the bare C-ABI call, table already there
        eng.trace(1, 0, mode["clip"])

    # clocks / power / limiter residency: a child process that samples amdsmi
    # from before the settle phase on (its start-up and first queries are
    # over long before the timed loop)
    tele = Telemetry(local_rank, period=0.01) if (
        rank == 0 and not os.environ.get("RT_BENCH_CHILD")) else None
    if tele is not None:
        tele.mark("settle:begin")
    settle(g, args.settle, clip)    # setup, not part of W or K
    lap("settle")
    if tele is not None:
        tele.mark("settle:end")

    final_gather = dist_mode and not args.gather_every_step
    plain = not dist_mode and not args.option

    image_only = unclipped = full_i = engine_leg = None
    if plain and (args.extras or (
            world == 1 and not args.no_configs and
            not os.environ.get("RT_BENCH_CHILD") and
            not _profiled_from_outside(os.environ))):
        # extension: keep only the image row (merit-function use): the
        # kernel leaves the HBM roofline for the FP64 one.  (Not under a
        # profiler: see the note at the configs below.)
        def step_image():
            g.propagate(clip=clip, keep=[0, -1])
        job.timed(step_image, 300, 10, False)    # its own settled state
        if tele is not None:
            tele.mark("imgrow:begin")
        # (~0.5 s: amdsmi's clock is a moving average of about that length)
        e_img, ev_img, _ = job.timed(step_image, max(args.steps, 900), 0,
                                     False)
        if tele is not None:
            tele.mark("imgrow:end")
        image_only = (e_img/max(args.steps, 900)*args.steps,
                      ev_img/max(args.steps, 900))
    if args.extras and plain and clip:
        # the reference's default: propagate(clip=False); the u rows of the
        # elements that do not bend the ray are not written either
        mode["clip"] = False
        e_nc, ev_nc, _ = job.timed(step, args.steps, args.warmup, False)
        unclipped = (e_nc, ev_nc/args.steps)
        mode["clip"] = clip
    if args.extras and plain:
        # reference point: every row of `i` written (80 B per op)
        eng.set_option("alias_i", 0)
        e_full, ev_full, _ = job.timed(step, args.steps, args.warmup, False)
        full_i = (e_full, ev_full/args.steps)
        eng.set_option("alias_i", 1)
    if not args.no_api_leg and not args.no_engine_leg and not dist_mode:
        g.propagate(clip=clip)
        e_eng, ev_eng, _ = job.timed(step_engine, args.steps, args.warmup,
                                     False)
        engine_leg = (e_eng, ev_eng/args.steps)

    job.chunks = max(1, args.gather_chunks)

    def last_step():            # the K-th step, its gather pipelined with it
        g.propagate(clip=mode["clip"], chunks=job.chunks,
                    after_chunk=job.gather_chunk)

    plain_loop = None
    if dist_mode and final_gather:
        # the same K steps WITHOUT the exchange, same process: what the
        # one exposed gather costs the job, and how this multi-process line
        # relates to the plain N = 1 one
        e_plain, _, _ = job.timed(step, args.steps, args.warmup, False)
        plain_loop = group.allreduce_max(e_plain)
    if tele is not None:
        tele.mark("loop:begin")
    elapsed, ev_ms, last_kernel_ms = job.timed(
        step, args.steps, args.warmup, final_gather,
        last_step if (final_gather and job.chunks > 1 and job.exchange)
        else None)
    if tele is not None:
        tele.mark("loop:end")
    lap("image-row / engine legs, warm-up and the timed loop")
    # where the result arrays live: pieces of device memory in a measured mix
    # of memory classes (rt_placement); store-bound traces then run four
    # workgroups per CU from the first launch on, two otherwise
    placement = eng.placement()
    placement["workgroups_per_cu_cap"] = 4 if placement["fast"] else 2
    placement["note"] = (
        "the speed of the trace's simultaneous row streams is a property of "
        "the physical memory behind the arrays (bare store pattern: 7.0 / "
        "6.3 / 5.65 TB/s); arrays > 1.5 GiB are built from pieces whose "
        "class is measured at allocation (~1 ms each) and mixed "
        "(csrc/rt_place.h); results do not depend on it")
    gather_ms = gather_exposed = None
    if dist_mode:
        if final_gather and job.exchange:
            # HIP events: the whole exchange, and what was left of it after
            # this rank's last trace kernel had finished
            tot, exp = eng.gather_ms()
            gather_exposed = [group.allreduce_max(tot),
                              group.allreduce_max(exp)]
        # the exchange alone, unpipelined (not part of the timed region)
        job.fence()
        t0 = time.perf_counter()
        job.gather()
        job.fence()
        gather_ms = group.allreduce_max((time.perf_counter() - t0)*1e3)
        elapsed = group.allreduce_max(elapsed)
    kernel_ms = (ev_ms/args.steps if not (dist_mode and args.gather_every_step)
                 else last_kernel_ms)
    per_rank_kernel_ms = group.gather(kernel_ms) if dist_mode else [kernel_ms]

    # sanity on the result of the last step (not timed): a few per cent of
    # the rays vignette, everything else reaches the image
    ylast = np.asarray(g.y[L - 1])
    ulast = np.asarray(g.u[L - 1])
    finite = float(np.isfinite(ulast[:, 0]).mean())
    if dist_mode and rank == 0 and job.exchange:
        gathered = eng.copy_to_host(d_dst, int(counts.sum())*3*8)
        gathered = gathered.reshape(3, -1)
        mine = gathered[:, :n].T
        assert np.array_equal(mine, ylast, equal_nan=True), \
            "gathered shard 0 differs from the local result"
        assert np.isfinite(gathered).mean() > 0.9
    if dist_mode and stand_in and job.exchange:
        # test mode: every rank's image row travels over the host group too
        # and rank 0 compares the whole gathered buffer with it
        rows = group.gather(ylast)
        if rank == 0:
            for have, want in zip(D.split_gathered(gathered, counts), rows):
                assert np.array_equal(have, want, equal_nan=True), \
                    "a gathered shard differs from its rank's result"
        del rows
    gathered = None

    api = None
    if engine_leg is not None and rank == 0:
        api = {"engine_trace_ms_per_step": engine_leg[0]*1e3/args.steps,
               "propagate_ms_per_step": elapsed*1e3/args.steps,
               "ratio": elapsed/engine_leg[0],
               "note": "`value` times the public GeometricTrace.propagate() "
                       "(re-pack + table hand-over + launch); "
                       "engine_trace = the bare rt_trace call in the same "
                       "timed loop"}
        if args.extras:     # launches of another batch size: kept out of
            # the default command so that every rt_trace_kernel launch a
            # profiler sees there is the headline workload
            api.update(small_batch_latency(ra, system, local_rank))

    generated = None
    if not dist_mode and plain and not args.no_api_leg:
        generated = run_generated(ra, system, local_rank, n, clip, args)
        lap("generated batch")

    configs4 = None
    if dist_mode and world > 1 and not args.no_configs4 and \
            not args.total_rays:
        del ylast, ulast, y, u
        configs4 = run_configs4(ra, system, g, job, group, world, rank, args,
                                clip)

    if rank != 0:
        group.barrier()
        group.close()
        return

    table, ns = pack_system(system, g.l, g.n[0])
    total_rays = int(counts.sum())
    ms_per_step = elapsed*1e3/args.steps
    value = total_rays*S*args.steps/elapsed
    from rayopt_amd._lib import F_ROTATED, F_REFRACT
    rot = (table["flags"] & F_ROTATED) != 0
    bends = (table["flags"] & F_REFRACT) != 0
    alias_on = not any(kv == "alias_i=0" for kv in args.option)
    stored_i = sum(1 for j in range(1, L)
                   if not alias_on or rot[j] or rot[j - 1])
    # an unclipped trace does not write u[j] where the element does not bend
    # the ray (u[j] is i[j] bit for bit: stop, image)
    skipped_u = sum(1 for j in range(1, L)
                    if alias_on and not clip and not bends[j])
    # what the launch rows cost: 48 B per ray, less where the seed kernel
    # found components uniform across 64-ray tiles (this workload: five
    # collimated bundles starting on a plane -> y2, u0, u1, u2)
    read_bytes, uniform_share = input_bytes(eng, n)
    alg_bytes = n*(56*S + 24*stored_i - 24*skipped_u) + read_bytes  # one shard
    achieved = alg_bytes/(kernel_ms*1e-3)/1e9
    traffic = traffic_source = traffic_detail = None
    plain_kernel = alias_on and not args.option
    if args.traffic == "live" and world == 1 and not dist_mode and \
            plain_kernel and not os.environ.get("RT_BENCH_CHILD") and \
            not _profiled_from_outside(os.environ):
        try:
            t0 = time.perf_counter()
            traffic, traffic_detail = traffic_live(n, clip)
            lap("live traffic (two rocprofv3 --pmc child runs)")
            traffic_source = (
                "measured in this run: rocprofv3 --kernel-trace --pmc "
                "FETCH_SIZE / WRITE_SIZE (separate passes, %d + %d launches "
                "of this workload in a child process, %.0f s), FETCH_SIZE "
                "x2 per MI355X_MICROARCH.md (gfx950)" % (
                    traffic_detail["launches"][0],
                    traffic_detail["launches"][1],
                    time.perf_counter() - t0))
        except Exception as err:
            log("[bench] live traffic measurement failed: %r" % (err,))
            traffic = None
    prof = traffic_from_profile() if (traffic is None and
                                      args.traffic != "off") else None
    if prof and prof.get("rays") == n and prof.get("clip") == clip and \
            prof.get("alias_i", 0) == int(alias_on):
        traffic = prof.get("hbm_bytes_per_launch")
        traffic_source = ("profiles/traffic.json (rocprofv3 --pmc passes of "
                          "this command on the GPU box, %s; not re-measured "
                          "in this run)" % prof.get("profile", "committed"))

    par = "ray shards x%d" % world
    if dist_mode:
        par += (", one process per GPU, host group over TCP (no PyTorch), "
                "RCCL gather of y[L-1] to rank 0 %s (gather alone: %.2f ms)"
                % ("in every step" if args.gather_every_step else
                   "once, after the last step, inside the timed region",
                   gather_ms))
    out = {
        "metric": "ray-surface-ops/sec",
        "value": value,
        "unit": "ray-surface-ops/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": "double-Gauss (BASELINE configs[%d]): L=%d elements, "
                        "S=%d propagated surfaces, %d rays/GPU in 5 field "
                        "bundles, clip=%s, one step = one "
                        "GeometricTrace.propagate()" % (
                            4 if args.total_rays == 10**8 else 2, L, S, n,
                            clip),
            "rays_per_gpu": n,
            "total_rays": total_rays,
            "surfaces": S,
            "clip": clip,
            "finite_fraction_at_image": finite,
            "settle_s": args.settle,
            "parallelism": par,
        },
        "roofline": {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved/HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_source": traffic_source,
            "traffic_detail": traffic_detail,
            "kernel": "rt_trace_kernel",
            "kernel_ms": kernel_ms,
            "algorithmic_bytes_per_launch": alg_bytes,
            "bytes_per_ray_surface_op": (56*S + 24*stored_i - 24*skipped_u)/S,
            "input_bytes_per_ray": read_bytes/n,
            "input_uniform_share_y0y1y2u0u1u2": uniform_share,
            # for comparison with lines written before the tile notes
            # (rounds 1-2 counted 48 B per ray read): NOT what is moved
            "frac_if_48B_per_ray_were_read":
                (alg_bytes - read_bytes + 48*n)/(kernel_ms*1e-3)/1e9 /
                HBM_PEAK_GBS,
            "frac_of_achievable_6290": achieved/HBM_ACHIEVABLE_GBS,
            "placement": placement,
        },
    }
    if dist_mode:
        out["gather_ms"] = gather_ms
        out["transport"] = comm_info
        out["kernel_ms_per_rank"] = per_rank_kernel_ms
        if gather_exposed is not None:
            out["gather_pipelined_ms"] = gather_exposed[0]
            out["gather_exposed_ms"] = gather_exposed[1]
            out["gather_chunks"] = job.chunks
        if plain_loop is not None:
            out["plain_loop_ms_per_step"] = plain_loop*1e3/args.steps
            out["exchange_cost_ratio"] = elapsed/plain_loop
            out["note"] = (
                "`value` includes the job's one exchange (RCCL gather of "
                "y[L-1] to rank 0, the last step traced in %d chunks so that "
                "all but the last chunk's gather overlaps tracing): "
                "ms_per_step x steps = plain_loop_ms_per_step x steps + the "
                "exposed part of the gather.  The plain N = 1 line (no host "
                "group, no exchange) is the plain loop of this line; "
                "processes of one box differ by up to +-4 %% "
                "(profiles/README.md)" % job.chunks)
        if share:
            out["test_mode"] = ("RT_BENCH_SHARE_DEVICE: all ranks on device "
                                "0, %s -- not a measurement" % (
                                    "rt_gather_final run over the stand-in "
                                    "transport %s, every gathered shard "
                                    "checked" % os.path.basename(stand_in)
                                    if stand_in else
                                    "RCCL exchange left out"))
        elif stand_in:
            out["test_mode"] = ("RT_TRANSPORT_LIBRARY=%s replaces RCCL -- "
                                "not a measurement" % stand_in)
    if configs4 is not None:
        out["configs4"] = configs4
    if generated is not None:
        counted = (traffic_detail or {}).pop("generated_batch", None)
        if counted:     # the same counter passes saw this leg's kernel too
            generated["traffic"] = counted["hbm_bytes_per_launch"]
            generated["traffic_detail"] = {
                k: counted[k] for k in ("fetch_bytes_corrected_x2",
                                        "write_bytes")}
        out["generated_batch"] = generated
    if api is not None:
        out["propagate_api"] = api

    if full_i is not None:
        e_full, k_full = full_i
        b_full = n*80*S + read_bytes
        out["full_i"] = {
            "value": total_rays*S*args.steps/e_full,
            "kernel_ms": k_full,
            "algorithmic_bytes_per_launch": b_full,
            "achieved": b_full/(k_full*1e-3)/1e9,
            "frac": b_full/(k_full*1e-3)/1e9/HBM_PEAK_GBS,
            "note": "every row of i materialised (alias_i=0): 80 B per op",
        }
    if unclipped is not None:
        e_nc, k_nc = unclipped
        b_nc = read_bytes + n*(56*S + 24*stored_i - 24*sum(
            1 for j in range(1, L) if alias_on and not bends[j]))
        out["unclipped"] = {
            "value": total_rays*S*args.steps/e_nc,
            "kernel_ms": k_nc,
            "algorithmic_bytes_per_launch": b_nc,
            "achieved": b_nc/(k_nc*1e-3)/1e9,
            "frac": b_nc/(k_nc*1e-3)/1e9/HBM_PEAK_GBS,
            "note": "propagate(clip=False), the reference's default: u rows "
                    "of stop and image are i rows bit for bit and are not "
                    "written",
        }
    if image_only is not None:
        e_img, k_img = image_only
        out["image_row_only"] = {
            "value": total_rays*S*args.steps/e_img,
            "kernel_ms": k_img,
            "note": "propagate(keep=[0, -1]): all %d surfaces traced, only "
                    "the image row stored (80 B/ray); FP64-VALU bound" % S,
        }

    if tele is not None:
        t = tele.stop() or {"samples": 0,
                            "error": "the telemetry child did not start"}
        if t:
            loop = t.get("loop") or {}
            uclk = (loop.get("hbm_uclk_mhz") or [None, None, None])[1]
            if uclk:
                out["roofline"]["hbm_uclk_mhz_observed"] = uclk
                out["roofline"]["frac_at_observed_hbm_clock"] = \
                    achieved/(HBM_PEAK_GBS*uclk/HBM_NOMINAL_MHZ)
            out["telemetry"] = t
            w = t.get("imgrow") or {}
            clk = (w.get("gfxclk_mhz") or [None]*3)[1]
            try:
                with open(os.path.join(ROOT, "profiles",
                                       "valu_counters.json")) as f:
                    c = json.load(f)["kinds"]["C3 image row only"]
            except (OSError, ValueError, KeyError):
                c = None
            if c and clk and "image_row_only" in out:
                io = out["image_row_only"]
                cyc = 1024*clk*1e6*io["kernel_ms"]*1e-3
                sc = n/c["rays"]
                io.update(
                    bound="fp64 valu issue", gfxclk_mhz_observed=clk,
                    power_limited_fraction=w.get("power_limited_fraction"),
                    valu_wave_instructions_per_launch=c["SQ_INSTS_VALU"]*sc,
                    valu_issue_frac=c["SQ_INSTS_VALU"]*sc*4/cyc,
                    valu_busy_frac=c["SQ_ACTIVE_INST_VALU"]*sc*4/cyc,
                    flop_equivalents_per_s=io["value"]*190.,
                    flop_equivalent_note="SURVEY 8(d): not an HBM leg (~190 "
                    "flop-equivalents per ray-surface op: 70 flop + 3 sqrt + "
                    "4 div); the "
                    "ceiling is 1024 SIMDs x gfx clock / 4 cycles per FP64 "
                    "wave-instruction (profiles/valu_counters.json)")
    # (not under a profiler: every rt_trace_kernel launch rocprofv3 sees in
    # this command is then the headline workload, so that its average can be
    # held against roofline.kernel_ms)
    if world == 1 and not dist_mode and plain and not args.no_configs and \
            not os.environ.get("RT_BENCH_CHILD") and \
            not _profiled_from_outside(os.environ):
        try:
            del ylast, ulast
            out["configs"] = [{
                "config": "C3 double-Gauss, %d rays in 5 field bundles "
                          "(the headline line above)" % n,
                "rays": n, "surfaces": S, "clip": clip,
                "kernel_ms": kernel_ms, "value": n*S/(kernel_ms*1e-3),
                "algorithmic_bytes_per_launch": alg_bytes,
                "achieved": achieved, "frac": achieved/HBM_PEAK_GBS}] + \
                run_configs(ra, local_rank, args)
        except Exception as err:      # reported extras, never fatal
            out["configs"] = {"error": repr(err)[:300]}
        lap("configs C1 C2 C3' C4 C4x C5 (with their parity subsamples)")
    if world == 1 and not dist_mode and args.cpu_sample > 0:
        out["cpu_baseline"] = cpu_one_core(table, system, y, u, clip, S, g, L,
                                           args.cpu_sample, g.l)
        if cpu_all is not None:
            out["cpu_baseline_all_cores"] = cpu_all
        try:
            out["cpu_baseline_c"] = cpu_c_oracle(table, y, u, clip, S, g, L)
        except Exception as err:      # a reported extra, never fatal
            out["cpu_baseline_c"] = {"error": repr(err)[:200]}
        lap("cpu_baseline: reference on one core, C port")
    if world == 1 and not dist_mode and plain and not args.no_configs and \
            not os.environ.get("RT_BENCH_CHILD") and \
            not _profiled_from_outside(os.environ):
        try:
            out["consumers"] = run_consumers(
                ra, g, system, n, len(FIELD_FRACTIONS),
                out.get("cpu_baseline"))
        except Exception as err:      # reported extras, never fatal
            out["consumers"] = {"error": repr(err)[:300]}
        lap("consumers")
    if rank == 0:
        out["wall_s"] = {"since_start": LAPS,
                         "note": "seconds since this interpreter reached "
                                 "bench.py, after each phase (stderr carries "
                                 "the same lines)"}
    emit(json.dumps(out))
    if dist_mode:
        group.barrier()
        group.close()


# --------------------------------------------------------------------------
# every BASELINE config on this GPU
# --------------------------------------------------------------------------

def input_bytes(eng, n):
    """Bytes of the launch rows (row 0 of Y and U) a trace from element 1 has
    to read: 8 per ray and component, except where a component is one bit
    pattern across a 64-ray tile -- the direction of a collimated bundle, z = 0
    of rays starting on a plane: noted by the seed kernel, fetched once per
    tile (8 B) -- plus the 4-byte note per tile."""
    uniform, tiles = eng.input_uniform()
    if not any(uniform):
        return 48*n, [0]*6
    return (sum(8*(n - 64*u) + 8*u for u in uniform) + 4*tiles,
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


def kernel_ms_of(g, clip, settle_s=.3, per_block=10, dwell_s=.5, mark=None):
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


def run_configs(ra, device, args):
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

    counters = {}
    try:
        with open(os.path.join(ROOT, "profiles", "valu_counters.json")) as f:
            counters = json.load(f)["kinds"]
    except (OSError, ValueError, KeyError):
        pass
    tele = Telemetry(device, period=0.01)
    windows = []

    def valu_roofline(rec, kind, n, ms, clock_mhz):
        """The FP64-issue ceiling: VALU wave-instructions per launch (PMC
        profile of the same workload, profiles/valu_counters.json, scaled to
        n rays) x 4 cycles / (1024 SIMDs x the gfx clock observed during
        THIS measurement x launch time).  `busy` uses SQ_ACTIVE_INST_VALU
        (quad-cycles the VALUs were executing, quarter-rate v_rcp / v_rsq
        included) instead of the instruction count."""
        c = counters.get(kind)
        if not c or not clock_mhz:
            return
        scale = n/c["rays"]
        cyc = 1024*clock_mhz*1e6*ms*1e-3
        rec["valu"] = {
            "wave_instructions_per_launch": c["SQ_INSTS_VALU"]*scale,
            "per_ray_surface_op": c["SQ_INSTS_VALU"]*scale*64/(
                n*rec["surfaces"]),
            "gfxclk_mhz_observed": clock_mhz,
            "valu_issue_frac": c["SQ_INSTS_VALU"]*scale*4/cyc,
            "valu_busy_frac": c["SQ_ACTIVE_INST_VALU"]*scale*4/cyc,
            "source": "profiles/valu_counters.json (rocprofv3 --pmc, round "
                      "4) + this run's clock and launch time"}

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
        rec["placement"] = {k: pl[k] for k in ("pieces", "piece_mib",
                                               "per_class", "fast", "ranges_tried", "range_kept",
                                               "store_pattern_GBps_per_range",
                                               "created", "ballast_blocks",
                                               "search_ms")}
        rec["_kind"] = kind
        if note:
            rec["note"] = note
        out.append(rec)
        log("[configs] %s: %.4f ms, frac %.3f" % (name, ms, rec["frac"]))

    # C1: singlet, 10^4 rays, one wavelength (launch-latency bound: 40 waves)
    s1 = ra.system_from_yaml(P.SINGLET)
    y, u = dc.bundle(10**4, 8., 0., 0)
    l1 = s1.wavelengths[0]
    g = ra.GeometricTrace(s1, device=device)
    g.rays_given(y, u, l1)
    record("C1 singlet, 10^4 rays", s1, g, len(y), l1, True, False,
           subsample_parity(ra, device, s1, y, u, l1, True, {}),
           reference_rate(P.SINGLET, y, u, l1, True, 10**4),
           "160 wavefronts on 256 CUs: bound by launch latency, not HBM")
    # C2: Cooke triplet, 10^6 rays x 3 wavelengths as ONE launch (ray groups)
    s2 = ra.system_from_yaml(P.COOKE % dict(
        air="air", sk16="SCHOTT-SK|N-SK16", f2="SCHOTT-F|N-F2"))
    ls = [587.56e-9, 656.27e-9, 486.13e-9]
    y, u = dc.bundle(10**6, 5.5, 5., 0)
    g = ra.GeometricTrace(s2, device=device)
    g.rays_given(y, u, l=ls)
    ref = None
    if refshim.available():
        rates = [reference_rate(P.cooke(lk), y, u, lk, True, 100_000)
                 for lk in ls]
        ref = {"value": sum(r["rays"] for r in rates)*(len(s2) - 1) /
               sum(r["seconds"] for r in rates), "kind": "reference",
               "rays": rates[0]["rays"], "note": "three traces, one per "
               "wavelength, as the reference has to run them"}
    record("C2 Cooke triplet, 10^6 rays x 3 wavelengths, one launch", s2, g,
           3*len(y), ls, True, False,
           subsample_parity(ra, device, s2, y, u, ls, True, {}, 64*1500),
           ref, kind="C2 3 x 10^6 rays")
    del g
    # C3 again with launch directions that differ from ray to ray (a bundle
    # as rays_given gets it from a caller's own generator): only z = 0 is
    # uniform across a 64-ray tile, the tile notes save 8 of 48 B per ray
    s3 = ra.system_from_yaml(P.DOUBLE_GAUSS)
    n3 = 10_000_000 if not args.rays or args.rays >= 10**6 else args.rays
    y, u = workload_rays(n3, 7)
    rng = np.random.default_rng(3)
    u[:, 0] += 1e-7*rng.standard_normal(n3)
    u[:, 1] += 1e-7*rng.standard_normal(n3)
    u[:, 2] = np.sqrt(1. - u[:, 0]**2 - u[:, 1]**2)
    g = ra.GeometricTrace(s3, device=device)
    g.rays_given(y, u)
    record("C3 double-Gauss, %d rays, per-ray launch directions (no "
           "uniform direction to fetch once per wavefront)" % n3, s3, g, n3,
           s3.wavelengths[0], True, False,
           subsample_parity(ra, device, s3, y, u, s3.wavelengths[0], True, {}),
           None, "the headline's bundles are collimated: their direction is "
           "read once per 64-ray tile; this is what a bundle with individual "
           "directions costs", kind="C3 host-seeded clip")
    del g, y, u
    # C4: aspheric phone lens, 10^7 rays: default and exact arithmetic
    s4 = ra.system_from_yaml(P.ASPHERE_PHONE)
    n4 = 10_000_000 if not args.rays or args.rays >= 10**6 else args.rays
    y, u = dc.bundle(n4, .6, 10., 4)
    y[:, 1] -= .5*np.tan(np.radians(10.))
    l4 = s4.wavelengths[0]
    ref4 = reference_rate(P.ASPHERE_PHONE, y, u, l4, True, 3000)
    if ref4 is not None:
        ref4["note"] = ("per-ray scipy.optimize.newton in a Python loop "
                        "(rayopt/elements.py:333-349): timed on 3000 rays "
                        "and extrapolated, BASELINE.md 3.4")
    for label, opts in (("default (FMA / rcp / rsq Newton, 1e-8 contract)",
                         {}), ("exact_asphere=True (the reference's bits)",
                               {"exact_asphere": 1})):
        g = ra.GeometricTrace(s4, device=device, **opts)
        g.rays_given(y, u, l4)
        record("C4 aspheric phone lens, %d rays, %s" % (n4, label), s4, g,
               n4, l4, True, False,
               subsample_parity(ra, device, s4, y, u, l4, True, opts), ref4,
               kind="C4 exact" if opts else "C4 default")
        del g
    del y, u
    # C5 on ONE GPU: double-Gauss, 10^8 rays built on the device (104 GB)
    if not args.no_configs5:
        s5 = ra.system_from_yaml(P.DOUBLE_GAUSS)
        nf = len(FIELD_FRACTIONS)
        m = (args.configs5_rays or 100_000_000)//nf//64*64
        pts = dc.disc_points(m, 91)
        g = ra.GeometricTrace(s5, device=device)
        g.rays_fields(np.c_[np.zeros(nf), FIELD_FRACTIONS], pts,
                      P.DOUBLE_GAUSS_PUPIL_Z, BUNDLE_RADIUS)
        g.propagate(clip=True)
        record("C5 on one GPU: double-Gauss, %d rays built on the device"
               % (m*nf), s5, g, m*nf, s5.wavelengths[0], True, True, None,
               None, "the 8-GPU form shards these rays and gathers y[L-1] "
               "over RCCL (bench.py --gpus 8 --total-rays 100000000)",
               kind="C3 host-seeded clip")
        ulast = np.asarray(g.u[-1])[::997, 0]
        out[-1]["finite_fraction_at_image_sampled"] = float(
            np.isfinite(ulast).mean())
        del g
        # the same rays as TEN batches of a tenth each, ten contexts traced
        # in turn: the cross-check of the layout in blocks (csrc/rt_lay.h) --
        # as ONE block this batch took 12.0 ms, the ten batches 10.3
        # (DESIGN.md section 9); in blocks the two agree
        try:
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
            t0 = time.perf_counter()
            for _ in range(20):
                turn()
            ms = (time.perf_counter() - t0)/20*1e3
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
        except Exception as err:      # a reported extra, never fatal
            out[-1]["as_ten_batches_in_turn"] = {"error": repr(err)[:200]}
    # what bounds each config: the store streams (HBM) or FP64 issue
    t = tele.stop() if tele is not None else None
    for k, rec in enumerate(out):
        kind = rec.pop("_kind", None)
        w = (t or {}).get(str(k)) or {}
        clock = (w.get("gfxclk_mhz") or [None]*3)[1]
        if w:
            rec["telemetry"] = {
                "gfxclk_mhz": clock,
                "socket_power_w": (w.get("socket_power_w") or [None]*3)[1],
                "power_limited_fraction": w.get("power_limited_fraction")}
        valu_roofline(rec, kind, rec["rays"], rec["kernel_ms"], clock)
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


if __name__ == "__main__":
    if len(sys.argv) == 4 and sys.argv[1] == "--telemetry-child":
        telemetry_child(int(sys.argv[2]), float(sys.argv[3]))
    else:
        main()
