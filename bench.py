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
  cpu_baseline  the numpy port of the reference path (oracle/trace_numpy.py,
                same whole-array numpy operations as rayopt) timed on this
                host, ONE core, on a bounded sample of the same workload;
                kind "reference" = rayopt itself, when /root/reference is
                present on the box
  cpu_baseline_all_cores   the same port on every host core (one forked
                process per core over contiguous ray shards)
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

HBM_PEAK_GBS = 8000.        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_ACHIEVABLE_GBS = 6290.  # same guide: measured float4 copy
FIELD_FRACTIONS = (0, .35, .5, .7, 1.)
BUNDLE_RADIUS = 17.


def log(*a):
    print(*a, file=sys.stderr, flush=True)


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
    """One propagate() of the numpy port -- or of rayopt itself where
    /root/reference exists -- on one core; doubles as a parity check of the
    bench run itself."""
    from oracle import trace_numpy as tn
    m = min(sample, y.shape[0])
    ys, us = y[:m], u[:m]
    tn.propagate(table, ys[:100000], us[:100000], clip=clip)   # warm
    t0 = time.perf_counter()
    Y, U, I, T = tn.propagate(table, ys, us, clip=clip)
    dt = time.perf_counter() - t0
    got = np.asarray(g.y[L - 1])[:m]
    ref = Y[-1]
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    fin = np.isfinite(ref)
    assert (np.abs(got[fin] - ref[fin]) <=
            1e-10*np.maximum(np.abs(ref[fin]), 1.)).all()
    out = {
        "value": m*S/dt, "unit": "ray-surface-ops/s", "cores": 1,
        "kind": "port",
        "sample": "first %d rays of the same workload, one propagate() of "
                  "the numpy port (%.1f s); host has %d cores" % (
                      m, dt, os.cpu_count()),
        "host": host_cpu(),
        # the port is the reference's arithmetic bit for bit (tests/): so is
        # what the GPU just computed in the timed loop
        "image_row_bit_identical_to_gpu": bool(
            np.array_equal(got, ref, equal_nan=True)),
    }
    from oracle import ref_timing
    ref = ref_timing.time_reference(ys, us, l, clip, want_image_row=Y[-1])
    if ref is not None:         # rayopt itself, where the box has it
        ref["port_value"] = m*S/dt
        out = ref
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

    def gather(self):
        from rayopt_amd._lib import RT_Y
        if self.exchange:
            self.eng.gather_final(RT_Y, self.L - 1, self.counts, 0,
                                  self.d_dst)

    def fence(self):
        self.eng.sync()
        if self.dist:
            if self.exchange:
                self.eng.comm_sync()
            self.group.barrier()

    def timed(self, step, steps, warmup, final_gather):
        """W untimed + exactly K timed calls of `step`, bracketed by device
        sync + barrier on both sides.  Returns (wall s, HIP-event ms over the
        K steps on the trace stream, ms of the last kernel)."""
        eng = self.eng
        for _ in range(warmup):
            step()
        self.fence()
        t0 = time.perf_counter()
        eng.event_record(0)
        for _ in range(steps):
            step()
        eng.event_record(1)
        if final_gather:
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
    ap.add_argument("--cpu-sample", type=int, default=10_000_000,
                    help="rays of the workload timed on the host (0: skip "
                         "every CPU leg); the default is the whole batch, "
                         "~10 s on one core")
    ap.add_argument("--cpu-procs", type=int, default=-1,
                    help="processes of the all-cores leg of the numpy port "
                         "(forked before the GPU is touched); -1 = one per "
                         "host core, 0 = skip")
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

    t0 = time.perf_counter()
    y, u = workload_rays(n, rank)
    cpu_all = None
    procs = os.cpu_count() if args.cpu_procs < 0 else args.cpu_procs
    if procs > 1 and args.cpu_sample > 0 and rank == 0 and not dist_mode:
        try:            # forks: before this process opens the GPU
            cpu_all = cpu_port_on_processes(system, y, u, clip, procs)
        except Exception as err:      # a reported extra, never fatal
            cpu_all = {"error": repr(err)[:200]}
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

    mode = {"clip": clip}

    def step():                 # the public call
        g.propagate(clip=mode["clip"])
        if dist_mode and args.gather_every_step:
            job.gather()

    def step_engine():          # the bare C-ABI call, table already there
        eng.trace(1, 0, mode["clip"])

    settle(g, args.settle, clip)    # setup, not part of W or K

    final_gather = dist_mode and not args.gather_every_step
    plain = not dist_mode and not args.option

    image_only = unclipped = full_i = engine_leg = None
    if args.extras and plain:
        # extension: keep only the image row (merit-function use): the
        # kernel leaves the HBM roofline for the FP64 one
        def step_image():
            g.propagate(clip=clip, keep=[0, -1])
        e_img, ev_img, _ = job.timed(step_image, args.steps, args.warmup,
                                     False)
        image_only = (e_img, ev_img/args.steps)
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

    elapsed, ev_ms, last_kernel_ms = job.timed(step, args.steps, args.warmup,
                                               final_gather)
    gather_ms = None
    if dist_mode:
        # the exchange alone (not part of `value`'s timed region)
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
    alg_bytes = n*(56*S + 24*stored_i - 24*skipped_u + 48)  # one GPU's shard
    achieved = alg_bytes/(kernel_ms*1e-3)/1e9
    traffic = traffic_source = traffic_detail = None
    plain_kernel = alias_on and not args.option
    if args.traffic == "live" and world == 1 and not dist_mode and \
            plain_kernel and not os.environ.get("RT_BENCH_CHILD") and \
            not _profiled_from_outside(os.environ):
        try:
            t0 = time.perf_counter()
            traffic, traffic_detail = traffic_live(n, clip)
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
            "frac_of_achievable_6290": achieved/HBM_ACHIEVABLE_GBS,
        },
    }
    if dist_mode:
        out["gather_ms"] = gather_ms
        out["kernel_ms_per_rank"] = per_rank_kernel_ms
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
        b_full = n*(80*S + 48)
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
        b_nc = n*(56*S + 24*stored_i + 48 - 24*sum(
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

    if world == 1 and not dist_mode and args.cpu_sample > 0:
        out["cpu_baseline"] = cpu_one_core(table, system, y, u, clip, S, g, L,
                                           args.cpu_sample, g.l)
        if cpu_all is not None:
            out["cpu_baseline_all_cores"] = cpu_all
        try:
            out["cpu_baseline_c"] = cpu_c_oracle(table, y, u, clip, S, g, L)
        except Exception as err:      # a reported extra, never fatal
            out["cpu_baseline_c"] = {"error": repr(err)[:200]}
    emit(json.dumps(out))
    if dist_mode:
        group.barrier()
        group.close()


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
    counts = np.array(group.broadcast(group.gather(m*nf)), dtype=np.int64)
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
    settle(g, args.settle, clip)
    elapsed, ev_ms, _ = job.timed(step, args.steps, args.warmup, True)
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
    main()
