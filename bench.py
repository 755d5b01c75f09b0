#!/usr/bin/env python
"""Headline benchmark: ray-surface-ops/s of GeometricTrace.propagate().

    python bench.py --gpus N --steps K --warmup W

A "step" is one fused pass of the hot path (all S = len(system)-1 elements)
over one batch of synthetic rays that is already resident in HBM.  Workload
at every N: BASELINE.json configs[2] -- the double-Gauss (L=13, S=12,
spherical + stop), 10^7 rays per GPU in five field bundles, clip=True (weak
scaling: each rank traces its own 10^7-ray shard, different seeds).  For N>1
(one process per GPU, launched by torch.distributed.run) the trace itself
needs no communication; the one exchange of the job -- the RCCL gather of the
last-surface intercepts y[L-1] of all ranks to rank 0 over xGMI -- runs once,
after the last step, INSIDE the timed region (results otherwise stay sharded
in HBM exactly as they stay in HBM at N=1).  `gather_ms` reports it alone;
--gather-every-step makes every step a complete job (trace + gather,
pipelined), which is bound by the root's xGMI ingest (24 B/ray over <= 7
links), not by the engine.

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
  full_i        (--extras) the same timed loop with every row of `i`
                materialised (rt_set_option alias_i=0): the 80 B/op figure of
                SURVEY 8(d)
  unclipped     (--extras) the same timed loop with clip=False, the
                reference's default
  image_row_only (--extras) propagate(keep=[-1]): the FP64 side of the kernel
  cpu_baseline  the numpy port of the reference path (oracle/trace_numpy.py,
                same whole-array numpy operations as rayopt) timed on this
                host, one core, on a bounded sample of the same workload
  cpu_baseline_c the independent plain-C port (oracle/trace_c.c) with OpenMP
                on every host core: the compiled multi-threaded CPU figure
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


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def workload_rays(n, rank):
    from rayopt_amd import prescriptions as P
    from rayopt_amd.bundles import multi_field_bundle
    fields = [f*P.DOUBLE_GAUSS_FIELD_DEG for f in (0, .35, .5, .7, 1.)]
    return multi_field_bundle(n, 17., fields, seed=1000*rank,
                              z_pupil=P.DOUBLE_GAUSS_PUPIL_Z)


_SHARD = {}


def _shard_worker(k):
    from oracle import trace_numpy as tn
    table, y, u, clip, bounds = (_SHARD[key] for key in
                                 ("table", "y", "u", "clip", "bounds"))
    lo, hi = bounds[k]
    Y, U, I, T = tn.propagate(table, y[lo:hi], u[lo:hi], clip=clip)
    return float(np.nansum(Y[-1]))      # touch the result


def cpu_c_oracle(table, y, u, clip, S, g, L, sample=2_000_000):
    """The independent plain-C oracle (oracle/trace_c.c, OpenMP over rays) on
    every host core: what a compiled multi-threaded CPU implementation of the
    same path reaches on this box.  Doubles as a second parity check."""
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
    # more threads are not always faster (cgroup quota, memory system): take
    # the best team size
    teams = sorted({min(os.cpu_count(), t) for t in (16, 64, os.cpu_count())})
    best, out, cores = None, None, os.cpu_count()
    build_c.propagate(table, ys, us, clip=clip)      # touch the output pages
    for team in (teams if gomp is not None else teams[-1:]):
        if gomp is not None:
            gomp.omp_set_num_threads(team)
        for _ in range(3):
            t0 = time.perf_counter()
            out = build_c.propagate(table, ys, us, clip=clip, out=out)
            dt = time.perf_counter() - t0
            if best is None or dt < best:
                best, cores = dt, team
    Y = out[0]
    got = np.asarray(g.y[L - 1])[:m]
    same = np.array_equal(got, Y[-1], equal_nan=True)
    return {
        "value": m*S/best,
        "unit": "ray-surface-ops/s",
        "cores": cores,
        "kind": "port",
        "sample": "first %d rays, best propagate() of the C port with OpenMP "
                  "over team sizes %s, same output arrays (%.3f s); host has "
                  "%d cores" % (m, teams, best, os.cpu_count()),
        "image_row_bit_identical_to_gpu": bool(same),
    }


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
    S = len(system) - 1
    return {"value": len(y)*S/dt, "unit": "ray-surface-ops/s",
            "cores": procs, "kind": "port",
            "sample": "the whole batch on %d processes, contiguous shards "
                      "(%.2f s)" % (procs, dt)}


class _DeviceView:
    """numpy-style view of raw device memory for torch.as_tensor."""
    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {
            "shape": tuple(shape), "typestr": "<f8", "data": (int(ptr), False),
            "version": 2, "strides": None}


class TorchGather:
    """Gather of row y[L-1] to rank 0 with torch.distributed send/recv on
    views of the engine's buffers; used only if the engine's own RCCL
    communicator cannot be created."""
    def __init__(self, torch, dist, eng, n, world, rank, d_dst, L):
        from rayopt_amd._lib import RT_Y
        self.torch, self.dist, self.eng = torch, dist, eng
        self.n, self.world, self.rank = n, world, rank
        self.row, self.d_dst = (RT_Y, L - 1), d_dst
        self.src = self.dst = None

    def _views(self):
        torch, n = self.torch, self.n
        src = torch.as_tensor(_DeviceView(self.eng.device_ptr(*self.row),
                                          (3, self.eng.ld)), device="cuda")
        self.src = src[:, :n]
        if self.rank == 0:
            self.dst = torch.as_tensor(
                _DeviceView(self.d_dst, (3, n*self.world)), device="cuda")

    def __call__(self):
        torch, dist = self.torch, self.dist
        self.eng.sync()                  # the trace that produced the row
        if self.src is None:
            self._views()
        if self.rank == 0:
            self.dst[:, :self.n].copy_(self.src)
            ops = []
            for r in range(1, self.world):
                for c in range(3):
                    ops.append(dist.P2POp(
                        dist.irecv, self.dst[c, r*self.n:(r + 1)*self.n], r))
        else:
            stage = self.src.contiguous()
            ops = [dist.P2POp(dist.isend, stage[c], 0) for c in range(3)]
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        torch.cuda.synchronize()


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


def main():
    # the contract is ONE JSON line on stdout: native libraries (RCCL prints a
    # version banner) must not get at it, so fd 1 is pointed at stderr for the
    # whole run and the line is written to the saved descriptor at the end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(line):
        sys.stdout.flush()
        os.write(real_stdout, (line + "\n").encode())

    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rays", type=int, default=10_000_000,
                    help="rays per GPU")
    ap.add_argument("--no-clip", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=10_000_000,
                    help="rays of the workload timed on the host (0: skip); "
                         "the default is the whole batch, ~10 s on one core")
    ap.add_argument("--settle", type=float, default=0.3,
                    help="seconds of untimed launches during setup so the "
                         "device reaches its sustained clocks (boxes of the "
                         "pool take ~50 launches); 0 disables")
    ap.add_argument("--extras", action="store_true",
                    help="also time the full_i (80 B/op) and image-row-only "
                         "modes (separate timed loops, reported as extra "
                         "objects); off by default so that every launch of "
                         "the default command is the headline kernel")
    ap.add_argument("--gather-every-step", action="store_true",
                    help="N>1: gather y[L-1] to rank 0 in every step")
    ap.add_argument("--cpu-procs", type=int, default=0,
                    help="also time the numpy port on this many host "
                         "processes over contiguous ray shards (forked "
                         "before the GPU is touched); reported as "
                         "cpu_baseline_procs")
    ap.add_argument("--option", action="append", default=[],
                    help="kernel variant key=value (rt_set_option)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit(
                "--gpus %d needs one process per GPU: launch with python -m "
                "torch.distributed.run --nproc-per-node %d ..." % (
                    args.gpus, args.gpus))
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    # RT_BENCH_FORCE_DIST=1 exercises the whole multi-process path (torch
    # rendezvous, RCCL communicator, pipelined gather) with a single rank
    dist_mode = world > 1 or bool(os.environ.get("RT_BENCH_FORCE_DIST"))
    dist = None
    torch = None
    if dist_mode:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device(
            "cuda", local_rank))

    import rayopt_amd as ra
    from rayopt_amd import prescriptions as P
    from rayopt_amd._lib import RT_Y

    clip = not args.no_clip
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    L = len(system)
    S = L - 1
    n = args.rays

    t0 = time.perf_counter()
    y, u = workload_rays(n, rank)
    cpu_procs = None
    if args.cpu_procs > 1 and rank == 0 and not dist_mode:
        cpu_procs = cpu_port_on_processes(system, y, u, clip, args.cpu_procs)
    g = ra.GeometricTrace(system, device=local_rank)
    eng = g.engine
    for kv in args.option:
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    g.rays_given(y, u)          # rays resident in HBM from here on
    log("[rank %d] %d rays generated + uploaded in %.2f s" % (
        rank, n, time.perf_counter() - t0))

    # RCCL gather of the final intercepts (only where there is an exchange)
    counts = None
    d_dst = 0
    torch_gather = None
    if dist_mode:
        from rayopt_amd.distributed import init_engine_comm, shard_counts
        counts = shard_counts(n*world, world)   # weak scaling: n per rank
        if rank == 0:
            d_dst = eng.scratch(int(counts.sum())*3*8)
        try:
            if os.environ.get("RT_BENCH_FORCE_TORCH_GATHER"):
                raise ra.EngineError("forced")
            init_engine_comm(eng, dist)
            ok = 1
        except ra.EngineError as exc:
            log("[rank %d] engine RCCL communicator unavailable (%s)" % (
                rank, exc))
            ok = 0
        flag = torch.tensor([ok], device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            # safety net for the exchange only: the same RCCL send/recv
            # through torch.distributed on views of the engine's device
            # memory (no host staging, no CPU path)
            torch_gather = TorchGather(torch, dist, eng, n, world, rank,
                                       d_dst, L)

    from rayopt_amd.pack import pack_system
    table, ns = pack_system(system, g.l, g.n[0])
    eng.upload_system(table)

    def gather():
        if torch_gather is not None:
            torch_gather()
        else:
            eng.gather_final(RT_Y, L - 1, counts, 0, d_dst)

    mode = {"clip": clip}

    def step():
        eng.trace(1, 0, mode["clip"])
        if dist_mode and args.gather_every_step:
            gather()

    def fence():
        eng.sync()
        if dist_mode:
            eng.comm_sync()
            torch.cuda.synchronize()
            dist.barrier()

    if args.settle > 0:         # setup, not part of W or K
        t_end = time.perf_counter() + args.settle
        while time.perf_counter() < t_end:
            for _ in range(10):
                eng.trace(1, 0, clip)
            eng.sync()

    def timed_loop():
        for _ in range(args.warmup):
            step()
        fence()
        t0 = time.perf_counter()
        eng.event_record(0)
        for _ in range(args.steps):
            step()
        eng.event_record(1)
        if dist_mode and not args.gather_every_step:
            # the job's one exchange: final intercepts to rank 0
            gather()
        fence()
        return (time.perf_counter() - t0, eng.event_elapsed(0, 1),
                eng.kernel_ms())

    image_only = None
    if args.extras and not dist_mode and not args.option:
        # extension: keep only the image row (merit-function use): the
        # kernel leaves the HBM roofline for the FP64 one
        mask = np.zeros(L, dtype=np.uint8)
        mask[0] = mask[L - 1] = 1
        eng.set_keep_rows(mask)
        e_img, ev_img, _ = timed_loop()
        image_only = (e_img, ev_img/args.steps)
        eng.set_keep_rows(None)
    unclipped = None
    if args.extras and not dist_mode and not args.option and clip:
        # the reference's default: propagate(clip=False); the u rows of the
        # elements that do not bend the ray are not written either
        mode["clip"] = False
        e_nc, ev_nc, _ = timed_loop()
        unclipped = (e_nc, ev_nc/args.steps)
        mode["clip"] = clip
    full_i = None
    if args.extras and not dist_mode and not any(kv.startswith("alias_i")
                                                 for kv in args.option):
        # reference point: every row of `i` written (80 B per op)
        eng.set_option("alias_i", 0)
        eng.upload_system(table)
        e_full, ev_full, _ = timed_loop()
        full_i = (e_full, ev_full/args.steps)
        eng.set_option("alias_i", 1)
        eng.upload_system(table)
    elapsed, ev_ms, last_kernel_ms = timed_loop()
    gather_ms = None
    if dist_mode:
        # the exchange alone (not part of `value`'s timed region)
        fence()
        t0 = time.perf_counter()
        gather()
        fence()
        gather_ms = (time.perf_counter() - t0)*1e3
    if dist_mode:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # sanity on the result of the last step (not timed): a few per cent of
    # the rays vignette, everything else reaches the image
    ylast = np.asarray(g.y[L - 1])
    ulast = np.asarray(g.u[L - 1])
    finite = float(np.isfinite(ulast[:, 0]).mean())
    if dist_mode and rank == 0:
        gathered = eng.copy_to_host(d_dst, int(counts.sum())*3*8)
        gathered = gathered.reshape(3, -1)
        mine = gathered[:, :n].T
        assert np.array_equal(mine, ylast, equal_nan=True), \
            "gathered shard 0 differs from the local result"
        assert np.isfinite(gathered).mean() > 0.9

    if rank != 0:
        if dist_mode:
            dist.barrier()
            dist.destroy_process_group()
        return

    total_rays = n*world
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
    kernel_ms = (ev_ms/args.steps if not (dist_mode and args.gather_every_step)
                 else last_kernel_ms)
    achieved = alg_bytes/(kernel_ms*1e-3)/1e9
    prof = traffic_from_profile()
    traffic = None
    if prof and prof.get("rays") == n and prof.get("clip") == clip and \
            prof.get("alias_i", 0) == int(alias_on):
        traffic = prof.get("hbm_bytes_per_launch")

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
            "workload": "double-Gauss (BASELINE configs[2]): L=%d elements, "
                        "S=%d propagated surfaces, %d rays/GPU in 5 field "
                        "bundles, clip=%s" % (L, S, n, clip),
            "rays_per_gpu": n,
            "surfaces": S,
            "clip": clip,
            "finite_fraction_at_image": finite,
            "settle_s": args.settle,
            "parallelism": "ray shards x%d%s" % (
                world, (", RCCL gather of y[L-1] to rank 0 %s (gather alone: "
                        "%.2f ms%s)" % ("in every step"
                                        if args.gather_every_step
                                        else "once, after the last step, "
                                        "inside the timed region", gather_ms,
                                        ", via torch.distributed" if
                                        torch_gather is not None else ""))
                if dist_mode else ""),
        },
        "roofline": {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved/HBM_PEAK_GBS,
            "traffic": traffic,
            "kernel": "rt_trace_kernel",
            "kernel_ms": kernel_ms,
            "algorithmic_bytes_per_launch": alg_bytes,
            "bytes_per_ray_surface_op": (56*S + 24*stored_i - 24*skipped_u)/S,
            "frac_of_achievable_6290": achieved/HBM_ACHIEVABLE_GBS,
        },
    }

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
            "note": "propagate(keep=[-1]): all %d surfaces traced, only the "
                    "image row stored (80 B/ray); FP64-VALU bound" % S,
        }

    if world == 1 and not dist_mode and args.cpu_sample > 0:
        from oracle import trace_numpy as tn
        m = min(args.cpu_sample, n)
        ys, us = y[:m], u[:m]
        tn.propagate(table, ys[:100000], us[:100000], clip=clip)   # warm
        t0 = time.perf_counter()
        Y, U, I, T = tn.propagate(table, ys, us, clip=clip)
        dt = time.perf_counter() - t0
        # the sample doubles as a parity check of the bench run itself
        got = np.asarray(g.y[L - 1])[:m]
        ref = Y[-1]
        assert np.array_equal(np.isnan(got), np.isnan(ref))
        fin = np.isfinite(ref)
        assert (np.abs(got[fin] - ref[fin]) <=
                1e-10*np.maximum(np.abs(ref[fin]), 1.)).all()
        out["cpu_baseline"] = {
            "value": m*S/dt,
            "unit": "ray-surface-ops/s",
            "cores": 1,
            "kind": "port",
            "sample": "first %d rays of the same workload, one "
                      "propagate() of the numpy port (%.1f s); host has %d "
                      "cores" % (m, dt, os.cpu_count()),
        }
        try:
            out["cpu_baseline_c"] = cpu_c_oracle(table, y, u, clip, S, g, L)
        except Exception as err:      # a reported extra, never fatal
            out["cpu_baseline_c"] = {"error": repr(err)[:200]}
    if cpu_procs is not None:
        out["cpu_baseline_procs"] = cpu_procs
    emit(json.dumps(out))
    if dist_mode:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
