#!/usr/bin/env python
"""Headline benchmark: ray-surface-ops/s of GeometricTrace.propagate().

    python bench.py --gpus N --steps K --warmup W

A "step" is one call of the public ``GeometricTrace.propagate(clip=True)`` --
re-packing the System, handing the table over and one fused pass of the hot
path over all S = len(system)-1 elements -- on one batch of synthetic rays
that is already resident in HBM.  Workload at every N: BASELINE.json
configs[2] -- the double-Gauss (L=13, S=12, spherical + stop), 10^7 rays per
GPU in five field bundles, clip=True (weak scaling: each rank traces its own
10^7-ray shard, different seeds).  W untimed and exactly K timed steps between
device sync + barrier on both sides; the maximum over ranks; rank 0 prints ONE
strict-JSON line.

N > 1 is one process per GPU.  ``python bench.py --gpus N`` starts the N
workers itself; under a per-GPU launcher (``python -m torch.distributed.run
--nproc-per-node N ... bench.py --gpus N``: RANK / LOCAL_RANK / WORLD_SIZE in
the environment) it uses the ranks it is given.  The host side is
PyTorch-free: rendezvous, barrier and max-over-ranks go over
rayopt_amd.distributed.HostGroup (TCP on 127.0.0.1), the device exchange is
the engine's own RCCL gather.  The trace needs no communication; the one
exchange of the job -- the gather of the last-surface intercepts y[L-1] of all
ranks to rank 0 over xGMI -- runs once, after the last step, INSIDE the timed
region (pipelined with the last step's trace in --gather-chunks pieces).

Besides the contract's keys the line carries ``roofline`` (achieved =
algorithmic bytes per launch / launch time from HIP events on the kernel's
own stream; algorithmic bytes = what row 0 costs + 56 B written per
ray-surface op (y 24, u 24, t 8) + 24 B where i has to be materialised;
``traffic`` = HBM bytes from PMC counters of a child run under rocprofv3 in
this very run), ``cpu_baseline`` (rayopt itself, imported unmodified from
oracle/_ref, one process on this host, a bounded sample; kind "reference")
and the legs of bench_legs.py: ``valu`` (SQ counters of the FP64-bound legs,
same child runs), ``end_to_end`` (host rays in, image row out: PCIe
inclusive, never `value`), ``generated_batch``, ``image_row_only``,
``propagate_api``, ``configs`` (one record per BASELINE config with a parity
subsample), ``consumers``, ``cpu_baseline_all_cores``, ``cpu_baseline_c``,
``telemetry``, ``wall_s``; --extras adds ``full_i`` / ``unclipped`` and the
C5 batch as ten batches in turn.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

import bench_legs as legs
from bench_legs import (ROOT, HBM_PEAK_GBS, HBM_ACHIEVABLE_GBS,     # noqa: F401
                        HBM_NOMINAL_MHZ, FIELD_FRACTIONS, BUNDLE_RADIUS, Job,
                        log, lap, workload_rays)


def parse_args():
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
                         "host core but at most 64, -2 = one per host core, "
                         "0 = skip")
    ap.add_argument("--settle", type=float, default=0.3,
                    help="seconds of untimed launches during setup so the "
                         "device reaches its sustained clocks; 0 disables")
    ap.add_argument("--extras", action="store_true",
                    help="also: full_i (80 B/op) and unclipped modes, the C "
                         "port per team size, C5 as ten batches in turn, "
                         "small-batch latency")
    ap.add_argument("--counters", "--traffic", dest="counters",
                    choices=("live", "profile", "off"), default="live",
                    help="roofline.traffic and the VALU counters: 'live' = "
                         "rocprofv3 --pmc passes of a child run on this box "
                         "(N = 1 only; falls back to 'profile'), 'profile' "
                         "= the committed profiles/traffic.json, 'off'")
    ap.add_argument("--no-engine-leg", action="store_true")
    ap.add_argument("--no-api-leg", action="store_true",
                    help="skip the propagate_api / generated-batch legs")
    ap.add_argument("--gather-every-step", action="store_true",
                    help="N>1: gather y[L-1] to rank 0 in every step")
    ap.add_argument("--no-configs4", action="store_true",
                    help="N>1: skip the BASELINE configs[4] leg")
    ap.add_argument("--gather-chunks", type=int, default=4,
                    help="N>1: the last step is traced in this many pieces "
                         "and the gather of piece k overlaps the trace of "
                         "piece k+1 (1: trace, then gather)")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the per-config, consumer and host-path legs")
    ap.add_argument("--no-configs5", action="store_true",
                    help="skip the 10^8-ray one-GPU batch (104 GB)")
    ap.add_argument("--configs5-rays", type=int, default=0)
    ap.add_argument("--only-config", choices=legs.ONLY_KEYS, default=None,
                    help="ONE config leg and nothing else (a leg's rocprofv3 "
                         "--kernel-trace --stats run: profiles/r06_final/"
                         "legs/); prints that leg's own JSON line")
    ap.add_argument("--option", action="append", default=[],
                    help="kernel variant key=value (rt_set_option)")
    return ap.parse_args()


def strict(obj):
    """NaN / inf -> null: the line is strict JSON."""
    if isinstance(obj, float):
        return obj if math.isfinite(obj) else None
    if isinstance(obj, (np.floating, np.integer, np.bool_)):
        return strict(obj.item())
    if isinstance(obj, dict):
        return {str(k): strict(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [strict(v) for v in obj]
    return obj


def supervise(argv):
    """N = 1: the whole run -- timed loop and side legs -- happens in a WORKER
    process; this one only waits and hands the worker's line on.  The
    platform can end a process that maps device memory with "Memory access
    fault by GPU" (round 5: the GPU suite once in about eight one-process
    runs; round 6: bench.py once, right after a search for pieces; DESIGN.md
    section 9) -- a process that dies prints no line at all.  A worker that
    dies is started again, the third time without the side legs; the line
    then says so (`attempts`, `died`).  Nothing of the measurement happens
    here."""
    import subprocess
    died = []
    for attempt in (1, 2, 3):
        extra = ["--no-configs"] if attempt == 3 else []
        res = subprocess.run(
            [sys.executable, os.path.abspath(__file__)] + argv + extra,
            env=dict(os.environ, RT_BENCH_WORKER="1"), stdout=subprocess.PIPE)
        lines = [ln for ln in res.stdout.decode(errors="replace").splitlines()
                 if ln.strip()]
        line = None
        if lines:
            try:
                line = json.loads(lines[-1])
                line = line if "metric" in line else None
            except ValueError:
                line = None
        if line is not None and res.returncode <= 0:
            # (a worker that printed its whole line and was killed while
            # leaving has measured all the same)
            if attempt == 1 and res.returncode == 0:
                sys.stdout.write(lines[-1] + "\n")
            else:
                if attempt > 1:
                    line["attempts"] = attempt
                    line["died"] = died
                if res.returncode:
                    line["worker_killed_after_its_line_by_signal"] = \
                        -res.returncode
                sys.stdout.write(json.dumps(line) + "\n")
            sys.stdout.flush()
            return 0
        if res.returncode >= 0:
            # an error the worker reported itself (its message is on stderr):
            # not a death, nothing a second start would change
            sys.stdout.write(res.stdout.decode(errors="replace"))
            return res.returncode or 1
        died.append("attempt %d: killed by signal %d" % (attempt,
                                                         -res.returncode))
        log("[bench] the worker process was killed by signal %d and printed "
            "no line (attempt %d)%s" % (
                -res.returncode, attempt,
                "; starting it again" + (" without the side legs"
                                         if attempt == 2 else "")
                if attempt < 3 else ""))
    return 1


def main():
    args = parse_args()
    if args.gpus == 1 and "WORLD_SIZE" not in os.environ and \
            not os.environ.get("RT_BENCH_WORKER") and not args.only_config:
        raise SystemExit(supervise(sys.argv[1:]))
    if args.only_config:
        import rayopt_amd as ra
        print(json.dumps(strict(legs.only_config(ra, 0, args.only_config,
                                                 args))))
        return
    from rayopt_amd import distributed as D
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: one worker per GPU, this process
        # only waits (rank 0's JSON line goes to the inherited stdout)
        raise SystemExit(D.spawn_workers(
            args.gpus,
            check_devices=not os.environ.get("RT_BENCH_SHARE_DEVICE")))
    world, rank, local_rank = D.world_info()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # RT_BENCH_SHARE_DEVICE=1 (tests only): every rank opens device 0 and the
    # RCCL exchange is left out (RCCL refuses two ranks on one device) --
    # unless RT_TRANSPORT_LIBRARY names a stand-in for librccl.so (the
    # shared-memory transport of tests/stubs): then the engine's gather runs
    # as it is and rank 0 checks every shard.  Such lines say "test_mode".
    share = bool(os.environ.get("RT_BENCH_SHARE_DEVICE"))
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
            local_rank = 0      # the launcher masks the devices per rank
        if local_rank >= have:
            raise SystemExit("--gpus %d: %d devices needed, %d visible"
                             % (args.gpus, max(world, local_rank + 1), have))
    if world > 1:
        check_device()
    # ONE JSON line on stdout: native libraries (RCCL prints a banner) must
    # not get at it, so fd 1 points at stderr for the whole run
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    # RT_BENCH_FORCE_DIST=1 exercises the whole multi-process path (host
    # group, RCCL communicator, pipelined gather) with a single rank
    dist_mode = world > 1 or bool(os.environ.get("RT_BENCH_FORCE_DIST"))
    group = D.HostGroup(world, rank) if dist_mode else None
    child = bool(os.environ.get("RT_BENCH_CHILD")) or \
        legs.profiled_from_outside(os.environ)

    import rayopt_amd as ra
    from rayopt_amd import prescriptions as P
    from rayopt_amd.pack import pack_system

    clip = not args.no_clip
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    L = len(system)
    S = L - 1
    counts = (D.shard_counts(args.total_rays, world) if args.total_rays
              else np.full(world, args.rays, dtype=np.int64))
    n = int(counts[rank])
    lap("imports, system")
    y, u = workload_rays(n, rank)
    lap("workload rays on the host")
    plain = not dist_mode and not args.option
    side_legs = plain and world == 1 and not args.no_configs and not child
    cpu_all = None
    if rank == 0 and not dist_mode and args.cpu_sample > 0:
        cpu_all = legs.cpu_all_cores(args, system, y, u, clip)  # forks: first
    if world == 1:
        check_device()
    g = ra.GeometricTrace(system, device=local_rank)
    eng = g.engine
    for kv in args.option:
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    t0 = time.perf_counter()
    g.rays_given(y, u)          # rays resident in HBM from here on
    upload_s = time.perf_counter() - t0
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
    job.chunks = max(1, args.gather_chunks)
    comm_info = None        # what the transport itself says it is
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

    def last_step():            # the K-th step, its gather pipelined with it
        g.propagate(clip=mode["clip"], chunks=job.chunks,
                    after_chunk=job.gather_chunk)

    tele = legs.Telemetry(local_rank, period=0.01) if (
        rank == 0 and not os.environ.get("RT_BENCH_CHILD")) else None
    mark = tele.mark if tele is not None else (lambda label: None)
    mark("settle:begin")
    legs.settle(g, args.settle, clip)    # setup, not part of W or K
    mark("settle:end")
    lap("settle")
    final_gather = dist_mode and not args.gather_every_step
    before = legs.legs_before_the_loop(args, job, g, mode, step, mark,
                                       plain, side_legs)
    plain_loop = None
    if dist_mode and final_gather:
        # the same K steps WITHOUT the exchange, same process: what the one
        # exposed gather costs the job
        e_plain, _, _ = job.timed(step, args.steps, args.warmup, False)
        plain_loop = group.allreduce_max(e_plain)
    mark("loop:begin")
    elapsed, ev_ms, last_kernel_ms = job.timed(
        step, args.steps, args.warmup, final_gather,
        last_step if (final_gather and job.chunks > 1 and job.exchange)
        else None)
    mark("loop:end")
    lap("image-row / engine legs, warm-up and the timed loop")
    gather_ms = gather_exposed = None
    if dist_mode:
        if final_gather and job.exchange:
            # HIP events: the whole exchange, and what was left of it after
            # this rank's last trace kernel had finished
            tot, exp = eng.gather_ms()
            gather_exposed = [group.allreduce_max(tot),
                              group.allreduce_max(exp)]
        job.fence()             # the exchange alone, unpipelined, untimed
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
    finite = float(np.isfinite(np.asarray(g.u[L - 1])[:, 0]).mean())
    if dist_mode and rank == 0 and job.exchange:
        gathered = eng.copy_to_host(d_dst, int(counts.sum())*3*8)
        gathered = gathered.reshape(3, -1)
        assert np.array_equal(gathered[:, :n].T, ylast, equal_nan=True), \
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
    generated = None
    if plain and not args.no_api_leg:
        generated = legs.run_generated(ra, system, local_rank, n, clip, args)
        lap("generated batch")
    configs4 = None
    if dist_mode and world > 1 and not args.no_configs4 and \
            not args.total_rays:
        del ylast, y, u
        configs4 = legs.run_configs4(ra, system, g, job, group, world, rank,
                                     args, clip)
    if rank != 0:
        group.barrier()
        group.close()
        return

    # ---- the line -------------------------------------------------------
    table, _ = pack_system(system, g.l, g.n[0])
    total_rays = int(counts.sum())
    alias_on = not any(kv == "alias_i=0" for kv in args.option)
    read_bytes, uniform_share = legs.input_bytes(eng, n)
    alg_bytes, per_op = legs.algorithmic_bytes(
        table[None], n, clip, alias=alias_on, read_bytes=read_bytes)
    achieved = alg_bytes/(kernel_ms*1e-3)/1e9
    placement = eng.placement()
    placement["workgroups_per_cu_cap"] = 4 if placement["fast"] else 2
    par = "ray shards x%d" % world
    if dist_mode:
        par += (", one process per GPU, host group over TCP (no PyTorch), "
                "RCCL gather of y[L-1] to rank 0 %s (gather alone: %.2f ms)"
                % ("in every step" if args.gather_every_step else
                   "once, after the last step, inside the timed region",
                   gather_ms))
    out = {
        "metric": "ray-surface-ops/sec",
        "value": total_rays*S*args.steps/elapsed,
        "unit": "ray-surface-ops/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed*1e3/args.steps,
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
            "rays_per_gpu": n, "total_rays": total_rays, "surfaces": S,
            "clip": clip, "finite_fraction_at_image": finite,
            "settle_s": args.settle, "parallelism": par,
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": achieved/HBM_PEAK_GBS,
            "traffic": None, "traffic_source": None,
            "kernel": "rt_trace_kernel", "kernel_ms": kernel_ms,
            "algorithmic_bytes_per_launch": alg_bytes,
            "bytes_per_ray_surface_op": per_op,
            "input_bytes_per_ray": read_bytes/n,
            "input_uniform_share_y0y1y2u0u1u2": uniform_share,
            # SURVEY 8(d) counts 80 B/op written and 48 B/ray read: the rows
            # of i that ARE rows of u are served, not stored, and launch
            # components uniform across a 64-ray tile are fetched once
            "frac_if_80B_per_op_and_48B_per_ray_were_moved":
                n*(80.*S + 48.)/(kernel_ms*1e-3)/1e9/HBM_PEAK_GBS,
            "frac_of_achievable_6290": achieved/HBM_ACHIEVABLE_GBS,
            "placement": placement,
        },
    }
    if dist_mode:
        out.update(gather_ms=gather_ms, transport=comm_info,
                   kernel_ms_per_rank=per_rank_kernel_ms)
        if gather_exposed is not None:
            out.update(gather_pipelined_ms=gather_exposed[0],
                       gather_exposed_ms=gather_exposed[1],
                       gather_chunks=job.chunks)
        if plain_loop is not None:
            out["plain_loop_ms_per_step"] = plain_loop*1e3/args.steps
            out["exchange_cost_ratio"] = elapsed/plain_loop
            out["note"] = (
                "`value` includes the job's one exchange (RCCL gather of "
                "y[L-1] to rank 0, the last step traced in %d chunks so that "
                "all but the last chunk's gather overlaps tracing): "
                "ms_per_step x steps = plain_loop_ms_per_step x steps + the "
                "exposed part of the gather" % job.chunks)
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
        out["generated_batch"] = generated
    legs.record_legs_before(out, before, args, n, S, L, total_rays, elapsed,
                            read_bytes, table, clip, alias_on)
    if args.extras and out.get("propagate_api") is not None:
        out["propagate_api"].update(
            legs.small_batch_latency(ra, system, local_rank))
    counters = None
    if world == 1 and not dist_mode and alias_on and not args.option and \
            not child and args.counters != "off":
        counters = legs.counters_for_the_line(out, args, n, clip)
    if tele is not None:
        legs.record_telemetry(out, tele.stop(), counters, n)
    if side_legs:
        del ylast
        legs.side_legs(out, ra, g, system, local_rank, args, n, S, clip,
                       kernel_ms, alg_bytes, achieved, counters, y, u,
                       upload_s)
    if world == 1 and not dist_mode and args.cpu_sample > 0:
        legs.cpu_legs(out, args, table, system, y, u, clip, S, g, L, cpu_all)
    if side_legs:
        try:
            out["consumers"] = legs.run_consumers(
                ra, g, system, n, len(FIELD_FRACTIONS),
                out.get("cpu_baseline"))
        except Exception as err:      # reported extras, never fatal
            out["consumers"] = {"error": repr(err)[:300]}
        lap("consumers")
    out["wall_s"] = {"since_start": legs.LAPS,
                     "note": "seconds since this interpreter reached "
                             "bench.py, after each phase"}
    sys.stdout.flush()
    # the full records go to a side file; the driver's line is the contract
    # plus one short record per leg (bench_legs.core_line, < 8 KB)
    full = strict(out)
    legs.leg_summaries(full)
    line = json.dumps(legs.core_line(full, legs.write_detail(full)),
                      allow_nan=False)
    assert len(line) < legs.CORE_LINE_LIMIT, len(line)
    os.write(real_stdout, (line + "\n").encode())
    if dist_mode:
        group.barrier()
        group.close()


if __name__ == "__main__":
    if len(sys.argv) == 4 and sys.argv[1] == "--telemetry-child":
        legs.telemetry_child(int(sys.argv[2]), float(sys.argv[3]))
    elif len(sys.argv) >= 2 and sys.argv[1] == "--pmc-child":
        legs.pmc_child(*sys.argv[2:])
    else:
        main()
