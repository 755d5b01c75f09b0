#!/usr/bin/env python
"""rays_given() of 10^7 rays (480 MB, pageable numpy arrays): the DMA engine
against a copy kernel reading the mapped staging buffers, per number of
staging-copy threads; and the image row's way out (all of it, x and y only).
One JSON line per case (a process each: the switches are read once)."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import numpy as np
    import rayopt_amd as ra
    from rayopt_amd import prescriptions as P
    from bench import workload_rays
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    L = len(system)
    n = 10_000_000
    y, u = workload_rays(n, 0)
    g = ra.GeometricTrace(system)
    g.rays_given(y, u)
    t = []
    for _ in range(6):
        t0 = time.perf_counter()
        g.rays_given(y, u)
        t.append((time.perf_counter() - t0)*1e3)
    g.propagate(clip=True)
    row, xy = [], []
    for _ in range(4):
        g.propagate(clip=True)
        g.engine.sync()
        t0 = time.perf_counter()
        a = g.y[L - 1, :, :2]
        xy.append((time.perf_counter() - t0)*1e3)
        g.propagate(clip=True)
        g.engine.sync()
        t0 = time.perf_counter()
        a = np.asarray(g.y[L - 1])
        row.append((time.perf_counter() - t0)*1e3)
    print(json.dumps({
        "h2d": "kernel" if os.environ.get("RT_H2D_KERNEL") else "dma",
        "threads": int(os.environ["RT_COPY_THREADS"]),
        "rays_given_ms": [round(v, 2) for v in t],
        "h2d_GBps_best": round(48*n/min(t)/1e6, 1),
        "row_ms": [round(v, 2) for v in row],
        "xy_ms": [round(v, 2) for v in xy]}), flush=True)


if __name__ == "__main__":
    if "--child" in sys.argv:
        child()
    else:
        for kern in (0, 1):
            for threads in (4, 8, 16):
                env = dict(os.environ, RT_COPY_THREADS=str(threads))
                env.pop("RT_H2D_KERNEL", None)
                if kern:
                    env["RT_H2D_KERNEL"] = "1"
                out = subprocess.run([sys.executable, __file__, "--child"],
                                     env=env, capture_output=True, text=True)
                sys.stdout.write(out.stdout or json.dumps(
                    {"error": out.stderr[-300:]}) + "\n")
                sys.stdout.flush()
