"""Where the time of a row download goes (round 5): the image row of 10^7
rays (240 MB) to a touched numpy array -- batch in one block and in two, the
staging copy on 1 / 4 / 8 threads -- next to rt_copy_to_host of the same
bytes and a pinned hipMemcpy (the ceiling).  One JSON line per case.

    python scripts/hostpath.py
"""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def case(block_rays, threads, noncoherent=0):
    env = dict(os.environ, RT_COPY_THREADS=str(threads),
               RT_HOSTPATH_BLOCK=str(block_rays),
               RT_PIN_NONCOHERENT=str(noncoherent))
    out = subprocess.run([sys.executable, __file__, "--child"], env=env,
                         capture_output=True, text=True)
    sys.stdout.write(out.stdout)
    if out.returncode:
        sys.stdout.write(json.dumps({"error": out.stderr[-300:]}) + "\n")


def child():
    import rayopt_amd as ra
    from rayopt_amd import prescriptions as P
    from rayopt_amd._lib import RT_Y
    from bench import workload_rays
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    L = len(system)
    n = 10_000_000
    y, u = workload_rays(n, 0)
    g = ra.GeometricTrace(system)
    block = int(os.environ["RT_HOSTPATH_BLOCK"])
    if block:
        g.engine.set_option("block_rays", block)
    g.rays_given(y, u)
    g.propagate(clip=True)
    eng = g.engine
    dst = np.empty((1, 3, n))
    dst[:] = 0.
    t = []
    for _ in range(5):
        t0 = time.perf_counter()
        eng.download(RT_Y, L - 1, L, out=dst)
        t.append((time.perf_counter() - t0)*1e3)
    flat = np.empty(3*n)
    flat[:] = 0.
    ptr = eng.device_ptr(RT_Y, L - 1)
    nb = eng.blocks()
    seg = min(n, nb[1])*8       # one component of block 0: contiguous
    tc = []
    for _ in range(5):
        t0 = time.perf_counter()
        eng.lib.rt_copy_to_host(eng.ctx, flat.ctypes.data, ptr, seg)
        tc.append((time.perf_counter() - t0)*1e3)
    print(json.dumps({"blocks": nb[0], "copy_threads": int(
        os.environ["RT_COPY_THREADS"]), "noncoherent_staging": int(
        os.environ["RT_PIN_NONCOHERENT"]), "download_image_row_ms": t,
        "GBps_best": 24*n/min(t)/1e6,
        "copy_to_host_one_segment_ms": tc, "segment_bytes": seg,
        "segment_GBps_best": seg/min(tc)/1e6}), flush=True)


if __name__ == "__main__":
    if "--child" in sys.argv:
        child()
    else:
        for noncoherent in (0, 1):
            for threads in (1, 4, 8, 16):
                case(0, threads, noncoherent)
