"""Tile-major result layouts ([tile of TR rays][L][10][TR]: a workgroup's whole
output contiguous) were within 1 % of SoA at full occupancy (round 2).  Again
with the workgroups resident per CU capped.  Laboratory build."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rayopt_amd as ra
from rayopt_amd import prescriptions as P
from bench import workload_rays

n = 10_000_000
system = ra.system_from_yaml(P.DOUBLE_GAUSS)
y, u = workload_rays(n, 0)
S = len(system) - 1
B = n*(56*S + 48)


def steady(eng, seconds=1.2):
    t_end = time.time() + seconds
    ms = []
    while time.time() < t_end:
        eng.event_record(0)
        for _ in range(10):
            eng.trace(1, 0, True)
        eng.event_record(1)
        ms.append(eng.event_elapsed(0, 1)/10)
    return float(np.median(ms[len(ms)//3:]))


g = ra.GeometricTrace(system)
eng = g.engine
for rep in range(2):
    for tile in (0, 256, 64, 1024):
        eng.set_option("tile_rays", tile)
        g.rays_given(y, u)
        eng.trace(1, 0, True)
        if rep == 0 and tile == 0:
            steady(eng, 4.)
        for pad in (0, 65536, 40960):
            eng.set_option("lds_pad", pad)
            ms = steady(eng)
            print(json.dumps(dict(rep=rep, tile_rays=tile, lds_pad=pad,
                                  launch_ms=ms, TBs=B/ms/1e9)), flush=True)
        eng.set_option("lds_pad", 0)
eng.set_option("tile_rays", 0)
