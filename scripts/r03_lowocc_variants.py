"""With the trace kernel capped at two workgroups per CU there is register
room for several rays per lane: do the rejected variants (2 / 4 rays per lane,
XCD-contiguous dealing, non-temporal stores) look different at low occupancy?
Laboratory build, C3 host-seeded, steady state."""
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
g = ra.GeometricTrace(system)
g.rays_given(y, u)
eng = g.engine
g.propagate(clip=True)
S = len(system) - 1
B = n*(56*S + 48)
DEFAULT = dict(rays_per_thread=1, block=256, lds_pad=0, xcd_remap=0,
               nontemporal=0)


def steady(seconds=1.2):
    t_end = time.time() + seconds
    ms = []
    while time.time() < t_end:
        eng.event_record(0)
        for _ in range(10):
            eng.trace(1, 0, True)
        eng.event_record(1)
        ms.append(eng.event_elapsed(0, 1)/10)
    return float(np.median(ms[len(ms)//3:]))


def cell(**kw):
    opts = dict(DEFAULT, **kw)
    for k, v in opts.items():
        eng.set_option(k, v)
    ms = steady()
    print(json.dumps(dict(opts, launch_ms=ms, TBs=B/ms/1e9)), flush=True)


steady(4.)
for rep in range(2):
    cell()
    cell(lds_pad=65536)
    cell(lds_pad=65536, xcd_remap=1)
    cell(lds_pad=65536, nontemporal=1)
    cell(lds_pad=65536, xcd_remap=1, nontemporal=1)
    cell(block=128, lds_pad=32768)
    cell(block=128, lds_pad=32768, xcd_remap=1)
    for r in (2, 4):
        for block, pad in ((256, 0), (256, 65536), (128, 0), (128, 32768),
                           (128, 65536), (64, 16384), (64, 32768)):
            cell(rays_per_thread=r, block=block, lds_pad=pad)
for k, v in DEFAULT.items():
    eng.set_option(k, v)
