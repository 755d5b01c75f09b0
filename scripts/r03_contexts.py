"""Does the launch time depend on WHICH allocation the results live in?  Six
contexts of the same library in one process, the same rays, the same kernel:
steady launch time of each, two rounds; then the device address of each
context's arrays."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rayopt_amd as ra
from rayopt_amd import prescriptions as P
from rayopt_amd.engine import Engine
from rayopt_amd._lib import RT_Y
from bench import workload_rays

n = 10_000_000
system = ra.system_from_yaml(P.DOUBLE_GAUSS)
y, u = workload_rays(n, 0)


def steady(eng, seconds=1.4):
    t_end = time.time() + seconds
    ms = []
    while time.time() < t_end:
        eng.event_record(0)
        for _ in range(10):
            eng.trace(1, 0, True)
        eng.event_record(1)
        ms.append(eng.event_elapsed(0, 1)/10)
    return float(np.median(ms[len(ms)//3:]))


traces = []
for k in range(6):
    g = ra.GeometricTrace(system, engine=Engine(0))
    g.rays_given(y, u)
    g.propagate(clip=True)
    traces.append(g)
steady(traces[0].engine, 2.)
res = [[] for _ in traces]
for rep in range(2):
    order = range(len(traces)) if rep == 0 else reversed(range(len(traces)))
    for k in order:
        res[k].append(steady(traces[k].engine))
for k, g in enumerate(traces):
    # (row 1: handing out row 0 would void the tile notes)
    addr = g.engine.device_ptr(RT_Y, 1)
    print(json.dumps({"context": k, "steady_ms": res[k],
                      "Y_row1_address": hex(addr),
                      "address_mod_2MiB": addr % (2 << 20),
                      "address_mod_1GiB_MiB": (addr % (1 << 30)) >> 20}),
          flush=True)
