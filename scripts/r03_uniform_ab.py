"""Same-process A/B of the tile notes on the headline workload (five
collimated field bundles starting on a plane: y2, u0, u1, u2 uniform across
every 64-ray tile): uniform_input 1 against 0, alternating blocks of
launches, steady state."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rayopt_amd as ra
from rayopt_amd import prescriptions as P
from bench import workload_rays, input_bytes

n = 10_000_000
system = ra.system_from_yaml(P.DOUBLE_GAUSS)
y, u = workload_rays(n, 0)
g = ra.GeometricTrace(system)
g.rays_given(y, u)
eng = g.engine
g.propagate(clip=True)
S = len(system) - 1
print(json.dumps({"input_uniform": eng.input_uniform(),
                  "input_bytes_per_ray": input_bytes(eng, n)[0]/n}))
want = np.array(np.asarray(g.y[-1]))


def block(k=10):
    eng.event_record(0)
    for _ in range(k):
        eng.trace(1, 0, True)
    eng.event_record(1)
    return eng.event_elapsed(0, 1)/k


t_end = time.time() + 4.
while time.time() < t_end:
    block()
res = {0: [], 1: []}
for rep in range(60):
    for on in (1, 0):
        eng.set_option("uniform_input", on)
        res[on].append(block())
eng.set_option("uniform_input", 1)
g.propagate(clip=True)
same = bool(np.array_equal(np.asarray(g.y[-1]), want, equal_nan=True))
on, off = (float(np.median(res[k])) for k in (1, 0))
rb = input_bytes(eng, n)[0]
print(json.dumps({"uniform_input_1_ms": on, "uniform_input_0_ms": off,
                  "gain": off/on - 1.,
                  "TBs_on": (n*56*S + rb)/on/1e9,
                  "TBs_off": n*(56*S + 48)/off/1e9,
                  "image_rows_identical": same}))
