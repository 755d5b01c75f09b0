"""Isolated launches (50 ms of idle before each: clocks high, power filter
relaxed) against back-to-back ones (the power limiter engaged), at two and at
seven workgroups per CU: is the steady state a power-limited state?"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rayopt_amd as ra
from rayopt_amd import prescriptions as P
from bench import workload_rays, Telemetry

n = 10_000_000
system = ra.system_from_yaml(P.DOUBLE_GAUSS)
y, u = workload_rays(n, 0)
g = ra.GeometricTrace(system)
g.rays_given(y, u)
eng = g.engine
g.propagate(clip=True)
eng.sync()
out = {}
for lds in (-1, 0):
    eng.set_option("resident_lds", lds)
    iso = []
    for _ in range(40):
        time.sleep(.05)
        eng.trace(1, 0, True)
        iso.append(eng.kernel_ms())
    # pairs: the second launch of a pair follows the first immediately
    pair = []
    for _ in range(30):
        time.sleep(.05)
        eng.trace(1, 0, True)
        eng.trace(1, 0, True)
        pair.append(eng.kernel_ms())
    t_end = time.time() + 3.
    while time.time() < t_end:
        eng.trace(1, 0, True)
    b2b = []
    for _ in range(60):
        eng.trace(1, 0, True)
        b2b.append(eng.kernel_ms())
    out["resident_lds=%d" % lds] = {
        "isolated_ms": float(np.median(iso)),
        "second_of_a_pair_ms": float(np.median(pair)),
        "back_to_back_steady_ms": float(np.median(b2b))}
    time.sleep(1.)
print(json.dumps(out))
