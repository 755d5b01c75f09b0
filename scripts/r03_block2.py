"""Two workgroups per CU: which workgroup size?  Laboratory build (lab kernels
read all 48 B/ray of row 0)."""
import json, os, sys, time
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


def block(k=10):
    eng.event_record(0)
    for _ in range(k):
        eng.trace(1, 0, True)
    eng.event_record(1)
    return eng.event_elapsed(0, 1)/k


eng.set_option("lds_pad", 65536)
t_end = time.time() + 4.
while time.time() < t_end:
    block()
sizes = (256, 192, 320, 384, 448, 128)
res = {b: [] for b in sizes}
for rep in range(25):
    for b in sizes:
        eng.set_option("block", b)
        res[b].append(block())
print(json.dumps({"lds_pad": 65536, "median_ms_by_block": {
    str(b): float(np.median(v)) for b, v in res.items()}}))
