"""Every row of i stored (alias_i = 0, 80 B per ray-surface op) and the tilted
golden system: steady launch time against the cap on resident workgroups per
CU (resident_lds), one process; run it several times for the spread between
processes."""
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


def steady(clip, seconds=1.3):
    t_end = time.time() + seconds
    ms = []
    while time.time() < t_end:
        eng.event_record(0)
        for _ in range(10):
            eng.trace(1, 0, clip)
        eng.event_record(1)
        ms.append(eng.event_elapsed(0, 1)/10)
    return float(np.median(ms[len(ms)//3:]))


g.propagate(clip=True)
steady(True, 2.)
for alias in (1, 0):
    eng.set_option("alias_i", alias)
    g.rays_given(y, u)
    g.propagate(clip=True)
    res = {}
    for rep in range(2):
        for lds in ((-1, 65536, 40960, 32768, 0) if rep == 0 else
                    (0, 32768, 40960, 65536, -1)):
            eng.set_option("resident_lds", lds)
            res.setdefault(str(lds), []).append(steady(True))
    print(json.dumps({"alias_i": alias, "pid": os.getpid(),
                      "steady_ms_by_resident_lds": res}), flush=True)
