"""Input rows read with plain vs non-temporal loads under the trace kernel's
store pattern (rt_probe modes 7 / 9 / 8), C3, 10^7 rays."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rayopt_amd as ra
from rayopt_amd import prescriptions as P
from bench import workload_rays
system = ra.system_from_yaml(P.DOUBLE_GAUSS)
y, u = workload_rays(10_000_000, 0)
g = ra.GeometricTrace(system)
g.rays_given(y, u)
for _ in range(60):
    g.propagate(clip=True)
for rep in range(3):
    for mode, name in ((7, "plain loads"), (9, "non-temporal loads"),
                       (10, "2 rays per lane in sequence, inputs up front"),
                       (11, "4 rays per lane in sequence"),
                       (12, "8 rays per lane in sequence"),
                       (13, "input rows in an UNCACHED allocation"),
                       (14, "input rows in a separate ordinary allocation"),
                       (8, "no read")):
        t = []
        for _ in range(10):
            ms, b = g.engine.probe(mode)
            t.append(ms)
        ms = float(np.median(t[2:]))
        print(json.dumps(dict(rep=rep, input=name, ms=ms, GBs=b/ms/1e6)), flush=True)
