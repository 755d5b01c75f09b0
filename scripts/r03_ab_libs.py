"""Same-process A/B of the shipped library against the laboratory build (the
round-2 kernel with its measurement parameters and template variants): the
headline workload (C3, 10^7 rays, host-seeded, clip) on two contexts of one
process, launches alternating, so that box / process plateaus cancel."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rayopt_amd as ra
from rayopt_amd import prescriptions as P, _build
from rayopt_amd.engine import Engine
from bench import workload_rays

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
system = ra.system_from_yaml(P.DOUBLE_GAUSS)
y, u = workload_rays(n, 0)
traces = {}
for name, path in (("shipped", _build.LIB), ("laboratory", _build.PROBES_LIB)):
    g = ra.GeometricTrace(system, engine=Engine(0, lib_path=path))
    g.rays_given(y, u)
    traces[name] = g
for _ in range(40):
    for g in traces.values():
        g.propagate(clip=True)
S = len(system) - 1
res = {k: [] for k in traces}
for rep in range(60):
    for name, g in traces.items():
        g.propagate(clip=True)
        res[name].append(g.kernel_ms())
a, b = (np.asarray(traces[k].y[-1]) for k in ("shipped", "laboratory"))
out = {k: {"median_ms": float(np.median(v)), "min_ms": float(np.min(v)),
           "TBs": n*(56*S + 48)/float(np.median(v))/1e9}
       for k, v in res.items()}
out["ratio_shipped_over_laboratory"] = out["shipped"]["median_ms"] / \
    out["laboratory"]["median_ms"]
out["image_rows_identical"] = bool(np.array_equal(a, b, equal_nan=True))
print(json.dumps(out))
