#!/usr/bin/env python
"""C3' (per-ray launch directions: 40 B/ray read among the saturated writes)
under other caps on the resident workgroups per CU (option "resident_lds"),
one context, the settled launch time of each."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import numpy as np                                  # noqa: E402
import rayopt_amd as ra                             # noqa: E402
from rayopt_amd import prescriptions as P           # noqa: E402
import bench_legs as legs                           # noqa: E402

s3 = ra.system_from_yaml(P.DOUBLE_GAUSS)
n = 10_000_000
y, u = legs.workload_rays(n, 7)
rng = np.random.default_rng(3)
u[:, 0] += 1e-7*rng.standard_normal(n)
u[:, 1] += 1e-7*rng.standard_normal(n)
u[:, 2] = np.sqrt(1. - u[:, 0]**2 - u[:, 1]**2)
g = ra.GeometricTrace(s3)
g.rays_given(y, u)
out = {"GBps": round(g.engine.placement()["store_pattern_GBps"])}
for rep in range(2):
    for wg in (-1, 4, 6, 8, 10, 0):
        g.engine.set_option("resident_lds", -1 if wg < 0 else 0 if wg == 0 else min(65536, 163840//wg))
        out["%s wg/CU (%d)" % ("default" if wg < 0 else wg, rep)] = round(
            legs.kernel_ms_of(g, True, settle_s=.2, dwell_s=.3), 4)
print(json.dumps(out))
yh, uh = legs.workload_rays(n, 0)
g.rays_given(yh, uh)
out = {"headline GBps": round(g.engine.placement()["store_pattern_GBps"])}
for wg in (-1, 4, 6, 8, 10, 0):
    g.engine.set_option("resident_lds", -1 if wg < 0 else 0 if wg == 0 else min(65536, 163840//wg))
    out["%s wg/CU" % ("default" if wg < 0 else wg)] = round(
        legs.kernel_ms_of(g, True, settle_s=.2, dwell_s=.3), 4)
print(json.dumps(out))
