#!/usr/bin/env python
"""The memory classes of (most of) the device, piece by piece: reserves a
batch of `rays` (default 2.2*10^8 double-Gauss rays = 229 GB = 214 pieces of
1 GiB) with RT_MI355_PLACE_LOG=1, so that csrc/rt_place.h prints the class of
every piece in the order the driver handed them out (stderr), and reports what
the placement measured (stdout, one JSON line).  Nothing is traced."""
import json
import os
import sys
import time

os.environ["RT_MI355_PLACE_LOG"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import rayopt_amd as ra                             # noqa: E402
from rayopt_amd import prescriptions as P           # noqa: E402

rays = int(float(sys.argv[1])) if len(sys.argv) > 1 else 220_000_000
system = ra.system_from_yaml(P.DOUBLE_GAUSS)
eng = ra.Engine()
g = ra.GeometricTrace(system, engine=eng)
y, u = ra.bundles.disc_bundle(1000, 12., 0., 1, P.DOUBLE_GAUSS_PUPIL_Z)
g.rays_given(y, u)
t0 = time.perf_counter()
eng.reserve(rays)
pl = eng.placement()
pl["reserve_s"] = time.perf_counter() - t0
pl["rays"] = rays
print(json.dumps(pl))
