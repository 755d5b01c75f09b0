"""Resident workgroups per CU again, now that the headline trace reads 16.6
instead of 48 B per ray (tile notes): C3 host-seeded and device-generated."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rayopt_amd as ra
from rayopt_amd import prescriptions as P
from bench import workload_rays, FIELD_FRACTIONS, BUNDLE_RADIUS
import digest_cases as dc

n = 10_000_000
system = ra.system_from_yaml(P.DOUBLE_GAUSS)
y, u = workload_rays(n, 0)
g = ra.GeometricTrace(system)
g.rays_given(y, u)
g.propagate(clip=True)
nf = len(FIELD_FRACTIONS)
h = ra.GeometricTrace(system)
h.rays_fields(np.c_[np.zeros(nf), FIELD_FRACTIONS],
              dc.disc_points(n//nf//64*64, 7), P.DOUBLE_GAUSS_PUPIL_Z,
              BUNDLE_RADIUS)
h.propagate(clip=True)


def block(eng, k=10):
    eng.event_record(0)
    for _ in range(k):
        eng.trace(1, 0, True)
    eng.event_record(1)
    return eng.event_elapsed(0, 1)/k


t_end = time.time() + 4.
while time.time() < t_end:
    block(g.engine)
pads = (65536, 53248, 40960, 32768, 24576, 0)
for name, t in (("host-seeded", g), ("device-generated", h)):
    res = {p: [] for p in pads}
    for rep in range(25):
        for p in pads:
            t.engine.set_option("resident_lds", p)
            res[p].append(block(t.engine))
    t.engine.set_option("resident_lds", -1)
    print(json.dumps({"what": name, "median_ms_by_resident_lds": {
        str(p): float(np.median(v)) for p, v in res.items()}}), flush=True)
