#!/usr/bin/env python
"""Does the store pattern's lottery depend on the ROW PITCH (round 6)?  C2's
shape -- Cooke triplet, three wavelength groups in one launch -- at ray counts
around 3 x 10^6 (the pitch of a one-block batch is the ray count), several
contexts each: the sets of pieces tried and the settled launch time per ray.

    python scripts/pitch_lab.py contexts n1 n2 ...
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import rayopt_amd as ra                             # noqa: E402
from rayopt_amd import prescriptions as P           # noqa: E402
import digest_cases as dc                           # noqa: E402
import bench_legs as legs                           # noqa: E402

reps = int(sys.argv[1])
s2 = ra.system_from_yaml(P.COOKE % dict(
    air="air", sk16="SCHOTT-SK|N-SK16", f2="SCHOTT-F|N-F2"))
ls = [587.56e-9, 656.27e-9, 486.13e-9]
for n in (int(a) for a in sys.argv[2:]):
    per = n//3//64*64
    y, u = dc.bundle(per, 5.5, 5., 0)
    for k in range(reps):
        g = ra.GeometricTrace(s2)
        g.rays_given(y, u, ls)
        ms = legs.kernel_ms_of(g, True, settle_s=.15, dwell_s=.2)
        pl = g.engine.placement()
        print(json.dumps({
            "rays": 3*per, "ld": g.engine.ld, "trace_ms": round(ms, 4),
            "ns_per_kray": round(ms*1e6/(3*per), 3),
            "per_class": pl["per_class"],
            "sets": [round(v) for v in
                     pl["store_pattern_GBps_per_piece_set"]]}), flush=True)
        del g
