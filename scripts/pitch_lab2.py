#!/usr/bin/env python
"""The store pattern of ONE set of pieces at many row pitches (round 6): a
context allocates for the largest ray count, then rt_reserve re-lays the same
buffer out for every other count (a reused buffer is only measured) and the
pattern's GB/s is read back.  C2's system (9 elements) in one block.

    python scripts/pitch_lab2.py contexts
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

s2 = ra.system_from_yaml(P.COOKE % dict(
    air="air", sk16="SCHOTT-SK|N-SK16", f2="SCHOTT-F|N-F2"))
y, u = dc.bundle(1024, 5.5, 5., 0)
pitches = [3_000_000 + 64*k for k in (0, 1, 3, 7, 15, 31, 100, 228, 1000, 1563,
                                      2275, 3000)] + \
    [2_949_120, 3_014_592, 3_100_032, 3_145_728, 3_211_200]
top = max(pitches) + 4096
for ctx in range(int(sys.argv[1])):
    eng = ra.Engine()
    g = ra.GeometricTrace(s2, engine=eng)
    g.rays_given(y, u)                  # uploads the table
    eng.reserve(top)
    pl = eng.placement()
    rec = {"context": ctx, "per_class": pl["per_class"],
           "sets_at_top": [round(v) for v in
                           pl["store_pattern_GBps_per_piece_set"]],
           "GBps_by_pitch": {}}
    for n in pitches:
        eng.reserve(n)
        assert eng.ld == n
        rec["GBps_by_pitch"][n] = round(
            eng.placement()["store_pattern_GBps"])
    # and once more: how repeatable is one pitch in one set of pieces?
    again = {}
    for n in pitches[:4]:
        eng.reserve(n)
        again[n] = round(eng.placement()["store_pattern_GBps"])
    rec["again"] = again
    print(json.dumps(rec), flush=True)
    eng.close()
