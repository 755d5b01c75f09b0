#!/usr/bin/env python
"""Row pitches that are multiples of 2^k rays (round 6): per context one set
of pieces, the same ray count laid out with the pitch rounded up to multiples
of 256 ... 2^17 rays (option "pitch_rays"), the store pattern's GB/s of each;
for C2's shape (3 x 10^6 rays, one block) and the headline's (10^7 rays, two
blocks).

    python scripts/pitch_lab3.py contexts
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
s3 = ra.system_from_yaml(P.DOUBLE_GAUSS)
y, u = dc.bundle(1024, 5.5, 5., 0)
quanta = [0, 1024, 4096, 16384, 32768, 65536, 131072, 262144]
for ctx in range(int(sys.argv[1])):
    for name, system, n in (("C2", s2, 3_000_000), ("C3", s3, 10_000_000)):
        eng = ra.Engine()
        g = ra.GeometricTrace(system, engine=eng)
        g.rays_given(y, u)              # uploads the table
        eng.set_option("pitch_rays", quanta[-1])
        eng.reserve(n + 2*quanta[-1])   # room for every pitch
        pl = eng.placement()
        rec = {"shape": name, "context": ctx, "per_class": pl["per_class"],
               "sets": [round(v) for v in
                        pl["store_pattern_GBps_per_piece_set"]], "by_q": {}}
        for q in quanta + quanta[:3]:
            eng.set_option("pitch_rays", q)
            eng.reserve(n + 64)         # (a new layout every time)
            eng.reserve(n)
            rec["by_q"].setdefault(str(q), []).append(
                [eng.blocks()[1], round(eng.placement()["store_pattern_GBps"])])
        print(json.dumps(rec), flush=True)
        eng.close()
