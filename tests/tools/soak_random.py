"""One-off soak: many more random systems than the test suite (engine vs
oracle at the contract tolerances).  python tests/tools/soak_random.py 60 400"""
import copy
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rayopt_amd as ra
from rayopt_amd.pack import pack_system
from oracle import trace_numpy as tn
from conftest import assert_parity, RTOL_SPHERICAL, RTOL_ASPHERE
from random_systems import random_prescription, random_rays

lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = 0
closed = exact = 0      # traces / of those, bit for bit equal to the oracle
#                         in every value of every array
for seed in range(lo, hi):
    p = random_prescription(seed)
    asph = any("aspherics" in e for e in p["elements"])
    system = ra.system_from_dict(copy.deepcopy(p))
    y, u = random_rays(seed, 5003, p)
    for clip in (True, False):
        g = ra.GeometricTrace(system)
        g.rays_given(y, u)
        g.propagate(clip=clip)
        table, ns = pack_system(system, g.l, g.n[0])
        with np.errstate(all="ignore"):
            want = tn.propagate(table, y, u, clip=clip)
        try:
            same = True
            for rows, b in zip((g.y, g.u, g.i, g.t), want):
                got = np.asarray(rows[1:])
                assert_parity(got, b,
                              RTOL_ASPHERE if asph else RTOL_SPHERICAL,
                              "seed %d" % seed)
                same = same and np.array_equal(got, b, equal_nan=True)
            closed += 1
            exact += same
            if not same:
                print("not bit-identical:", seed, clip, flush=True)
        except AssertionError as e:
            bad += 1
            print("FAIL", seed, clip, str(e)[:200], flush=True)
print("soak %d..%d done, %d failures; %d of %d traces (aspheric systems "
      "included) bit-identical to the oracle" % (lo, hi, bad, exact, closed))
