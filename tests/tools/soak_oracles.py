"""One-off soak on the CPU (needs /root/reference): the plain-C oracle and the
numpy oracle against the unmodified reference on many more random systems
than the suite.  python tests/tools/soak_oracles.py 1000 3000"""
import copy
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from rayopt_amd.pack import pack_system
from oracle import trace_numpy as tn
from oracle import build_c, refshim
from conftest import assert_parity
from random_systems import random_prescription, random_rays

ro = refshim.load()
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(lo, hi):
    p = random_prescription(seed)
    asph = any("aspherics" in e for e in p["elements"])
    tilt = any("angles" in e or "direction" in e for e in p["elements"])
    y, u = random_rays(seed, 32 if asph else 300, p)
    ref_sys = ro.System(**copy.deepcopy(p))
    for clip in (True, False):
        g = ro.GeometricTrace(ref_sys)
        g.rays_given(y, u)
        with np.errstate(all="ignore"):
            g.propagate(clip=clip)
            want = (g.y[1:], g.u[1:], g.i[1:], g.t[1:])
            table, ns = pack_system(ref_sys, g.l, g.n[0])
            for name, got in (("numpy", tn.propagate(table, y, u, clip=clip)),
                              ("C", build_c.propagate(table, y, u,
                                                      clip=clip))):
                try:
                    for a, b in zip(got, want):
                        if tilt and name == "C":
                            # 3x3 products: numpy calls BLAS, C adds in
                            # index order; the last-bit difference is
                            # amplified by ill-conditioned geometry (seen:
                            # 2.4e-11) -- the contract is 1e-10
                            assert_parity(a, b, 1e-10, name)
                        elif asph:
                            assert_parity(a, b, 1e-11, name)
                        else:
                            assert np.array_equal(a, b, equal_nan=True)
                except AssertionError as e:
                    bad += 1
                    print("FAIL", name, seed, clip, str(e)[:160], flush=True)
    if seed % 200 == 0:
        print("seed", seed, "failures so far", bad, flush=True)
print("soak %d..%d done, %d failures" % (lo, hi, bad))
