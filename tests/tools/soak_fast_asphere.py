"""fast_asphere against the exact path on many random aspheric systems (host
build of the kernel arithmetic; CPU): worst deviation and NaN-mask agreement.

    python tests/tools/soak_fast_asphere.py 0 4000
"""
import copy
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rayopt_amd as ra
from rayopt_amd._lib import F_ASPH, F_FAST
from rayopt_amd.pack import pack_system
import conftest
from random_systems import random_prescription, random_rays

dll = ctypes.CDLL(conftest.build_hostemu())


def emu(table, y0, u0, stop, clip):
    y0, u0 = np.ascontiguousarray(y0), np.ascontiguousarray(u0)
    table = np.ascontiguousarray(table)
    n, rows = len(y0), stop - 1
    out = [np.empty((rows, n, 3)), np.empty((rows, n, 3)),
           np.empty((rows, n, 3)), np.empty((rows, n))]
    rc = dll.emu_trace(ctypes.c_void_p(table.ctypes.data), 1, stop, int(clip),
                       1, ctypes.c_void_p(y0.ctypes.data),
                       ctypes.c_void_p(u0.ctypes.data), ctypes.c_int64(n),
                       *(ctypes.c_void_p(a.ctypes.data) for a in out))
    assert rc == 0
    return out


lo, hi = int(sys.argv[1]), int(sys.argv[2])
found, flips, systems = [], 0, 0
for seed in range(lo, hi):
    p = random_prescription(seed)
    if not any("aspherics" in e for e in p["elements"]):
        continue
    systems += 1
    system = ra.system_from_dict(copy.deepcopy(p))
    y, u = random_rays(seed, 300, p)
    table, _ = pack_system(system, 587.56e-9, system.refractive_index(587.56e-9, 0))
    fast = table.copy()
    fast["flags"] = np.where(fast["flags"] & F_ASPH, fast["flags"] | F_FAST,
                             fast["flags"])
    a = emu(table, y, u, len(table), True)
    b = emu(fast, y, u, len(table), True)
    worst = 0.
    for x, w in zip(b, a):
        for j in range(w.shape[0]):
            flips += int((np.isnan(x[j]) != np.isnan(w[j])).sum())
            fin = np.isfinite(w[j]) & np.isfinite(x[j])
            if fin.any():
                scale = np.abs(w[j][fin]).max()
                worst = max(worst, float(np.abs(x[j][fin] - w[j][fin]).max()/scale))
    found.append((worst, seed))
found.sort(reverse=True)
print("aspheric systems:", systems, " NaN-mask entries that differ:", flips)
for worst, seed in found[:8]:
    print("seed %5d  worst relative deviation fast vs exact %.3g" % (seed, worst))
