"""Which random TILTED systems put the kernel arithmetic furthest from the
reference?  (CPU, needs /root/reference.)  The kernel's per-ray header compiled
for the host (tests/hostemu) against the live reference on the seeded random
systems of tests/random_systems.py that contain tilted or decentred elements;
prints the worst seeds.  The five worst are committed as goldens
(tests/golden/cases.py: tilted_seed_*).

    python tests/tools/soak_tilted.py 1000 9000
"""
import copy
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from rayopt_amd.pack import pack_system
from oracle import refshim
import conftest
from random_systems import random_prescription, random_rays

ro = refshim.load()
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


def deviation(got, want):
    worst = 0.
    for a, b in zip(got, want):
        for j in range(b.shape[0]):
            if not np.array_equal(np.isnan(a[j]), np.isnan(b[j])):
                return np.inf
            fin = np.isfinite(b[j])
            if fin.any():
                scale = np.abs(b[j][fin]).max()
                worst = max(worst, float(np.abs(a[j][fin] - b[j][fin]).max()
                                         / scale))
    return worst


lo, hi = int(sys.argv[1]), int(sys.argv[2])
found = []
for seed in range(lo, hi):
    p = random_prescription(seed)
    if not any("angles" in e or "direction" in e for e in p["elements"]):
        continue
    if any("aspherics" in e for e in p["elements"]):
        continue                        # Newton path: its own contract
    y, u = random_rays(seed, 300, p)
    ref_sys = ro.System(**copy.deepcopy(p))
    g = ro.GeometricTrace(ref_sys)
    g.rays_given(y, u)
    with np.errstate(all="ignore"):
        g.propagate(clip=True)
    table, ns = pack_system(ref_sys, g.l, g.n[0])
    dev = deviation(emu(table, y, u, len(ref_sys), True),
                    (g.y[1:], g.u[1:], g.i[1:], g.t[1:]))
    found.append((dev, seed))
found.sort(reverse=True)
print("tilted spherical systems checked:", len(found))
for dev, seed in found[:12]:
    print("seed %5d  worst relative deviation %.3g" % (seed, dev))
