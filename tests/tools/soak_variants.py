"""One-off soak on the GPU: every kernel variant against the default kernel
on random systems (tilts, conics, aspheres, mirrors, vignetting).

    python tests/tools/soak_variants.py 0 1500

For each seed the default kernel's rows are the baseline (itself soaked
against the oracle by soak_random.py).  Bit for bit equal to it, NaN masks
included (payloads aside), must be: 2 and 4 rays per lane, non-temporal
stores, XCD dealing, 512-ray workgroups, every `i` row materialised, the
compacting kernel forced on (asking at every / every 3rd element), a trace
in two halves (propagate to k, then from k), kept-row subsets, and the same
rays traced as two wavelength groups with equal tables.  The fast asphere
arithmetic must keep the NaN masks and stay within 1e-8."""
import copy
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rayopt_amd as ra
from random_systems import random_prescription, random_rays

DEFAULTS = dict(rays_per_thread=1, nontemporal=0, xcd_remap=0, block=256,
                alias_i=1, compact=0, compact_every=4, fast_asphere=0)
VARIANTS = [
    dict(rays_per_thread=2), dict(rays_per_thread=4), dict(nontemporal=1),
    dict(xcd_remap=1), dict(block=512), dict(alias_i=0),
    dict(compact=2, compact_every=1), dict(compact=2, compact_every=3),
]


def rows_of(g, L):
    return [np.array(np.asarray(rows[j])) for rows in (g.y, g.u, g.i, g.t)
            for j in range(L)]


def same(a, b):
    return all(np.array_equal(x, y, equal_nan=True) for x, y in zip(a, b))


def main():
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 3001
    bad = 0
    for seed in range(lo, hi):
        p = random_prescription(seed)
        system = ra.system_from_dict(copy.deepcopy(p))
        L = len(system)
        y, u = random_rays(seed, n, p)
        clip = bool(seed % 3)
        g = ra.GeometricTrace(system)
        eng = g.engine
        for k, v in DEFAULTS.items():
            eng.set_option(k, v)
        g.rays_given(y, u)
        g.propagate(clip=clip)
        base = rows_of(g, L)

        def fail(what):
            nonlocal bad
            bad += 1
            print("FAIL seed %d clip %s: %s" % (seed, clip, what), flush=True)

        for v in VARIANTS:
            for k, val in dict(DEFAULTS, **v).items():
                eng.set_option(k, val)
            g.rays_given(y, u)
            g.propagate(clip=clip)
            if not same(rows_of(g, L), base):
                fail(v)
        for k, v in DEFAULTS.items():
            eng.set_option(k, v)
        # two halves
        k = 1 + seed % (L - 1)
        g.rays_given(y, u)
        g.propagate(stop=k, clip=clip)
        g.propagate(start=k, clip=clip)
        if not same(rows_of(g, L), base):
            fail("two halves at %d" % k)
        # kept rows only
        keep = sorted({0, L - 1, 1 + seed % (L - 1)})
        g.rays_given(y, u)
        g.propagate(clip=clip, keep=keep)
        for a, rows in enumerate((g.y, g.u, g.i, g.t)):
            for j in keep:
                if not np.array_equal(np.asarray(rows[j]), base[a*L + j],
                                      equal_nan=True):
                    fail("keep %s row %d array %d" % (keep, j, a))
        g.propagate(clip=clip)          # all rows again
        # the same rays at "two wavelengths" (the same one twice):
        # rays_given(y, u, l=[...]) makes one ray group per wavelength, each
        # marched through its own table
        m = (n//64)*64
        if m:
            g2 = ra.GeometricTrace(system)
            l = system.wavelengths[0]
            g2.rays_given(y[:m], u[:m], l=[l, l])
            g2.propagate(clip=clip)
            for a, rows in enumerate((g2.y, g2.u, g2.i, g2.t)):
                for j in range(1, L):
                    got = np.asarray(rows[j])
                    want = base[a*L + j][:m]
                    if not (np.array_equal(got[:m], want, equal_nan=True)
                            and np.array_equal(got[m:], want,
                                               equal_nan=True)):
                        fail("groups row %d array %d" % (j, a))
                        break
        # fast asphere
        if any("aspherics" in e for e in p["elements"]):
            eng = g.engine
            eng.set_option("fast_asphere", 1)
            g.rays_given(y, u)
            g.propagate(clip=clip)
            fast = rows_of(g, L)
            eng.set_option("fast_asphere", 0)
            for a, b in zip(fast, base):
                if not np.array_equal(np.isnan(a), np.isnan(b)):
                    fail("fast asphere NaN mask")
                    break
                fin = np.isfinite(b)
                if fin.any():
                    scale = max(1., np.abs(b[fin]).max())
                    if np.abs(a[fin] - b[fin]).max() > 1e-8*scale:
                        fail("fast asphere %.3g" % (
                            np.abs(a[fin] - b[fin]).max()/scale))
                        break
        if seed % 100 == 0:
            print("seed", seed, "failures so far", bad, flush=True)
    print("variant soak %d..%d (%d rays): %d failures" % (lo, hi, n, bad))


if __name__ == "__main__":
    main()
