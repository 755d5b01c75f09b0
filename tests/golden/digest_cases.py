"""Full-size cases whose reference results are pinned by DIGEST: the BASELINE
configs at sizes the reference finishes in seconds here, every value of
``y, u, i, t`` of every traced row hashed (SHA-256) instead of stored.

The inputs must come out the same to the last bit on every machine, so they
use no vectorised transcendental function (numpy's SIMD sin / cos may differ
by an ulp between CPUs): launch points by rejection from the unit square
(multiplications and comparisons only), the collimated direction of a field
from scalar libm calls.  The digest of the inputs is stored next to the
digests of the results; a test first checks that it reproduces."""
import hashlib
import math

import numpy as np

import rayopt_amd as ra
from rayopt_amd import prescriptions as P


def disc_points(n, seed):
    """n points uniform in the unit disc, no trigonometry: rejection from the
    square (PCG64 doubles, * and <= only)."""
    rng = np.random.default_rng(seed)
    out = np.empty((n, 2))
    have = 0
    while have < n:
        xy = rng.random((2*(n - have) + 64, 2))*2. - 1.
        xy = xy[(xy*xy).sum(1) <= 1.]
        k = min(len(xy), n - have)
        out[have:have + k] = xy[:k]
        have += k
    return out


def bundle(n, radius, theta_deg, seed, z_pupil=0.):
    th = math.radians(theta_deg)
    y = np.zeros((n, 3))
    y[:, :2] = disc_points(n, seed)*radius
    y[:, 1] -= z_pupil*math.tan(th)
    u = np.zeros((n, 3))
    u[:, 1] = math.sin(th)
    u[:, 2] = math.cos(th)
    return y, u


def bundles(n, radius, thetas, seed, z_pupil=0.):
    per = n//len(thetas)
    parts = [bundle(per, radius, th, seed + k, z_pupil)
             for k, th in enumerate(thetas)]
    return (np.concatenate([p[0] for p in parts]),
            np.concatenate([p[1] for p in parts]))


def cases(heavy=True):
    """``heavy=False`` leaves out the 10^7-ray case (its rays alone are
    480 MB)."""
    for case in _cases():
        if heavy or not case.get("heavy"):
            yield case


def _cases():
    """``rays`` is a function: the arrays are built when a test asks."""
    fields = [f*P.DOUBLE_GAUSS_FIELD_DEG for f in (0, .35, .5, .7, 1.)]
    pz = P.DOUBLE_GAUSS_PUPIL_Z
    yield dict(name="C1_singlet_1e4", yaml=P.SINGLET, l=587.56e-9, clip=True,
               rays=lambda: bundle(10**4, 8., 0., 0))
    for l in (587.56e-9, 656.27e-9, 486.13e-9):
        yield dict(name="C2_cooke_1e6_%.0fnm" % (l*1e9), yaml=P.cooke(l), l=l,
                   clip=True, rays=lambda: bundle(10**6, 5.5, 5., 0))
    yield dict(name="C3_double_gauss_1e6_5_fields", yaml=P.DOUBLE_GAUSS,
               l=587.56e-9, clip=True,
               rays=lambda: bundles(10**6, 17., fields, 0, pz))
    yield dict(name="C3_double_gauss_1e6_unclipped", yaml=P.DOUBLE_GAUSS,
               l=587.56e-9, clip=False,
               rays=lambda: bundles(10**6, 17., fields, 0, pz))
    # the headline workload at its full size: 1.2*10^8 ray-surface ops, 10
    # values each (the reference needs ~1 min and ~25 GB for it)
    yield dict(name="C3_double_gauss_1e7_5_fields", yaml=P.DOUBLE_GAUSS,
               l=587.56e-9, clip=True, heavy=True,
               rays=lambda: bundles(10**7, 17., fields, 0, pz))

    def two_fields():
        y0, u0 = bundle(10**4, .6, 0., 1)
        y1, u1 = bundle(10**4, .6, 17.5, 2)
        y1[:, 1] -= .5*math.tan(math.radians(17.5))
        return np.concatenate([y0, y1]), np.concatenate([u0, u1])
    yield dict(name="C4_asphere_2e4_two_fields", yaml=P.ASPHERE_PHONE,
               l=587.56e-9, clip=True, rays=two_fields)
    yield dict(name="torture_2e5", yaml=P.TORTURE, l=587.56e-9, clip=True,
               rays=lambda: bundle(2*10**5, 12., 2., 3))


GENERATED = dict(name="generated_double_gauss_5_fields_1e6",
                 yaml=P.DOUBLE_GAUSS, l=587.56e-9, clip=True,
                 fields=[(0., 0.), (0., .35), (0., .5), (0., .7), (0., 1.)],
                 z=P.DOUBLE_GAUSS_PUPIL_Z, a=17.,
                 pupil=lambda: disc_points(200_000, 77)*.95)
"""Bundles the reference builds with System.aim(yo, yp, z, a, filter=False)
(rayopt/system.py:504, rayopt/conjugates.py:236-255) field by field and hands
to rays_given; here built on the device (rays_fields), first trace fused with
the generation, re-traces rebuilding the rays."""


def digest_rows(rows_of):
    """SHA-256 over rows 1..L-1 of y, u, i (each (N,3) C-contiguous) and t
    ((N,)), in that order; ``rows_of(name, j)`` returns the row."""
    h = hashlib.sha256()
    j = 1
    while True:
        try:
            blocks = [rows_of(k, j) for k in "yuit"]
        except IndexError:
            break
        for b in blocks:
            h.update(canonical_nan(b).tobytes())
        j += 1
    return h.hexdigest()


def digest_inputs(y, u):
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(y).tobytes())
    h.update(np.ascontiguousarray(u).tobytes())
    return h.hexdigest()


def canonical_nan(a):
    """NaN payloads and signs are not part of the contract: every NaN is
    hashed as the same bit pattern (0.0 and -0.0 stay distinct)."""
    a = np.array(a, dtype=np.float64)
    a[np.isnan(a)] = np.nan
    return a
