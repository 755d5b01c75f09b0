"""Deterministic synthetic ray bundles.

Launch points are uniform in a disc on the z=0 plane of surface 0
(``r = R sqrt(U1)``, ``phi = 2 pi U2``, ``np.random.default_rng(seed)``), the
direction is collimated per field, ``u = (0, sin theta, cos theta)``; the
bundle centre is shifted by ``-z_pupil tan(theta)`` so that it passes through
the entrance pupil.  Host-side, O(N); used by tests, fixtures and bench.py.
"""
import numpy as np


def disc_bundle(n, radius, theta_deg=0., seed=0, z_pupil=0.):
    """(y, u) as (n,3) float64 arrays."""
    rng = np.random.default_rng(seed)
    r = radius*np.sqrt(rng.random(n))
    phi = 2*np.pi*rng.random(n)
    theta = np.radians(theta_deg)
    y = np.zeros((n, 3))
    y[:, 0] = r*np.cos(phi)
    y[:, 1] = r*np.sin(phi) - z_pupil*np.tan(theta)
    u = np.zeros((n, 3))
    u[:, 1] = np.sin(theta)
    u[:, 2] = np.cos(theta)
    return y, u


def multi_field_bundle(n, radius, thetas_deg, seed=0, z_pupil=0.):
    """``len(thetas_deg)`` equal sub-bundles concatenated (config C3: five
    field points in one batch)."""
    per = n//len(thetas_deg)
    ys, us = [], []
    for k, th in enumerate(thetas_deg):
        m = per if k < len(thetas_deg) - 1 else n - per*(len(thetas_deg) - 1)
        y, u = disc_bundle(m, radius, th, seed + k, z_pupil)
        ys.append(y)
        us.append(u)
    return np.concatenate(ys), np.concatenate(us)
