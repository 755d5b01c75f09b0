"""TEST INFRASTRUCTURE ONLY -- CPU oracle for ray generation.

numpy restatement of ``InfiniteConjugate.aim`` / ``FiniteConjugate.aim``
(rayopt/conjugates.py:137-166, 236-255) with ``Pupil.map``
(rayopt/pupils.py:97-107, filter=False) and ``sagittal_meridional``
(rayopt/utils.py:106-114), rectilinear projection, for one field point and a
set of pupil coordinates.  Checks the device-side generator
(``rt_generate_rays``).  Pinned against the reference's own ``System.aim`` by
tests/golden/aim_*.npz and live in tests/test_generate.py.
"""
import numpy as np


def sagittal_meridional(u, z):
    s = np.cross(u, z)
    axial = np.all(s == 0, axis=-1)[..., None]
    s = np.where(axial, (1., 0, 0), s)
    m = np.cross(u, s)
    s = s/np.sqrt(np.square(s).sum(-1))[..., None]
    m = m/np.sqrt(np.square(m).sum(-1))[..., None]
    return s, m


def plane_or_sphere_intercept(c, k, y, u):
    """Spheroid.intercept without aspherics (rayopt/elements.py:477-501)."""
    if c == 0:
        return -y[:, 2]/u[:, 2]
    w = np.array([(1, 1, 1 + k)])
    uy = (u*y*w).sum(1) if k else (u*y).sum(1)
    uu = (np.square(u)*w).sum(1) if k else 1.
    yy = (np.square(y)*w).sum(1) if k else np.square(y).sum(1)
    d = c*uy - u[:, 2]
    e = c*uu
    f = c*yy - 2*y[:, 2]
    g = np.sqrt(np.square(d) - e*f)
    return -(d + g)/e


def project(yo, a, p="rectilinear"):
    """InfiniteConjugate.map (rayopt/conjugates.py:208-234)."""
    n = yo.shape[0]
    if p == "rectilinear":
        y = yo*np.tan(a)
        u = np.hstack((y, np.ones((n, 1))))
        u /= np.sqrt(np.square(u).sum(-1))[:, None]
    elif p == "stereographic":
        y = yo*(2*np.tan(a/2))
        r = np.square(y).sum(-1)[:, None]/4
        u = np.hstack((y, 1 - r))/(r + 1)
    elif p == "equisolid":
        y = yo*(2*np.sin(a/2))
        r = np.square(y).sum(-1)[:, None]
        u = np.hstack((y*np.sqrt(1 - r/4), 1 - r/2))
    elif p == "orthographic":
        y = yo*np.sin(a)
        r = np.square(y).sum(-1)[:, None]
        u = np.hstack((y, np.sqrt(1 - r)))
    elif p == "equidistant":
        y = yo*a
        b = np.square(y).sum(-1) > (np.pi/2)**2
        y = np.sin(y)
        z = np.sqrt(np.square(y).sum(-1))
        z = np.where(b, -z, z)[:, None]
        u = np.hstack((y, z))
    return u


def aim_infinite(angle, yo, yp, z, a, c0=0., k0=0., projection="rectilinear"):
    """InfiniteConjugate.aim(yo, yp, z, a, surface=system[0], filter=False);
    ``c0, k0``: curvature/conic of element 0."""
    yo = np.atleast_2d(yo)
    a = np.asarray(a, dtype=float)
    yp = np.atleast_2d(yp)*np.fabs(a).max()            # Pupil.map
    yo, yp = np.broadcast_arrays(yo, yp)
    u = project(yo, angle, projection)
    yz = (0, 0, z)
    y = yz - z*u
    s, m = sagittal_meridional(u, yz)
    y += yp[..., 0, None]*s + yp[..., 1, None]*m
    y += plane_or_sphere_intercept(c0, k0, y, u)[..., None]*u
    return y, u


def aim_finite(radius, telecentric, yo, yp, z, a, sag0=None):
    """FiniteConjugate.aim(yo, yp, z, a, surface=system[0], filter=False);
    ``sag0(y)`` = ``-surface.surface_sag(y)`` of element 0 (None: plane)."""
    yo = np.atleast_2d(yo)
    a = np.arctan2(np.asarray(a, dtype=float), z)
    yp = np.atleast_2d(yp)*np.fabs(a).max()
    yp = z*np.tan(yp)
    yo, yp = np.broadcast_arrays(yo, yp)
    y = np.zeros((yo.shape[0], 3))
    y[..., :2] = -yo*radius
    if sag0 is not None:
        y[..., 2] = sag0(y)
    uz = (0, 0, z)
    u = uz if telecentric else uz - y
    s, m = sagittal_meridional(u, uz)
    u = u + (yp[..., 0, None]*s + yp[..., 1, None]*m)
    u = u/np.sqrt(np.square(u).sum(-1))[..., None]
    if z < 0:
        u = u*-1
    return y, u
