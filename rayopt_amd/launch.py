"""Per-field launch frames for on-device ray generation.

For every field point the host evaluates the O(1) frame that the reference's
``Conjugate.aim`` builds before it broadcasts over the pupil coordinates
(rayopt/conjugates.py:137-166 finite, :236-255 infinite, all five projections
of InfiniteConjugate.map :208-234; sagittal/meridional
unit vectors rayopt/utils.py:106-114); the GPU expands it over the pupil
grid (``rt_generate_rays``).  ``system.object`` may be this package's
``Conjugate`` or a reference conjugate object (``finite``, ``angle`` /
``radius``, ``projection``, ``pupil.telecentric`` are read).
"""
import numpy as np

from ._lib import FIELD_DTYPE, AIM_SEED_DTYPE


def _unit(v):
    return v/np.sqrt(np.square(v).sum(-1))


def _frame(u, z):
    """Sagittal and meridional unit vectors of direction ``u`` about the
    axis ``(0,0,z)``."""
    axis = np.array((0., 0., z))
    s = np.cross(u, axis)
    if np.all(s == 0):
        s = np.array((1., 0., 0.))
    m = np.cross(u, s)
    return _unit(s), _unit(m)


def _sag0(element, y):
    """-surface_sag of element 0 at the object point(s) ``y`` ((3,) or
    (F,3)) (z of the object surface), Spheroid.surface_sag
    rayopt/elements.py:440-455."""
    y = np.asarray(y)
    c = getattr(element, "curvature", 0.)
    asph = getattr(element, "aspherics", None)
    if not c and asph is None:
        return -y[..., 2]
    r2 = y[..., 0]*y[..., 0] + y[..., 1]*y[..., 1]
    e = y[..., 2]
    if c:
        k = getattr(element, "conic", 0.)
        e = e - c*r2/(1 + np.sqrt(1 - (1 + k)*c**2*r2))
    if asph is not None:
        d = 0.
        for ai in reversed(asph):
            d += ai
            d *= r2
        e = e - d
    return -e


def _direction(yo, angle, projection):
    """Unit directions of the fields ``yo`` (fractional, (F,2) or (2,)) for
    an object at infinity with semi-angle ``angle`` (InfiniteConjugate.map,
    rayopt/conjugates.py:208-234); every field at once, entry by entry the
    arithmetic of the one-field call."""
    yo = np.asarray(yo, dtype=float)
    u = np.empty(yo.shape[:-1] + (3,))
    if projection == "rectilinear":
        u[..., :2] = yo*np.tan(angle)
        u[..., 2] = 1.
        return u/np.sqrt(np.square(u).sum(-1))[..., None]
    if projection == "stereographic":
        y = yo*(2*np.tan(angle/2))
        r = np.square(y).sum(-1)/4
        u[..., :2] = y
        u[..., 2] = 1 - r
        return u/(r + 1)[..., None]
    if projection == "equisolid":
        y = yo*(2*np.sin(angle/2))
        r = np.square(y).sum(-1)
        u[..., :2] = y*np.sqrt(1 - r/4)[..., None]
        u[..., 2] = 1 - r/2
        return u
    if projection == "orthographic":
        y = yo*np.sin(angle)
        u[..., :2] = y
        u[..., 2] = np.sqrt(1 - np.square(y).sum(-1))
        return u
    if projection == "equidistant":
        y = yo*angle
        behind = np.square(y).sum(-1) > (np.pi/2)**2
        y = np.sin(y)
        z = np.sqrt(np.square(y).sum(-1))
        u[..., :2] = y
        u[..., 2] = np.where(behind, -z, z)
        return u
    raise NotImplementedError("projection %r" % projection)


def _telecentric(obj):
    pupil = obj.pupil
    if isinstance(pupil, dict):
        return bool(pupil.get("telecentric", False))
    return bool(pupil.telecentric)


def _cross(a, b):
    """``np.cross`` of (...,3) arrays, the same products and differences,
    without its axis shuffling."""
    out = np.empty(np.broadcast(a, b).shape)
    out[..., 0] = a[..., 1]*b[..., 2] - a[..., 2]*b[..., 1]
    out[..., 1] = a[..., 2]*b[..., 0] - a[..., 0]*b[..., 2]
    out[..., 2] = a[..., 0]*b[..., 1] - a[..., 1]*b[..., 0]
    return out


_SIGNS = np.array(((-1., -1.), (1., 1.)))


def _apertures(a, nf):
    """Pupil apertures as (F,2,2): from a scalar radius, (F,) radii, one
    (2,2) ``[[-sag,-mer],[+sag,+mer]]`` or (F,2,2)."""
    a = np.asarray(a, dtype=float)
    if a.ndim <= 1:     # scalar radius, or one radius per field
        a = a[..., None, None]*_SIGNS
    return np.broadcast_to(a, (nf, 2, 2))


def field_frames(system, yo, z, a, groups=None):
    """``FIELD_DTYPE`` array, one entry per row of ``yo`` (F,2) fractional
    object coordinates.  ``z``: pupil distance(s) from the vertex of element
    0, scalar or (F,); ``a``: pupil aperture(s): scalar radius, (F,) radii,
    (2,2) ``[[-sag,-mer],[+sag,+mer]]`` or (F,2,2).

    ``groups=G``: ``z`` and ``a`` are sequences of G such entries (one per
    wavelength / variant) and the result holds G*F frames, group-major --
    the concatenation of the G single calls, built in one go."""
    obj = system.object
    projection = getattr(obj, "projection", None) or \
        getattr(obj, "extra", {}).get("projection", "rectilinear")
    yo = np.atleast_2d(np.asarray(yo, dtype=float))
    if groups is not None:
        per = yo.shape[0]
        z = np.concatenate([np.broadcast_to(np.asarray(zg, dtype=float),
                                            (per,)) for zg in z])
        a = np.concatenate([_apertures(ag, per) for ag in a])
        yo = np.tile(yo, (groups, 1))
    nf = yo.shape[0]
    z = np.broadcast_to(np.asarray(z, dtype=float), (nf,))
    a = _apertures(a, nf)
    out = np.zeros(nf, dtype=FIELD_DTYPE)
    out["z"] = z
    axis = np.zeros((nf, 3))
    axis[:, 2] = z
    if not obj.finite:
        u = _direction(yo, obj.angle, projection)
        out["am"] = np.fabs(a).max((1, 2))
        out["base"] = axis - z[:, None]*u
    else:
        y = np.zeros((nf, 3))
        y[:, :2] = -yo*obj.radius
        y[:, 2] = _sag0(system[0], y)
        u = axis if _telecentric(obj) else axis - y
        out["finite"] = 1
        out["flip"] = z < 0
        out["am"] = np.fabs(np.arctan2(a, z[:, None, None])).max((1, 2))
        out["base"] = y
    # sagittal / meridional unit vectors about the axis (0, 0, z), all
    # fields at once (_frame)
    s = _cross(u, axis)
    s[np.all(s == 0, axis=1)] = (1., 0., 0.)
    m = _cross(u, s)
    out["u"] = u
    out["s"] = s/np.sqrt(np.square(s).sum(-1))[:, None]
    out["m"] = m/np.sqrt(np.square(m).sum(-1))[:, None]
    return out


def aim_seeds(system, yo, z0, a0, group=0):
    """``AIM_SEED_DTYPE`` array for ``rt_aim_pupil``: per field the parts of
    the launch frame that do not depend on the pupil distance (direction of
    a field at infinity in the object's projection; object point of a finite
    field), from which the device rebuilds :func:`field_frames` for every
    trial distance; the starting pupil ``z0, a0`` and the surface table
    (wavelength) ``group`` the fields are aimed at.

    ``z0, a0, group`` may be sequences of G values: the result then holds
    G*F seeds, group-major (every field once per table)."""
    obj = system.object
    projection = getattr(obj, "projection", None) or \
        getattr(obj, "extra", {}).get("projection", "rectilinear")
    yo = np.atleast_2d(np.asarray(yo, dtype=float))
    per = len(yo)
    out = np.zeros(per, dtype=AIM_SEED_DTYPE)
    out["yo"] = yo
    if not obj.finite:
        out["dir"] = _direction(yo, obj.angle, projection)
    else:
        y = np.zeros((per, 3))
        y[:, :2] = -yo*obj.radius
        y[:, 2] = _sag0(system[0], y)
        out["finite"] = 1
        out["telecentric"] = _telecentric(obj)
        out["point"] = y
    if np.ndim(group):
        out = np.tile(out, len(group))
        z0, a0, group = (np.repeat(np.asarray(v), per)
                         for v in (z0, a0, group))
    out["z0"], out["a0"], out["group"] = z0, a0, group
    return out
