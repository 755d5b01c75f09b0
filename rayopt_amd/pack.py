"""Pack a System into the ``rt_surface`` table the kernel consumes.

The packer is duck-typed: it reads only public attributes of the elements
(``curvature, conic, aspherics, alternate_intersection, radius, offset,
rotated, rot_normal`` and ``get_n_mu``), so it accepts this package's own
:class:`rayopt_amd.model.System` *and* an unmodified reference
``rayopt.System`` -- that is what makes the engine a drop-in for
``GeometricTrace.propagate`` (rayopt/geometric_trace.py:72-80).

Every per-element scalar is evaluated here with the reference's own Python
expression (cited per line) so that the float handed to the kernel is
bit-identical to the one numpy would have broadcast.
"""
import numpy as np

from ._lib import (SURFACE_DTYPE, RT_MAX_ASPH, RT_MAX_SURFACES, F_ROTATED,
                   F_CURVED, F_CONIC, F_ASPH, F_ALT, F_REFRACT, F_MIRROR)


_SCALARS = ("c", "k", "kw", "kc2", "radius2", "mu", "muf", "smu", "mu2m1",
            "n0", "nasph", "flags")
_IDENTITY = np.eye(3).reshape(9)


def _sign(x):
    """np.sign for a Python float (NaN stays NaN)."""
    return float(int(x > 0) - int(x < 0)) if x == x else x


def resolve_range(length, start=1, stop=None):
    """Absolute [start, stop) of ``system[start:stop]`` (system.py:460)."""
    idx = range(length)[start:stop]
    return idx.start, max(idx.start, idx.stop)


def pack_system(system, wavelength, n_init, start=1, stop=None):
    """Return ``(table, n)``.

    ``table`` is a ``SURFACE_DTYPE`` array with one entry per element of
    ``system``; ``n[j]`` is the refractive index after element ``j`` for
    ``j`` in ``[start, stop)`` (``n[start-1] = n_init``), i.e. what
    System.propagate yields as ``n`` (rayopt/system.py:459-464).  Entries
    outside ``[start-1, stop)`` only carry their geometry.
    """
    length = len(system)
    if length > RT_MAX_SURFACES:
        raise ValueError("system has %d elements, the engine limit is %d"
                         % (length, RT_MAX_SURFACES))
    start, stop = resolve_range(length, start, stop)
    n = np.full(length, np.nan)
    n0 = float(n_init)
    if start >= 1:
        n[start - 1] = n0
    # one Python pass collecting columns, then one assignment per field: the
    # table is rebuilt on every propagate(), so this runs once per merit
    # evaluation / aiming iteration and must stay in the tens of microseconds
    offsets, rots, aspheres = [], [], []
    cols = {name: [] for name in _SCALARS}
    for j, el in enumerate(system):
        flags = 0
        # --- frame: TransformMixin (elements.py:120-154) ---
        offsets.append(el.offset)
        if getattr(el, "rotated", False):
            flags |= F_ROTATED
            rots.append(np.asarray(el.rot_normal, dtype=float).reshape(9))
        else:
            rots.append(_IDENTITY)
        # --- shape: Spheroid (elements.py:411-501) ---
        c = getattr(el, "curvature", 0.)
        k = getattr(el, "conic", 0.)
        asph = getattr(el, "aspherics", None)
        if c:                                  # `if self.curvature:` :450
            flags |= F_CURVED
        if k:                                  # `if not k:` :484
            flags |= F_CONIC
        nasph = 0
        if asph is not None:                   # elements.py:478
            nasph = len(asph)
            if nasph > RT_MAX_ASPH:
                raise ValueError("element %d: %d aspheric terms, limit %d"
                                 % (j, nasph, RT_MAX_ASPH))
            flags |= F_ASPH
            aspheres.append((j, asph))
        if getattr(el, "alternate_intersection", False):
            flags |= F_ALT                     # elements.py:497
        # --- index / Snell: Interface.get_n_mu (elements.py:283-289) ---
        mu = 1.
        inside = start <= j < stop
        before = n0 if inside else 1.
        if inside:
            if hasattr(el, "get_n_mu"):
                nj, mu = el.get_n_mu(n0, wavelength)
            else:                              # plain Element.propagate :230
                nj, mu = n0, 1.
            n[j] = nj
            n0 = nj
        if mu and mu != 1:                     # elements.py:313, :356
            flags |= F_REFRACT
            if mu == -1:                       # elements.py:363
                flags |= F_MIRROR
        for name, value in zip(_SCALARS, (
                c, k,
                1 + k,                         # kw, elements.py:489
                (1 + k)*c**2,                  # kc2, elements.py:451,468
                el.radius**2,                  # elements.py:207
                mu,
                abs(mu),                       # muf, elements.py:358
                _sign(mu),                     # smu, elements.py:366
                mu**2 - 1,                     # mu2m1, elements.py:365
                before, nasph, flags)):
            cols[name].append(value)
    table = np.zeros(length, dtype=SURFACE_DTYPE)
    table["offset"] = np.asarray(offsets, dtype=float)
    table["rot"] = rots
    for name in _SCALARS:
        table[name] = cols[name]
    for j, asph in aspheres:
        a = np.asarray(asph, dtype=float)
        table["asph"][j, :len(a)] = a                              # :452-454
        table["dasph"][j, :len(a)] = 2*(np.arange(len(a)) + 1)*a   # :471-472
    return table, n
