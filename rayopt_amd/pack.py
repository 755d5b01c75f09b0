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
    table = np.zeros(length, dtype=SURFACE_DTYPE)
    n = np.full(length, np.nan)
    n0 = float(n_init)
    if start >= 1:
        n[start - 1] = n0
    for j, el in enumerate(system):
        row = table[j]
        flags = 0
        # --- frame: TransformMixin (elements.py:120-154) ---
        row["offset"] = np.asarray(el.offset, dtype=float)
        if getattr(el, "rotated", False):
            flags |= F_ROTATED
            row["rot"] = np.asarray(el.rot_normal, dtype=float).reshape(9)
        else:
            row["rot"] = np.eye(3).reshape(9)
        # --- shape: Spheroid (elements.py:411-501) ---
        c = getattr(el, "curvature", 0.)
        k = getattr(el, "conic", 0.)
        asph = getattr(el, "aspherics", None)
        row["c"] = c
        row["k"] = k
        row["kw"] = 1 + k                      # elements.py:489
        row["kc2"] = (1 + k)*c**2              # elements.py:451,468
        if c:                                  # `if self.curvature:` :450
            flags |= F_CURVED
        if k:                                  # `if not k:` :484
            flags |= F_CONIC
        if asph is not None:                   # elements.py:478
            if len(asph) > RT_MAX_ASPH:
                raise ValueError("element %d: %d aspheric terms, limit %d"
                                 % (j, len(asph), RT_MAX_ASPH))
            flags |= F_ASPH
            row["nasph"] = len(asph)
            for i, ai in enumerate(asph):
                row["asph"][i] = ai            # elements.py:452-454
                row["dasph"][i] = 2*(i + 1)*ai  # elements.py:471-472
        if getattr(el, "alternate_intersection", False):
            flags |= F_ALT                     # elements.py:497
        row["radius2"] = el.radius**2          # elements.py:207
        # --- index / Snell: Interface.get_n_mu (elements.py:283-289) ---
        mu = 1.
        row["n0"] = n0 if start <= j < stop else 1.
        if start <= j < stop:
            if hasattr(el, "get_n_mu"):
                nj, mu = el.get_n_mu(n0, wavelength)
            else:                              # plain Element.propagate :230
                nj, mu = n0, 1.
            n[j] = nj
            n0 = nj
        row["mu"] = mu
        row["muf"] = abs(mu)                   # elements.py:358
        row["smu"] = np.sign(mu)               # elements.py:366
        row["mu2m1"] = mu**2 - 1               # elements.py:365
        if mu and mu != 1:                     # elements.py:313, :356
            flags |= F_REFRACT
            if mu == -1:                       # elements.py:363
                flags |= F_MIRROR
        row["flags"] = flags
    return table, n
