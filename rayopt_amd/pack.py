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
import struct

import numpy as np

from ._lib import (SURFACE_DTYPE, RT_MAX_ASPH, RT_MAX_SURFACES, F_ROTATED,
                   F_CURVED, F_CONIC, F_ASPH, F_ALT, F_REFRACT, F_MIRROR)


_IDENTITY = (1., 0., 0., 0., 1., 0., 0., 0., 1.)
_NO_ASPH = (0.,)*(2*RT_MAX_ASPH)
# rt_surface is 22 + 2*RT_MAX_ASPH doubles (c k kw kc2 radius2 mu muf smu
# mu2m1 n0 offset[3] rot[9] asph[] dasph[]) followed by nasph, flags and rc
# (the library's: the device fills it in, the caller hands over 0)
_ROW_FMT = "%ddiId" % (22 + 2*RT_MAX_ASPH)
assert struct.calcsize("<" + _ROW_FMT) == SURFACE_DTYPE.itemsize
assert [SURFACE_DTYPE.fields[f][1] for f in ("offset", "rot", "asph", "nasph",
                                             "flags")] == \
    [80, 104, 176, 176 + 16*RT_MAX_ASPH, 180 + 16*RT_MAX_ASPH]
_STRUCTS = {}


def _row_struct(length):
    st = _STRUCTS.get(length)
    if st is None:
        st = _STRUCTS[length] = struct.Struct("<" + _ROW_FMT*length)
    return st


def _note(el):
    """What would have to change for the packed row of ``el`` to change, or
    False if that cannot be told (an element or material of another
    package)."""
    stamp = getattr(el, "_stamp", None)
    if stamp is None:
        return False
    mat = getattr(el, "material", None)
    if mat is None:
        mkey = None
    elif hasattr(mat, "_pack_key"):
        mkey = mat._pack_key()
    else:
        return False
    asph = getattr(el, "aspherics", None)
    # the arrays the packer reads are handed out mutable, as the reference
    # hands them out (``el.offset[1] += .3``, ``set_path((j, "offset", 1),
    # v)``; rayopt/system.py:461 re-reads e.offset on every call): their
    # VALUES are part of the note, an in-place edit is a miss
    off = el.offset
    rot = el.rot_normal if getattr(el, "rotated", False) else None
    return (stamp, mkey, None if asph is None else tuple(asph),
            off.tobytes() if hasattr(off, "tobytes") else tuple(off),
            None if rot is None else
            rot.tobytes() if hasattr(rot, "tobytes") else repr(rot))


def _notes(system):
    if not hasattr(system, "__dict__"):
        return None
    notes = tuple([_note(el) for el in system])
    return None if False in notes else notes


def _floats(v, n):
    """``n`` Python floats of a small vector / matrix attribute."""
    v = v.tolist() if hasattr(v, "tolist") else list(v)
    if n == 9 and len(v) == 3:
        v = v[0] + v[1] + v[2]
    if len(v) != n:
        raise ValueError("expected %d numbers, got %r" % (n, v))
    return v


def _sign(x):
    """np.sign for a Python float (NaN stays NaN)."""
    return (1. if x > 0 else -1. if x < 0 else 0.) if x == x else x


def resolve_range(length, start=1, stop=None):
    """Absolute [start, stop) of ``system[start:stop]`` (system.py:460)."""
    idx = range(length)[start:stop]
    return idx.start, max(idx.start, idx.stop)


def pack_system(system, wavelength, n_init, start=1, stop=None, _notes_of=None):
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
    # one Python pass appending plain floats to ONE flat list, then a single
    # struct.pack into the table's memory: the table is rebuilt on every
    # propagate() (elements are mutable between calls,
    # rayopt/geometric_trace.py:98-99), so this runs once per merit
    # evaluation / aiming iteration and decides the wall time of small
    # traces (list-of-lists -> ndarray conversions and per-field assignments
    # cost several times more than the attribute reads themselves)
    # --- nothing touched since the last call with these arguments: the same
    # table (every element and material of this package counts its attribute
    # assignments, model.Stamped; values that can change in place -- aspheric
    # and dispersion coefficients -- are part of the note)
    notes = _notes(system) if _notes_of is None else _notes_of[0]
    whole = None
    if notes is not None:
        # (a few tables are kept side by side: a polychromatic trace packs
        # one per wavelength on every propagate(), and a cache of one would
        # be evicted by the next wavelength each time -- 195 -> 30 us for
        # BASELINE config C2's three wavelengths)
        whole = (wavelength, float(n_init), start, stop, notes)
        key = (wavelength, float(n_init), start, stop)
        tables = system.__dict__.get("_pack_table")
        kept = tables.get(key) if tables is not None else None
        if kept is not None and kept[0] == whole:
            return (np.frombuffer(bytearray(kept[1]), dtype=SURFACE_DTYPE),
                    kept[2].copy())
    flat = []
    extend = flat.extend
    for j, el in enumerate(system):
        inside = start <= j < stop
        # --- an element this package made that has not been touched since
        # its row was last built: the same row (model.Stamped)
        slot = None
        note = notes[j] if notes is not None else _note(el)
        if note is not False:
            slot = (wavelength, n0 if inside else None)
            rows = el.__dict__.get("_pack_rows")
            hit = rows.get(slot) if rows is not None else None
            if hit is not None and hit[0] == note:
                extend(hit[1])
                if inside:
                    n[j] = n0 = hit[2]
                continue
        first = len(flat)
        flags = 0
        # --- shape: Spheroid (elements.py:411-501) ---
        c = getattr(el, "curvature", 0.)
        k = getattr(el, "conic", 0.)
        asph = getattr(el, "aspherics", None)
        if c:                                  # `if self.curvature:` :450
            flags |= F_CURVED
        if k:                                  # `if not k:` :484
            flags |= F_CONIC
        if getattr(el, "alternate_intersection", False):
            flags |= F_ALT                     # elements.py:497
        # --- index / Snell: Interface.get_n_mu (elements.py:283-289) ---
        mu = 1.
        before = 1.
        if inside:
            before = n0
            if hasattr(el, "get_n_mu"):
                n0, mu = el.get_n_mu(n0, wavelength)
            # else: plain Element.propagate :230 leaves n and the ray alone
            n[j] = n0
        if mu and mu != 1:                     # elements.py:313, :356
            flags |= F_REFRACT
            if mu == -1:                       # elements.py:363
                flags |= F_MIRROR
        extend((c, k,
                1 + k,                         # kw, elements.py:489
                (1 + k)*c**2,                  # kc2, elements.py:451,468
                el.radius**2,                  # elements.py:207
                mu,
                abs(mu),                       # muf, elements.py:358
                _sign(mu),                     # smu, elements.py:366
                mu**2 - 1,                     # mu2m1, elements.py:365
                before))
        # --- frame: TransformMixin (elements.py:120-154) ---
        extend(_floats(el.offset, 3))
        if getattr(el, "rotated", False):
            flags |= F_ROTATED
            extend(_floats(el.rot_normal, 9))
        else:
            extend(_IDENTITY)
        nasph = 0
        if asph is None:                       # elements.py:478
            extend(_NO_ASPH)
        else:
            nasph = len(asph)
            if nasph > RT_MAX_ASPH:
                raise ValueError("element %d: %d aspheric terms, limit %d"
                                 % (j, nasph, RT_MAX_ASPH))
            flags |= F_ASPH
            a = [float(x) for x in asph]
            pad = (0.,)*(RT_MAX_ASPH - nasph)
            extend(a)                                              # :452-454
            extend(pad)
            extend([2*(q + 1)*x for q, x in enumerate(a)])          # :471-472
            extend(pad)
        extend((nasph, flags, 0.))
        if slot is not None:
            rows = el.__dict__.get("_pack_rows")
            if rows is None or len(rows) > 16:
                rows = el.__dict__["_pack_rows"] = {}
            rows[slot] = (note, tuple(flat[first:]), n0)
    packed = _row_struct(length).pack(*flat)
    table = np.frombuffer(bytearray(packed), dtype=SURFACE_DTYPE)
    if whole is not None:
        tables = system.__dict__.get("_pack_table")
        if tables is None or len(tables) > 8:
            tables = system.__dict__["_pack_table"] = {}
        tables[(wavelength, float(n_init), start, stop)] = (
            whole, packed, n.copy())
    return table, n


def pack_tables(system, wavelengths, n_inits, start=1, stop=None):
    """``(tables (G,L), n (G,L))``: one table per wavelength of a
    polychromatic batch (ray groups).  What every table's cache entry is
    checked against -- the notes of all elements -- is gathered ONCE for the
    G tables (50 us of the 65 us a cached pack_system call takes)."""
    shared = [_notes(system)]
    packed = [pack_system(system, l, n0, start, stop, _notes_of=shared)
              for l, n0 in zip(wavelengths, n_inits)]
    tables = np.frombuffer(bytearray(b"".join(
        t.tobytes() for t, _ in packed)), dtype=SURFACE_DTYPE).reshape(
            len(packed), -1)
    return tables, np.stack([n for _, n in packed])
