"""Host-side optical model: the objects GeometricTrace walks.

This is the host mirror of the reference's public interface for the hot path
(SURVEY.md section 8b): a ``System`` is a ``list`` of elements whose public
attributes carry the prescription; names, argument meaning and conventions
follow rayopt so prescriptions (YAML dicts) and user code carry over:

  =====================  ==================================================
  here                   reference
  =====================  ==================================================
  ``Pose``               ``TransformMixin``      rayopt/elements.py:29-175
  ``Element``            ``Element``             rayopt/elements.py:178-272
  ``Interface``          ``Interface``           rayopt/elements.py:275-330
  ``Spheroid``           ``Spheroid``            rayopt/elements.py:411-438
  ``Material.make``      ``Material.make``       rayopt/material.py:84-115
  ``System``             ``System``              rayopt/system.py:34-68,
                                                 193-199, 413-442, 459-464
  =====================  ==================================================

Only state and O(L) scalar geometry live here.  The per-ray arithmetic
(intercept, clip, refract, frame changes) is *not* implemented on the host:
it runs in the HIP kernel (csrc/rt_math.h) and nowhere else in this package.
"""
import itertools
import math

import numpy as np

from .design import DesignMixin, pupil_set_radius

# Fraunhofer lines used as default wavelengths (d, C, F), metres.
LAMBDA_D, LAMBDA_C, LAMBDA_F = 587.56e-9, 656.27e-9, 486.13e-9


# --------------------------------------------------------------------------
# materials: scalar n(lambda), evaluated on the host once per (element, l)
# --------------------------------------------------------------------------

_STAMPS = itertools.count(1)


class Stamped:
    """Every attribute assignment gives the object a new ``_stamp``, drawn
    from ONE process-wide counter -- so a stamp is never seen twice, not on
    the same object after a change and not on another object (a new element
    that happens to reuse the address and construction history of a deleted
    one included).  The surface table is re-packed on every ``propagate()``
    because elements are mutable between calls
    (rayopt/geometric_trace.py:98-99); the packer keeps an element's row as
    long as its stamp, its material's stamp and the values it cannot see
    being changed in place (aspheric coefficients, dispersion coefficients)
    are the same (rayopt_amd/pack.py).  A copy (``copy.deepcopy``) starts
    with the stamp of its original, whose content it shares at that moment.
    Names starting with ``_pack`` are the packer's own notes and do not
    count."""
    _stamp = 0

    def __setattr__(self, name, value):
        object.__setattr__(self, name, value)
        if name[:5] != "_pack":
            object.__setattr__(self, "_stamp", next(_STAMPS))


class Material(Stamped):
    """Refractive medium; ``mirror`` marks a reflecting coating."""
    solid = True
    mirror = False

    def _pack_key(self):
        """What identifies n(lambda) of this medium to the packer's cache:
        its stamp (attribute assignments) and the VALUES of every array it
        holds (``GasFormula.b/c``, dispersion coefficients: they can be
        edited in place, which no stamp sees)."""
        return (self._stamp,) + tuple(
            v.tobytes() for v in self.__dict__.values()
            if type(v) is np.ndarray)

    catalog = None

    def __init__(self, name="-", solid=True, mirror=False, catalog=None):
        self.name = name
        self.solid = solid
        self.mirror = mirror
        if catalog is not None:
            self.catalog = catalog

    def refractive_index(self, wavelength):
        return 1.

    def delta_n(self, short, long):
        return self.refractive_index(short) - self.refractive_index(long)

    def dispersion(self, short, mid, long):
        """Abbe number over (short, mid, long); inf without dispersion."""
        dn = self.delta_n(short, long)
        return (self.refractive_index(mid) - 1)/dn if dn else math.inf

    @property
    def nd(self):
        return self.refractive_index(LAMBDA_D)

    @property
    def vd(self):
        return self.dispersion(LAMBDA_F, LAMBDA_D, LAMBDA_C)

    def __str__(self):
        """``catalog/name`` (rayopt/material.py:117-121): what
        :meth:`make` resolves again."""
        if self.catalog is not None:
            return "%s/%s" % (self.catalog, self.name)
        return self.name

    def spec(self):
        """What a prescription file holds for this medium: something
        :meth:`make` turns into an equal material again."""
        return str(self)

    @staticmethod
    def make(spec):
        """Same dispatch as rayopt's Material.make for the forms that need no
        catalogue database: ``None``, a Material, a float (constant index), an
        ``(nd, vd)`` tuple or ``"nd/vd"`` string (Abbe model), and the basic
        names ``vacuum``, ``air``, ``mirror`` (optionally ``basic/<name>``);
        other names (``N-BK7``, ``schott/N-BK7``) resolve in the catalogue
        files loaded with :func:`rayopt_amd.catalog.load_agf` and then, as
        ``[source/][catalog/]name``, in the glass library
        (:mod:`rayopt_amd.library`: the user's rayopt ``library.sqlite`` if
        there is one, else the built-in table of refractiveindex.info
        formulas -- ``SCHOTT-SK|N-SK16``).
        """
        if spec is None or isinstance(spec, Material):
            return spec
        if hasattr(spec, "refractive_index"):   # foreign material object
            return spec
        if isinstance(spec, dict):               # {typ:, coefficients:, ...}
            spec = dict(spec)
            spec.pop("type", None)
            return DispersionGlass(spec.pop("typ", "sellmeier"),
                                   spec.pop("coefficients"), **{
                k: spec[k] for k in ("name", "solid", "mirror") if k in spec})
        if type(spec) is float:
            return ConstantIndex(spec)
        if type(spec) is tuple:
            return AbbeGlass(spec[0], spec[1])
        if isinstance(spec, (int, np.integer, np.floating)):
            return ConstantIndex(float(spec))
        text = str(spec)
        parts = text.split("/")
        if len(parts) <= 2:                      # "n" or "n/v"
            try:
                numbers = [float(part) for part in parts]
            except ValueError:
                pass
            else:
                return (AbbeGlass(*numbers) if len(numbers) == 2
                        else ConstantIndex(numbers[0]))
        key = parts[-1].lower()
        if (len(parts) == 1 or parts[-2].lower() == "basic") and key in BASIC:
            return BASIC[key]
        from .catalog import catalogs            # glasses by name
        glass = catalogs.find(text)
        if glass is not None:
            return glass
        # [source/][catalog/]name in the glass library, as the reference
        # resolves it (rayopt/material.py:104-115)
        from .library import Library
        name = parts[-1]
        catalog = parts[-2] if len(parts) > 1 else None
        source = parts[-3] if len(parts) > 2 else None
        try:
            return Library.one().get("material", name, catalog, source)
        except KeyError:
            raise KeyError(
                "material %r: not in a loaded glass catalogue "
                "(rayopt_amd.catalog.load_agf) nor in the glass library "
                "(rayopt_amd.library); or give a numeric index or an "
                "'nd/vd' pair" % (spec,)) from None


class ConstantIndex(Material):
    def __init__(self, n=1., **kw):
        super().__init__(**kw)
        self.n = n

    def refractive_index(self, wavelength):
        return self.n

    def __str__(self):
        if self.name and self.name != "-":
            return super().__str__()
        return repr(self.n)

    def spec(self):
        if self.name and self.name != "-":
            return super().spec()
        return float(self.n)


class AbbeGlass(Material):
    """Linear dispersion model from (n_ref, Abbe number)."""
    def __init__(self, n=1., v=math.inf, lambda_ref=LAMBDA_D,
                 lambda_long=LAMBDA_C, lambda_short=LAMBDA_F, **kw):
        super().__init__(**kw)
        self.n, self.v = n, v
        self.lambda_ref = lambda_ref
        self.lambda_long, self.lambda_short = lambda_long, lambda_short

    def refractive_index(self, wavelength):
        return (self.n + (wavelength - self.lambda_ref) /
                (self.lambda_long - self.lambda_short)*(1 - self.n)/self.v)

    def __str__(self):
        if self.name and self.name != "-":
            return super().__str__()
        return "%r/%r" % (self.n, self.v)


class GasFormula(Material):
    """n = 1 + sum(B_i / (C_i - w^-2)), w in micrometres (standard air)."""
    def __init__(self, b, c, **kw):
        super().__init__(**kw)
        self.b = np.asarray(b, dtype=float)
        self.c = np.asarray(c, dtype=float)

    def refractive_index(self, wavelength):
        w = wavelength/1e-6
        return 1. + (self.b/(self.c - w**-2)).sum()


def _pairs(c):
    return np.asarray(c, dtype=float).reshape(-1, 2).T


def _laurent(head, tail, w):
    """head + sum_i tail[i] w^(-2(i+1))"""
    n = head
    for i, ci in enumerate(tail):
        n = n + ci*w**(-2*(i + 1))
    return n


# Published dispersion formulas, w = wavelength in micrometres, c = the
# coefficient vector in the order the glass catalogues (Zemax .agf, OSLO
# .glc, refractiveindex.info) list them; names as in rayopt
# (rayopt/material.py:240-322).
DISPERSION = {
    "schott": lambda w, c: np.sqrt(_laurent(c[0] + c[1]*w**2, c[2:], w)),
    "sellmeier": lambda w, c: np.sqrt(
        1. + (_pairs(c)[0]*w**2/(w**2 - _pairs(c)[1]**2)).sum()),
    "sellmeier_squared": lambda w, c: np.sqrt(
        1. + (_pairs(c)[0]*w**2/(w**2 - _pairs(c)[1])).sum()),
    "sellmeier_squared_transposed": lambda w, c: np.sqrt(
        1. + (c.reshape(2, -1)[0]*w**2/(w**2 - c.reshape(2, -1)[1])).sum()),
    # Zemax "Sellmeier 2" (.agf formula 4; no counterpart in the reference):
    # n^2 - 1 = A + B1 w^2/(w^2 - l1^2) + B2/(w^2 - l2^2)
    "sellmeier2": lambda w, c: np.sqrt(
        1. + c[0] + c[1]*w**2/(w**2 - c[2]**2) + c[3]/(w**2 - c[4]**2)),
    "conrady": lambda w, c: c[0] + c[1]/w + c[2]/w**3.5,
    "herzberger": lambda w, c: (
        c[0] + c[1]/(w**2 - .028) + c[2]/(w**2 - .028)**2 + c[3]*w**2 +
        c[4]*w**4 + c[5]*w**6),
    "sellmeier_offset": lambda w, c: np.sqrt(
        1. + c[0] + (_pairs(c[1:1 + (len(c) - 1)//2*2])[0]*w**2 /
                     (w**2 - _pairs(c[1:1 + (len(c) - 1)//2*2])[1]**2)).sum()),
    "sellmeier_squared_offset": lambda w, c: np.sqrt(
        1. + c[0] + (_pairs(c[1:1 + (len(c) - 1)//2*2])[0]*w**2 /
                     (w**2 - _pairs(c[1:1 + (len(c) - 1)//2*2])[1])).sum()),
    "handbook_of_optics1": lambda w, c: np.sqrt(
        c[0] + c[1]/(w**2 - c[2]) - c[3]*w**2),
    "handbook_of_optics2": lambda w, c: np.sqrt(
        c[0] + c[1]*w**2/(w**2 - c[2]) - c[3]*w**2),
    "extended2": lambda w, c: np.sqrt(_laurent(
        c[0] + c[1]*w**2 + c[6]*w**4 + c[7]*w**6, c[2:6], w)),
    "hikari": lambda w, c: np.sqrt(_laurent(
        c[0] + c[1]*w**2 + c[2]*w**4, c[3:], w)),
    "gas": lambda w, c: 1. + (c.reshape(2, -1)[0] /
                              (c.reshape(2, -1)[1] - w**-2)).sum(),
    "gas_offset": lambda w, c: c[0] + 1. + (
        c[1:].reshape(2, -1)[0]/(c[1:].reshape(2, -1)[1] - w**-2)).sum(),
    "refractiveindex_info": lambda w, c: np.sqrt(
        c[0] + c[1]*w**c[2]/(w**2 - c[3]**c[4]) +
        c[5]*w**c[6]/(w**2 - c[7]**c[8]) +
        (_pairs(c[9:])[0]*w**_pairs(c[9:])[1]).sum()),
    "retro": lambda w, c: np.sqrt(
        2 + 1/(c[0] + c[1]*w**2/(w**2 - c[2]) + c[3]*w**2 - 1)),
    "cauchy": lambda w, c: c[0] + (_pairs(c[1:])[0]*w**_pairs(c[1:])[1]).sum(),
    "polynomial": lambda w, c: np.sqrt(
        c[0] + (_pairs(c[1:])[0]*w**_pairs(c[1:])[1]).sum()),
    "exotic": lambda w, c: np.sqrt(
        c[0] + c[1]/(w**2 - c[2]) +
        c[3]*(w - c[4])/((w - c[4])**2 + c[5])),
}


class DispersionGlass(Material):
    """Glass given by a catalogue dispersion formula and its coefficients,
    e.g. ``DispersionGlass("sellmeier_squared", [B1, C1, B2, C2, B3, C3])``."""
    def __init__(self, typ, coefficients, **kw):
        super().__init__(**kw)
        if typ not in DISPERSION:
            raise KeyError("unknown dispersion formula %r" % typ)
        self.typ = typ
        self.coefficients = np.atleast_1d(np.asarray(coefficients, float))

    def _pack_key(self):
        c = self.coefficients
        if type(c) is np.ndarray:
            return super()._pack_key()
        return (self._stamp, tuple(c))

    def refractive_index(self, wavelength):
        n = DISPERSION[self.typ](wavelength/1e-6, self.coefficients)
        return -n if self.mirror else n

    def __str__(self):
        if self.name and self.name != "-":      # a catalogue glass
            return super().__str__()
        return "%s%r" % (self.typ, [float(c) for c in self.coefficients])

    def spec(self):
        if self.name and self.name != "-":
            return super().spec()
        dat = {"typ": self.typ,
               "coefficients": [float(c) for c in self.coefficients]}
        if not self.solid:
            dat["solid"] = False
        if self.mirror:
            dat["mirror"] = True
        return dat


BASIC = {
    "vacuum": ConstantIndex(1., name="vacuum", solid=False, catalog="basic"),
    "mirror": Material(name="mirror", solid=False, mirror=True,
                       catalog="basic"),
    "air": GasFormula([.05792105, .00167917], [238.0185, 57.362],
                      name="air", solid=False, catalog="basic"),
}


# --------------------------------------------------------------------------
# element pose
# --------------------------------------------------------------------------

def _axis_angle(angle, axis):
    """Rotation about ``axis`` by ``angle`` (Rodrigues), 3x3:
    ``R[i][j] = (delta_ij cos + d_i d_j (1 - cos)) + eps_ikj d_k sin``.
    Each entry is summed in that order -- diagonal term, then the projector
    term, then the cross-product term -- which is the order the reference's
    ``rotation_matrix`` accumulates them in (used at rayopt/elements.py:147),
    so the entries carry the same rounding."""
    d = np.array(axis, dtype=float)
    d = (d/math.sqrt(np.dot(d, d))).tolist()
    ca, sa = math.cos(angle), math.sin(angle)
    # (k, sign) of the cross-product matrix [d]x entry (i, j)
    cross = {(0, 1): (2, -1.), (0, 2): (1, 1.), (1, 0): (2, 1.),
             (1, 2): (0, -1.), (2, 0): (1, -1.), (2, 1): (0, 1.)}
    rot = np.empty((3, 3))
    for i in range(3):
        for j in range(3):
            entry = (ca if i == j else 0.) + (d[i]*d[j])*(1. - ca)
            if i != j:
                k, sign = cross[i, j]
                entry = entry + sign*(d[k]*sa)
            rot[i, j] = entry
    return rot


def _euler_rxyz(ax, ay, az):
    """Rotating-frame x-y-z Euler matrix Rx(ax) Ry(ay) Rz(az) in closed
    form.  The nine entries are the products the reference's
    ``euler_matrix(ax, ay, az, "rxyz")`` forms (rayopt/transformations.py:
    1047; "rxyz" = first axis z, odd parity, rotating frame: the angles enter
    negated and in reverse order), term for term in Python floats, so that
    ``rot_normal`` -- and with it every tilted trace -- equals the
    reference's to the last bit rather than to 1e-17."""
    s3, s2, s1 = math.sin(-az), math.sin(-ay), math.sin(-ax)
    c3, c2, c1 = math.cos(-az), math.cos(-ay), math.cos(-ax)
    c3c1, c3s1 = c3*c1, c3*s1
    s3c1, s3s1 = s3*c1, s3*s1
    return np.array([
        [c2*c3, c2*s3, -s2],
        [s2*c3s1 - s3c1, s2*s3s1 + c3c1, c2*s1],
        [s2*c3c1 + s3s1, s2*s3c1 - c3s1, c2*c1]])


def _euler_angles_rxyz(rot):
    """Inverse of :func:`_euler_rxyz`: the angles ``(ax, ay, az)`` of a
    rotation matrix.  With s_k = sin(-a_k), c_k = cos(-a_k) the matrix reads
    ``[[c2 c3, c2 s3, -s2], [., ., c2 s1], [., ., c2 c1]]``, so ax and az
    follow from the ratios in the last column / first row and ay from -s2
    against c2 = hypot(R12, R22); at the pole (c2 = 0) only ax + az is
    defined and az is set to 0 (the convention of the reference's
    ``euler_from_matrix(rot, "rxyz")``, rayopt/elements.py:117)."""
    c2 = math.sqrt(rot[2][2]*rot[2][2] + rot[1][2]*rot[1][2])
    if c2 > 4.*np.finfo(float).eps:
        a_x = math.atan2(rot[1][2], rot[2][2])
        a_z = math.atan2(rot[0][1], rot[0][0])
    else:                       # gimbal lock: all of ax + az goes to az
        a_x = 0.
        a_z = math.atan2(-rot[1][0], rot[1][1])
    a_y = math.atan2(-rot[0][2], c2)
    return -a_x, -a_y, -a_z


class Pose(Stamped):
    """Placement of an element relative to the previous one.

    ``offset = distance*direction`` is expressed in the global, unrotated
    frame and accumulates along the system; ``angles`` (rotating x-y-z Euler)
    tilt the element normal relative to ``direction``.  A negative distance
    flips the direction (used after mirrors).  ``rot_normal`` maps the
    element-normal frame to the global one for row vectors:
    ``from_normal(v) = v @ rot_normal``, ``to_normal(v) = v @ rot_normal.T``.
    """
    def __init__(self, distance=0., direction=(0, 0, 1.), angles=(0, 0, 0),
                 offset=None):
        self.update(distance, direction, angles)
        if offset is not None:
            self.offset = offset

    def update(self, distance, direction, angles):
        u = np.array(direction, dtype=float)
        norm = np.linalg.norm(u)
        u = u/norm if norm else np.array([0., 0., 1.])
        if distance < 0:
            distance, u = -distance, -u
        self._distance, self._direction = distance, u
        self._offset = distance*u
        self._angles = np.array(angles, dtype=float)
        # np.allclose(u, (0, 0, 1)) and np.allclose(angles, 0) of the
        # reference (rayopt/elements.py:135-136), on Python floats: this runs
        # whenever a distance changes (refocus, thickness variables)
        ux, uy, uz = u.tolist()
        self.straight = (abs(ux) <= 1e-8 and abs(uy) <= 1e-8 and
                         abs(uz - 1.) <= 1e-8 + 1e-5)
        self.normal = all(abs(a) <= 1e-8 for a in self._angles.tolist())
        self.rotated = not (self.straight and self.normal)
        self.rot_axis = self.rot_normal = None
        if not self.rotated:
            return
        rot = np.eye(3)
        if not self.straight:
            axis = np.cross(u, (0, 0, 1.))
            angle = np.arcsin(min(1., np.linalg.norm(axis)))
            if u[2] < 0:
                angle = np.pi - angle
            if np.allclose(axis, 0):
                axis = (1., 0, 0)
            self.rot_axis = _axis_angle(angle, axis)
            rot = np.dot(rot, self.rot_axis)
        if not self.normal:
            rot = np.dot(rot, _euler_rxyz(*self._angles))
        self.rot_normal = rot

    def align(self, direction, mu):
        """Tilt the element so that a ray arriving along this element's
        ``direction`` leaves along ``direction`` after refraction with index
        ratio ``mu`` (or reflection): the surface normal is put along
        ``mu*incident - excident`` (Snell in vector form;
        rayopt/elements.py:103-118, called by ``System.align``)."""
        incident = self._direction
        normal = mu*incident - np.asarray(direction, dtype=float)
        if mu < 1:
            normal = -normal
        if np.allclose(normal, 0):
            normal = np.array([0., 0., 1.])
        normal = normal/np.linalg.norm(normal)
        axis = np.cross(incident, normal)
        tilt = np.arcsin(np.linalg.norm(axis))
        if np.allclose(axis, 0):
            axis = (1., 0., 0.)
        self.update(self._distance, self._direction,
                    _euler_angles_rxyz(_axis_angle(tilt, axis).T))

    distance = property(lambda self: self._distance,
                        lambda self, d: self.update(d, self._direction,
                                                    self._angles))
    direction = property(lambda self: self._direction,
                         lambda self, d: self.update(self._distance, d,
                                                     self._angles))
    angles = property(lambda self: self._angles,
                      lambda self, a: self.update(self._distance,
                                                  self._direction, a))

    @property
    def offset(self):
        return self._offset

    @offset.setter
    def offset(self, offset):
        offset = np.asarray(offset, dtype=float)
        d = float(np.linalg.norm(offset))
        self.update(d, offset/d if d else (0, 0, 1.), self._angles)

    # O(1) frame changes of a few host vectors (origins, axis directions);
    # ray batches are rotated in the kernel, not here.
    @staticmethod
    def _apply(rot, active, vectors):
        if active:
            vectors = tuple(np.dot(v, rot) for v in vectors)
        return vectors[0] if len(vectors) == 1 else vectors

    def from_axis(self, *v):
        return self._apply(self.rot_axis, not self.straight, v)

    def to_axis(self, *v):
        return self._apply(None if self.straight else self.rot_axis.T,
                           not self.straight, v)

    def from_normal(self, *v):
        return self._apply(self.rot_normal, self.rotated, v)

    def to_normal(self, *v):
        return self._apply(None if not self.rotated else self.rot_normal.T,
                           self.rotated, v)

    @property
    def incidence(self):
        return self.to_normal(self._direction)


class Element(Pose):
    """A reference plane with a circular clear aperture of ``radius``."""
    typeletter = "E"

    def __init__(self, radius=math.inf, diameter=None, **kw):
        super().__init__(**kw)
        self.radius = diameter/2 if diameter is not None else radius

    def dict(self):
        dat = {}
        if self.distance:
            dat["distance"] = float(self.distance)
        if not self.straight:
            dat["direction"] = [float(x) for x in self.direction]
        if not self.normal:
            dat["angles"] = [float(x) for x in self.angles]
        if np.isfinite(self.radius):
            dat["radius"] = float(self.radius)
        return dat

    def rescale(self, scale):
        self.distance *= scale
        self.radius *= scale

    def reverse(self):
        """A plane looks the same from behind."""

    # per-ray arithmetic of a single element: runs on the GPU like the
    # system trace (same kernel, a two-row table)
    def propagate(self, y0, u0, n0, l, clip=True):
        """(y, u, n, t*n0) for rays given in this element's normal frame
        relative to its vertex (rayopt/elements.py:230-236, 306-315)."""
        from .engine import element_propagate
        return element_propagate(self, y0, u0, n0, l, clip)

    def intercept(self, y, u):
        """Ray length to the surface (rayopt/elements.py:195-201, 333-349,
        477-501)."""
        from .engine import element_propagate
        return element_propagate(self, y, u, 1., None, False)[3]


class Interface(Element):
    """An element separating two media."""
    typeletter = "I"

    def __init__(self, material=None, **kw):
        super().__init__(**kw)
        self.material = Material.make(material) if material else None

    def refractive_index(self, wavelength):
        return self.material.refractive_index(wavelength)

    def get_n_mu(self, n0, l):
        """(index behind the element, mu = n0/n); mu = -1 marks a mirror."""
        if self.material is None:
            return n0, 1.
        if self.material.mirror:
            return n0, -1.
        n = self.refractive_index(l)
        return n, n0/n

    def dict(self):
        dat = super().dict()
        if self.material is not None:
            spec = getattr(self.material, "spec", None)
            dat["material"] = spec() if spec else str(self.material)
        return dat

    def edge_sag(self, axis=1):
        """Surface residual ``z - sag`` at the rim for ``z = 0``
        (rayopt/elements.py:328-331)."""
        from .design import sag
        return 0. - sag(self, self.radius)


class Spheroid(Interface):
    """Rotationally symmetric conic + even-asphere surface.

    sag(r) = c r^2 / (1 + sqrt(1 - (1+k) c^2 r^2)) + sum_i a_i r^(2(i+1)),
    ``aspherics[0]`` multiplies r^2 (Zemax EVENASPH PARM 1).
    """
    typeletter = "S"

    def __init__(self, curvature=0., conic=0., aspherics=None, roc=None,
                 alternate_intersection=False, **kw):
        super().__init__(**kw)
        if roc is not None:
            curvature = 1./roc
        self.curvature = curvature
        self.conic = conic
        self.aspherics = list(aspherics) if aspherics is not None else None
        self.alternate_intersection = alternate_intersection
        if curvature and np.isfinite(self.radius) and conic > -1:
            if self.radius**2 > 1/((1 + conic)*curvature**2):
                raise ValueError("aperture radius %g exceeds the conic's "
                                 "extent" % self.radius)

    def dict(self):
        dat = super().dict()
        if self.curvature:
            dat["curvature"] = float(self.curvature)
        if self.conic:
            dat["conic"] = float(self.conic)
        if self.aspherics is not None:
            dat["aspherics"] = [float(a) for a in self.aspherics]
        if self.alternate_intersection:
            dat["alternate_intersection"] = True
        return dat

    def rescale(self, scale):
        super().rescale(scale)
        self.curvature /= scale
        if self.aspherics is not None:
            self.aspherics = [a/scale**(2*i + 1)
                              for i, a in enumerate(self.aspherics)]

    def reverse(self):
        super().reverse()
        self.curvature *= -1
        if self.aspherics is not None:
            self.aspherics = [-a for a in self.aspherics]


ELEMENT_TYPES = {"spheroid": Spheroid, "interface": Interface,
                 "element": Element}


def make_element(spec):
    """Element from a prescription dict (default type: spheroid)."""
    if isinstance(spec, Pose):
        return spec
    spec = dict(spec)
    cls = ELEMENT_TYPES[spec.pop("type", "spheroid")]
    return cls(**spec)


# --------------------------------------------------------------------------
# system
# --------------------------------------------------------------------------

class Conjugate:
    """Object/image specification, kept as data for ray generation."""
    def __init__(self, spec, finite_default):
        spec = dict(spec or {})
        typ = spec.pop("type", None)
        if typ is None:
            typ = "finite" if finite_default else "infinite"
        self.type = typ
        self.finite = typ == "finite"
        self.pupil = dict(spec.pop("pupil", {}) or {})
        if "angle_deg" in spec:
            spec["angle"] = math.radians(spec.pop("angle_deg"))
        self.angle = spec.pop("angle", 0.)
        self.radius = spec.pop("radius", 0.)
        self.extra = spec

    @property
    def point(self):
        """No field extent (rayopt/conjugates.py:101-103,180-182)."""
        return not (self.radius if self.finite else self.angle)

    def text(self):
        from .design import conjugate_text
        return conjugate_text(self)

    def rescale(self, scale):
        from .design import pupil_rescale
        pupil_rescale(self.pupil, scale)
        if self.finite:
            self.radius *= scale

    def dict(self):
        dat = {"type": self.type, "pupil": dict(self.pupil)}
        if self.finite:
            dat["radius"] = self.radius
        else:
            dat["angle"] = self.angle
        dat.update(self.extra)
        return dat


class System(DesignMixin, list):
    """Sequential optical system: a list of elements, object first, image
    last.  Mutating elements between traces is allowed; the surface table is
    re-packed on every propagate()."""

    def __init__(self, elements=None, description="", scale=1e-3,
                 wavelengths=None, stop=1, fields=None, object=None,
                 image=None, pickups=None, validators=None, solves=None):
        super().__init__(make_element(e) for e in (elements or []))
        self.description = description
        self.scale = scale
        self.wavelengths = list(wavelengths or
                                [LAMBDA_D, LAMBDA_C, LAMBDA_F])
        self.stop = stop
        # no object / image given: an axial point at infinity and the image
        # plane's own aperture, pupils tracking the stop
        # (rayopt/system.py:46-57)
        if not object:
            object = {"type": "infinite", "angle": 0.,
                      "pupil": {"update_radius": True}}
        if not image:
            image = {"type": "finite", "radius": 0., "update_radius": True,
                     "pupil": {"update_radius": True}}
        self.object = Conjugate(object, finite_default=False)
        self.image = Conjugate(image, finite_default=True)
        if fields is None:      # rayopt/system.py:58-63
            fields = [0.] if self.object.point else [0., .7, 1.]
        self.fields = fields
        self.pickups = pickups or []
        self.validators = validators or []
        self.solves = solves or []
        self._engine = None

    def dict(self):
        return {"description": self.description, "stop": self.stop,
                "scale": float(self.scale),
                "wavelengths": [float(w) for w in self.wavelengths],
                "object": self.object.dict(), "image": self.image.dict(),
                "pickups": [dict(p) for p in self.pickups],
                "validators": [dict(v) for v in self.validators],
                "solves": [dict(v) for v in self.solves],
                "elements": [e.dict() for e in self]}

    def update(self):
        """The part of rayopt's ``System.update()`` (rayopt/system.py:
        201-211) the traced path reads afterwards: the first-order pupils.
        The paraxial images of the stop in object and image space, at the
        first wavelength, are stored in ``object.pupil`` / ``image.pupil``
        under the reference's rules (``Pupil.update``, rayopt/pupils.py:
        43-47: ``distance`` unless ``update_distance`` is off, ``radius``
        only if ``update_radius`` is on) -- what
        ``ParaxialTrace.update_conjugates`` does (rayopt/paraxial_trace.py:
        326-341).  Unaimed launches (``object.pupil.aim`` off) and the
        default reference sphere of ``opd()`` use the stored values until
        the next ``update()``, as in the reference; without any ``update()``
        they are evaluated on the fly.  Declarative pickups and solves are
        applied before, validators checked after (rayopt_amd/design.py);
        the paraxial trace itself is a design tool outside the accelerated
        path."""
        from .aiming import entrance_pupil, exit_pupil
        self.__dict__.pop("_reference_aimers", None)    # their guess caches
        self.pickup()
        self.solve()
        l = self.wavelengths[0]
        self.object.pupil["refractive_index"] = self.refractive_index(l, 0)
        self.image.pupil["refractive_index"] = self.refractive_index(l, -1)
        self._update_pupils(entrance_pupil, exit_pupil)
        self.validate()
        return self

    def _update_pupils(self, entrance_pupil, exit_pupil):
        l = self.wavelengths[0]
        try:
            found = ((self.object, entrance_pupil(self, l)),
                     (self.image, exit_pupil(self, l)))
        except (ZeroDivisionError, IndexError, np.linalg.LinAlgError):
            return
        apertures = (self[0].radius, self[-1].radius)
        for (conjugate, (distance, radius)), field in zip(found, apertures):
            # the field extent follows the aperture of the first / last
            # element where asked to (rayopt/conjugates.py:120-123,190-194)
            if conjugate.extra.get("update_radius", False):
                if conjugate.finite:
                    conjugate.radius = field
                elif np.isfinite(distance):
                    conjugate.angle = float(np.arctan2(field, distance))
            pupil = conjugate.pupil
            if pupil.get("update_distance", True) and np.isfinite(distance):
                pupil["distance"] = float(distance)
            if pupil.get("update_radius", False) and np.isfinite(radius):
                pupil_set_radius(pupil, float(radius))

    def _walk(self, path):
        node = self
        for key in path:
            node = getattr(node, key) if isinstance(key, str) else node[key]
        return node

    def get_path(self, path):
        """Follow ``path`` from the system: strings are attributes, anything
        else an index -- ``(2, "curvature")`` is ``self[2].curvature``
        (rayopt/system.py:112-119)."""
        return self._walk(path)

    def set_path(self, path, value):
        """Assign to what :meth:`get_path` reads (rayopt/system.py:121-132)."""
        *head, last = path
        node = self._walk(head)
        if isinstance(last, str):
            setattr(node, last, value)
        else:
            node[last] = value

    @property
    def aperture(self):
        return self[self.stop]

    def refractive_index(self, wavelength, index):
        """Index of the medium behind element ``index``: the last material
        at or before it (1. if none)."""
        if index < 0:
            index += len(self)
        for el in reversed(self[:index + 1]):
            mat = getattr(el, "material", None)
            if mat is not None:
                return mat.refractive_index(wavelength)
        return 1.

    @property
    def origins(self):
        """Vertex positions in the global frame, (L,3)."""
        return np.cumsum([el.offset for el in self], axis=0)

    @property
    def path(self):
        return np.cumsum([el.distance for el in self])

    @property
    def track(self):
        return self.origins[:, 2]

    def align(self, n):
        """Tilt every element for the axial ray: element j is aligned to
        send its incident axis into the direction of element j+1, given the
        indices ``n[j]`` behind the elements; the image is left untilted
        (rayopt/system.py:430-436)."""
        before = n[0]
        for j in range(len(self) - 1):
            self[j].align(self[j + 1].direction, before/n[j])
            before = n[j]
        self[-1].angles = (0., 0., 0.)

    @property
    def mirrored(self):
        return np.cumprod([-1 if getattr(getattr(el, "material", None),
                                         "mirror", False) else 1
                           for el in self])

    # -- launch side of the path: pupils and launch rays ----------------------
    def pupil(self, yo, l=None, stop=None, aiming="device", engine=None):
        """``(z, a)``: distance of the aimed pupil of field ``yo`` from the
        vertex of the first element and its apertures
        ``[[-sag, -mer], [+sag, +mer]]`` (rayopt/system.py:585-593).  Chief and
        marginal rays are aimed as the object pupil's ``aim`` flag says
        (``stop=-1``: marginal rays to the rim of the limiting aperture).
        ``aiming="device"``: the batched kernel, iterated to 1e-9;
        ``"reference"``: rayopt's procedure -- solvers, tolerances, guess
        cache -- as rayopt_amd/aiming_reference.py restates it; ``"rayopt"``:
        the installed rayopt's own methods on device traces
        (rayopt_amd/dropin/aiming_rayopt.py)."""
        if stop not in (None, -1):
            raise NotImplementedError("pupil(): stop is None or -1")
        if engine is None:
            from .engine import get_engine
            engine = get_engine()
        wavelength = self.wavelengths[0] if l is None else l
        if aiming in ("reference", "rayopt"):
            from .aiming_reference import reference_aimer
            return reference_aimer(self, engine, wavelength, stop, l,
                                   aiming).pupil(yo)
        from .aiming import FieldAimer
        z, a = FieldAimer(self, wavelength, engine, aim=None).pupil(
            [yo], rim=(stop == -1))
        return float(z[0]), a[0]

    def aim(self, yo, yp=None, z=None, a=None, filter=True, l=None,
            engine=None):
        """Launch rays ``(y, u)`` -- host arrays (N,3) -- of field ``yo``
        through the normalised pupil coordinates ``yp`` (default: the chief
        ray), for the pupil ``(z, a)`` (default: the first-order pupil):
        ``System.aim`` of the reference (rayopt/system.py:503-504 ->
        ``Conjugate.aim``, rayopt/conjugates.py:137-166,236-255).  The rays
        are built by the device's generation kernel; ``filter`` drops the
        pupil points outside the aimed ellipse (``Pupil.map``)."""
        from .aiming import start_pupil
        from .geometric_trace import GeometricTrace
        if engine is None:
            from .engine import get_engine
            engine = get_engine()
        wavelength = self.wavelengths[0] if l is None else l
        if z is None or a is None:
            z0, a0 = start_pupil(self, wavelength)
            z = z0 if z is None else z
            a = a0*np.array([[-1., -1.], [1., 1.]]) if a is None else a
        yp = np.zeros((1, 2)) if yp is None else np.atleast_2d(
            np.asarray(yp, dtype=float))
        if filter and np.ndim(a) == 2:
            af = np.arctan2(a, z) if self.object.finite else np.asarray(a)
            am = np.fabs(af).max()
            centre, half = np.sum(af, axis=0)/2, np.diff(af, axis=0)/2
            yp = yp[(np.square(yp*am - centre)/np.square(half)).sum(1) <= 1]
        trace = GeometricTrace(self, engine=engine)
        trace.rays_fields([yo], yp, [z], np.asarray(a, dtype=float)[None]
                          if np.ndim(a) == 2 else a, wavelength)
        return np.array(trace.y[0]), np.array(trace.u[0])

    def propagate(self, y, u, n, l, start=1, stop=None, clip=False):
        """Generator with the reference's contract (system.py:459-464):
        yields ``(y, u, n, i, t)`` per element of ``self[start:stop]``.
        ``y, u`` are in the global orientation relative to the vertex of
        element ``start-1``.  The whole march runs as one fused GPU trace;
        the tuples are then handed out surface by surface."""
        from .engine import march_rows
        yield from march_rows(self, y, u, n, l, start, stop, clip)
