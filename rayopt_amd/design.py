"""Design-time housekeeping of a System: what a prescription carries besides
its surfaces and what a user does to it between traces.

None of this touches rays; it is O(len(system)) scalar work on the host that
the traced path depends on only through the element attributes it leaves
behind.  Behaviour follows the reference so that its own fixtures load and
print the same way:

* pupils of kind ``radius`` / ``slope`` / ``na`` / ``fno``
  (rayopt/pupils.py:108-230) -- here plain dictionaries, the kind read from
  ``type`` or from the key that is present;
* declarative pickups, solves and validators (rayopt/system.py:133-247:
  ``get`` / ``set`` paths, ``factor``, ``offset``, ``target``, ``minimum``,
  ``maximum``, ``equality``).  The reference also accepts Python source
  (``get_eval``, ``get_func``, ``set_exec``, ``exec``); prescriptions are
  data here and such entries are refused, not evaluated;
* ``text()`` / ``str(system)`` (rayopt/system.py:281-309,395-402), ``rescale``,
  ``reverse``, ``resize_convex``, ``edge_thickness``, ``groups``
  (rayopt/system.py:89-110,311-352,376-393).
"""
import math

import numpy as np

_CODE_KEYS = ("get_eval", "get_func", "set_exec", "set_func", "exec")


# --------------------------------------------------------------------------
# pupils (dictionaries)
# --------------------------------------------------------------------------

def pupil_kind(pupil):
    kind = pupil.get("type")
    if kind:
        return kind
    for key in ("na", "fno", "slope"):
        if key in pupil:
            return key
    return "radius"


def _tan_of_asin(x):
    return x/math.sqrt(1. - x*x)


def _sin_of_atan(x):
    return x/math.sqrt(1. + x*x)


def pupil_slope(pupil):
    """tan of the marginal ray angle (rayopt/pupils.py:85-87,129-131,
    212-218)."""
    kind = pupil_kind(pupil)
    index = pupil.get("refractive_index", 1.)
    if kind == "slope":
        return pupil["slope"]
    if kind == "na":
        return _tan_of_asin(pupil["na"]/index)
    if kind == "fno":
        return _tan_of_asin(1/(2.*pupil["fno"])/index)
    return pupil.get("radius", 0.)/pupil.get("distance", 1.)


def pupil_radius(pupil):
    """Pupil radius, or None where the prescription gives none."""
    if pupil_kind(pupil) == "radius":
        return pupil.get("radius")
    return pupil_slope(pupil)*pupil.get("distance", 1.)


def pupil_set_radius(pupil, radius):
    """Store ``radius`` in the quantity the pupil is specified by."""
    kind = pupil_kind(pupil)
    distance = pupil.get("distance", 1.)
    index = pupil.get("refractive_index", 1.)
    if kind == "slope":
        pupil["slope"] = radius/distance
    elif kind == "na":
        pupil["na"] = index*_sin_of_atan(radius/distance)
    elif kind == "fno":
        pupil["fno"] = 1/(2*index*_sin_of_atan(radius/distance))
    else:
        pupil["radius"] = radius


def pupil_rescale(pupil, scale):
    if "distance" in pupil:
        pupil["distance"] *= scale
    if pupil_kind(pupil) == "radius" and pupil.get("radius") is not None:
        pupil["radius"] *= scale


def pupil_text(pupil):
    yield "Pupil Distance: %g" % pupil.get("distance", 1.)
    if pupil.get("telecentric", False):
        yield "Telecentric: %s" % pupil["telecentric"]
    if pupil.get("refractive_index", 1.) != 1.:
        yield "Refractive Index: %g" % pupil["refractive_index"]
    if pupil.get("projection", "rectilinear") != "rectilinear":
        yield "Projection: %s" % pupil["projection"]
    if not pupil.get("update_distance", True):
        yield "Track Distance: %s" % pupil["update_distance"]
    if pupil.get("update_radius", False):
        yield "Update Radius: %s" % pupil["update_radius"]
    if pupil.get("aim", False):
        yield "Aim: %s" % pupil["aim"]
    label, key = {"radius": ("Radius", "radius"), "slope": ("Slope", "slope"),
                  "na": ("NA", "na"),
                  "fno": ("F-Number", "fno")}[pupil_kind(pupil)]
    value = pupil.get(key)
    yield "%s: %g" % (label, 0. if value is None else value)


def conjugate_text(conjugate):
    if conjugate.finite:
        yield "Radius: %.3g" % conjugate.radius
    else:
        yield "Semi-Angle: %.3g deg" % math.degrees(conjugate.angle)
    projection = conjugate.extra.get("projection", "rectilinear")
    if projection != "rectilinear":
        yield "Projection: %s" % projection
    if conjugate.extra.get("update_radius", False):
        yield "Update Radius: %s" % conjugate.extra["update_radius"]
    yield "Pupil:"
    for line in pupil_text(conjugate.pupil):
        yield "  %s" % line


# --------------------------------------------------------------------------
# surfaces
# --------------------------------------------------------------------------

def sag(element, r):
    """Surface height at radial distance ``r`` (scalar, host): the conic
    term plus the even-asphere polynomial (rayopt/elements.py:440-455)."""
    c = getattr(element, "curvature", 0.)
    k = getattr(element, "conic", 0.)
    r2 = r*r
    z = c*r2/(1 + math.sqrt(1 - (1 + k)*c*c*r2)) if c else 0.
    for i, a in enumerate(getattr(element, "aspherics", None) or ()):
        z += a*r2**(i + 1)
    return z


# --------------------------------------------------------------------------
# System mix-in
# --------------------------------------------------------------------------

def _refuse_code(entry, what):
    for key in _CODE_KEYS:
        if key in entry:
            raise ValueError(
                "%s %r carries Python source (%s): prescriptions are data "
                "here, only get/set paths are applied" % (what, entry, key))


def _label(material):
    """A material's column entry: ``catalog/name``, the name, or ``-`` for a
    medium given by numbers (whose ``str`` is those numbers, so that a
    prescription written out reads back)."""
    name = getattr(material, "name", None)
    if name is None:
        return material
    catalog = getattr(material, "catalog", None)
    return "%s/%s" % (catalog, name) if catalog is not None else name


class DesignMixin:
    # -- pickups / solves / validators ---------------------------------------
    def pickup(self):
        """Copy values between parameters: ``{get: path, set: path,
        factor:, offset:}``."""
        for entry in self.pickups:
            _refuse_code(entry, "pickup")
            value = self.get_path(entry["get"])
            if "factor" in entry:
                value = value*entry["factor"]
            if "offset" in entry:
                value = value + entry["offset"]
            self.set_path(entry["set"], value)

    def solve(self):
        """Drive ``get`` to ``target`` by varying ``set`` (scalar Newton /
        secant, as the reference: tol 1e-8, 20 iterations)."""
        from scipy.optimize import newton
        for entry in self.solves:
            _refuse_code(entry, "solve")
            target = entry.get("target", 0.)

            def residual(x, entry=entry, target=target):
                self.set_path(entry["set"], x)
                self.pickup()
                return self.get_path(entry["get"]) - target

            start = entry["init"] if "init" in entry else \
                self.get_path(entry["set"])
            x = newton(residual, start, tol=entry.get("tol", 1e-8),
                       maxiter=entry.get("maxiter", 20))
            residual(x)
            if "init_current" in entry:
                entry["init"] = float(x)

    def validate(self, fix=False):
        """Check ``{get: path, minimum:, maximum:, equality:}`` entries;
        ``fix`` moves an offending parameter onto its bound instead of
        raising ValueError."""
        tests = (("minimum", lambda v, b: v < b, "<"),
                 ("maximum", lambda v, b: v > b, ">"),
                 ("equality", lambda v, b: v != b, "!="))
        for entry in self.validators:
            _refuse_code(entry, "validator")
            value = self.get_path(entry["get"])
            for key, fails, sign in tests:
                if key in entry and fails(value, entry[key]):
                    if not fix:
                        raise ValueError("%s %s %s (%s)" % (
                            value, sign, entry[key], entry))
                    self.set_path(entry["get"], entry[key])

    # -- geometry of the prescription -----------------------------------------
    def groups(self):
        """Index lists of the lens groups: gas, solids, (mirror, solids)*,
        gas -- or a mirror on its own."""
        open_group = []
        for index, element in enumerate(self):
            if not hasattr(element, "material"):    # a plane without medium
                if open_group:
                    open_group.append(index)
                continue
            medium = element.material
            if getattr(medium, "solid", False):
                open_group.append(index)
            elif open_group or getattr(medium, "mirror", False):
                yield open_group + [index]
                open_group = []
        if open_group:
            yield open_group

    def edge_thickness(self, axis=1):
        """Axial gap in front of every element measured at the rim
        (element radius) instead of the vertex."""
        rim = np.array([el.edge_sag(axis) if hasattr(el, "edge_sag") else 0.
                        for el in self])
        vertex = np.array([el.distance for el in self])
        return (vertex - rim) + np.concatenate([[0.], rim[:-1]])

    @property
    def edge_y(self):
        return self.edge_thickness(axis=1)

    @property
    def edge_x(self):
        return self.edge_thickness(axis=0)

    def resize_convex(self):
        """Make the convex side of a lens at least as large as the surface
        that closes it (a lens can then be edged from one side)."""
        front = None            # (surface that opened a solid, its curvature)
        for surface in self[1:-1]:
            if not hasattr(surface, "material"):
                continue
            curvature = getattr(surface, "curvature", 0)
            if front is not None:
                opener, opened_with = front
                larger = max(surface.radius, opener.radius)
                if curvature <= 0:          # closes convex (or flat)
                    surface.radius = larger
                if opened_with > 0:         # opened convex
                    opener.radius = larger
            starts_solid = (not surface.material) or surface.material.solid
            front = (surface, curvature) if starts_solid else None

    def reverse(self):
        """Turn the system around: the element order, every surface, the
        material behind each surface and the distances swap ends; object and
        image change places."""
        count = len(self)
        distance_behind = [self[i + 1].distance if i + 1 < count else 0.
                           for i in range(count)]
        medium_before = [getattr(self[i - 1], "material", None) if i else None
                         for i in range(count)]
        for element, gap, medium in zip(self, distance_behind, medium_before):
            element.reverse()
            element.distance = gap
            element.material = medium
        self[:] = self[::-1]
        self.object, self.image = self.image, self.object

    def rescale(self, scale=None):
        """Multiply every length by ``scale`` (default: to millimetres)."""
        factor = self.scale/1e-3 if scale is None else scale
        for part in (*self, self.object, self.image):
            part.rescale(factor)
        self.scale = self.scale/factor

    # -- text -------------------------------------------------------------------
    def base_text(self):
        yield "System: %s" % self.description
        yield "Scale: %s mm" % (self.scale/1e-3)
        yield "Wavelengths: %s nm" % ", ".join(
            "%.0f" % (w/1e-9) for w in self.wavelengths)
        yield "Fields: %s" % ", ".join("%g" % f for f in self.fields)
        for title, conjugate in (("Object:", self.object),
                                 ("Image:", self.image)):
            yield title
            for line in conjugate.text():
                yield " " + line
        yield "Stop: %i" % self.stop
        yield "Elements:"
        yield "%2s %1s %10s %10s %10s %17s %7s %7s %7s" % (
            "#", "T", "Distance", "Rad Curv", "Diameter", "Material", "n",
            "nd", "Vd")
        for i, el in enumerate(self):
            c = getattr(el, "curvature", 0)
            mat = getattr(el, "material", "")
            nd = getattr(mat, "nd", np.nan)
            vd = getattr(mat, "vd", np.nan)
            n = self.refractive_index(self.wavelengths[0], i) if mat else nd
            yield "%2i %1s %10.5g %10.4g %10.5g %17s %7.3f %7.3f %7.2f" % (
                i, el.typeletter, el.distance, 1./c if c else np.inf,
                el.radius*2, _label(mat), n, nd, vd)

    def text(self):
        yield from self.base_text()
        yield ""

    def __str__(self):
        return "\n".join(self.text())
