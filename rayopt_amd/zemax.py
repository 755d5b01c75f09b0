"""Zemax ``.zmx`` sequential prescriptions -> System.

Reads the operands rayopt's importer reads (rayopt/zemax.py:90-183) with the
same meaning: UNIT, NAME, SURF, CURV, DISZ (thickness *behind* the surface,
i.e. the distance of the next one), DIAM (semi-diameter), GLAS, CONI, PARM
(``PARM i`` = even-asphere coefficient of r^(2i): ``aspherics[i-1]``), STOP,
WAVL/WAVM, MIRR.  Like the reference the result starts with an extra air
element in front of ``SURF 0``.  Glasses: without a catalogue database only
model glasses can be resolved -- the ``GLAS`` line's index/Abbe fields
(``GLAS name 1 0 nd vd ...``) -> Abbe model; ``MIRROR`` -> mirror.  Unlike the
reference, ``STOP`` also sets ``system.stop``.
"""
from .model import System, Spheroid, Material, AbbeGlass, BASIC

UNITS = {"MM": 1e-3, "CM": 1e-2, "M": 1., "METER": 1., "INCH": 25.4e-3,
         "IN": 25.4e-3}


def _glass(args):
    name = args[0].upper()
    if name == "MIRROR":
        return BASIC["mirror"]
    try:
        return Material.make(args[0])
    except (KeyError, ValueError):
        pass
    try:
        nd, vd = float(args[3]), float(args[4])
    except (IndexError, ValueError):
        raise KeyError("glass %r: no catalogue and no model nd/vd on the "
                       "GLAS line" % args[0])
    return AbbeGlass(nd, vd if vd else float("inf"), name=args[0])


class _ZmxParse:
    """What is known while walking a ``.zmx`` file: the system so far (its
    last element is the surface the next operands apply to) and the
    thickness that will precede the next ``SURF``."""
    def __init__(self):
        self.system = System()
        self.system.append(Spheroid(material=BASIC["air"]))
        self.system.wavelengths = []
        self.thickness = 0.

    @property
    def surface(self):
        return self.system[-1]


def _unit(st, words, rest):
    st.system.scale = UNITS[words[0].upper()]


def _name(st, words, rest):
    st.system.description = rest.strip().strip('"') if words else ""


def _surface(st, words, rest):
    st.system.append(Spheroid(distance=st.thickness, material=BASIC["air"]))


def _attribute(name):
    def action(st, words, rest):
        setattr(st.surface, name, float(words[0]))   # "INFINITY" -> inf
    return action


def _thickness(st, words, rest):
    st.thickness = float(words[0])


def _material(st, words, rest):
    st.surface.material = _glass(words)


def _even_asphere_term(st, words, rest):
    # PARM i = coefficient of r^(2 i); PARM 0 is not a polynomial term
    term, value = int(words[0]) - 1, float(words[1])
    if term >= 0:
        terms = st.surface.aspherics or []
        terms += [0.]*(term + 1 - len(terms))
        terms[term] = value
        st.surface.aspherics = terms


def _stop(st, words, rest):
    st.surface.stop = True
    st.system.stop = len(st.system) - 1


def _wavelengths(st, words, rest):
    st.system.wavelengths = [float(w)*1e-6 for w in words]


def _wavelength_entry(st, words, rest):
    # WAVM index micrometres weight
    if float(words[2]) > 0 or not st.system.wavelengths:
        st.system.wavelengths.append(float(words[1])*1e-6)


ACTIONS = {
    "UNIT": _unit, "NAME": _name, "SURF": _surface,
    "CURV": _attribute("curvature"), "DIAM": _attribute("radius"),
    "CONI": _attribute("conic"), "DISZ": _thickness, "GLAS": _material,
    "PARM": _even_asphere_term, "STOP": _stop, "WAVL": _wavelengths,
    "WAVM": _wavelength_entry,
    # MIRR 2 is a substrate flag, not a mirror: nothing to do
}


def zmx_to_system(text):
    """Parse the text of a ``.zmx`` file (or an open file)."""
    if hasattr(text, "read"):
        text = text.read()
    st = _ZmxParse()
    for line in text.splitlines():
        opcode, _, rest = line.strip().partition(" ")
        action = ACTIONS.get(opcode)
        if action is not None:
            action(st, rest.split(), rest)
    if not st.system.wavelengths:
        st.system.wavelengths = [587.56e-9]
    return st.system
