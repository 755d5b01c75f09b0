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


def zmx_to_system(text):
    air = BASIC["air"]
    s = System()
    s.append(Spheroid(material=air))
    s.wavelengths = []
    thickness = 0.
    for raw in text.splitlines():
        parts = raw.strip().split(None, 1)
        if not parts:
            continue
        cmd = parts[0]
        args = parts[1].split() if len(parts) == 2 else []
        el = s[-1]
        if cmd == "UNIT":
            s.scale = UNITS[args[0].upper()]
        elif cmd == "NAME":
            s.description = parts[1].strip().strip('"') if args else ""
        elif cmd == "SURF":
            s.append(Spheroid(distance=thickness, material=air))
        elif cmd == "CURV":
            el.curvature = float(args[0])
        elif cmd == "DISZ":
            thickness = float(args[0])       # "INFINITY" -> inf
        elif cmd == "DIAM":
            el.radius = float(args[0])
        elif cmd == "GLAS":
            el.material = _glass(args)
        elif cmd == "MIRR" and args and int(float(args[0])) == 2:
            pass                             # substrate flag, not a mirror
        elif cmd == "CONI":
            el.conic = float(args[0])
        elif cmd == "PARM":
            i, v = int(args[0]) - 1, float(args[1])
            if i < 0:
                continue
            if el.aspherics is None:
                el.aspherics = []
            while len(el.aspherics) <= i:
                el.aspherics.append(0.)
            el.aspherics[i] = v
        elif cmd == "STOP":
            el.stop = True
            s.stop = len(s) - 1
        elif cmd == "WAVL":
            s.wavelengths = [float(w)*1e-6 for w in args]
        elif cmd == "WAVM":
            if float(args[2]) > 0 or not s.wavelengths:
                s.wavelengths.append(float(args[1])*1e-6)
    if not s.wavelengths:
        s.wavelengths = [587.56e-9]
    return s
