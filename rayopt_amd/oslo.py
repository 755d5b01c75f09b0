"""OSLO ``.len`` prescriptions -> System: the operands rayopt's importer
reads (rayopt/oslo.py:129-167) -- LEN, UNI, AIR, TH, AP, GLA, AST, RD, NXT,
END -- with the same meaning (``TH`` is the thickness behind the surface;
values above 100 are treated as infinite, as the reference does).  Glasses
resolve through ``Material.make`` (``n/v`` strings, basic names); ``AST`` also
sets ``system.stop``."""
import numpy as np

from .model import System, Spheroid, Material, BASIC


def len_to_system(text):
    s = System()
    s.wavelengths = [587.56e-9]
    el = Spheroid()
    thickness = 0.
    for raw in text.splitlines():
        p = raw.split()
        if not p:
            continue
        cmd, args = p[0], p[1:]
        if cmd == "LEN":
            s.description = " ".join(args[1:-2]).strip('"')
        elif cmd == "UNI":
            s.scale = float(args[0])*1e-3
        elif cmd == "AIR":
            el.material = BASIC["air"]
        elif cmd == "TH":
            thickness = float(args[0])
            if thickness > 1e2:
                thickness = np.inf
        elif cmd == "AP":
            if args[0] == "CHK":
                args = args[1:]
            el.radius = float(args[0])
        elif cmd == "GLA":
            el.material = Material.make(args[0])
        elif cmd == "AST":
            el.stop = True
            s.stop = len(s)
        elif cmd == "RD":
            el.curvature = 1/float(args[0])
        elif cmd in ("NXT", "END"):
            s.append(el)
            el = Spheroid()
            el.distance = thickness
    return s
