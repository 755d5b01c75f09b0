"""OSLO ``.len`` prescriptions -> System.

A ``.len`` file is a flat list of ``OPCODE args...`` lines; surfaces are
delimited by ``NXT`` (and the final ``END``), everything in between describes
the surface being built.  The importer is a dispatch table from opcode to a
small action on the parse state -- the operands rayopt's importer understands
(rayopt/oslo.py:129-167: LEN UNI AIR TH AP GLA AST RD NXT END), with the same
meaning, including two conventions of that importer that a user's files rely
on: ``TH`` is the gap *behind* the surface, so it becomes the ``distance`` of
the NEXT element, and a gap above 100 lens units stands for infinity.  Glasses
resolve through ``Material.make`` (``n/v`` strings, basic names); ``AST`` also
records ``system.stop``.  Opcodes that carry no geometry are ignored; unknown
ones are collected in ``system.oslo_unhandled`` instead of being printed.
"""
import math

from .model import System, Spheroid, Material, BASIC

INFINITE_BEYOND = 100.      # lens units; the reference's threshold
IGNORED = frozenset(["//", "DES", "EBR", "GIH", "DLRS", "WW", "WV"])


class _LenParse:
    """What is known while walking the file: the system so far, the surface
    under construction and the gap that will precede the next one."""
    def __init__(self):
        self.system = System()
        self.system.wavelengths = [587.56e-9]
        self.surface = Spheroid()
        self.gap = 0.
        self.unhandled = []

    def close_surface(self):
        self.system.append(self.surface)
        self.surface = Spheroid()
        self.surface.distance = self.gap


def _title(st, words):
    # LEN NEW "name with blanks" <two trailing numbers>
    st.system.description = " ".join(words[1:-2]).strip('"')


def _units(st, words):
    st.system.scale = float(words[0])*1e-3          # given in mm


def _air(st, words):
    st.surface.material = BASIC["air"]


def _glass(st, words):
    st.surface.material = Material.make(words[0])


def _gap(st, words):
    gap = float(words[0])
    st.gap = math.inf if gap > INFINITE_BEYOND else gap


def _aperture(st, words):
    value = words[1] if words[0] == "CHK" else words[0]
    st.surface.radius = float(value)


def _stop(st, words):
    st.surface.stop = True
    st.system.stop = len(st.system)     # index this surface will get


def _radius_of_curvature(st, words):
    st.surface.curvature = 1/float(words[0])


def _next(st, words):
    st.close_surface()


ACTIONS = {"LEN": _title, "UNI": _units, "AIR": _air, "GLA": _glass,
           "TH": _gap, "AP": _aperture, "AST": _stop,
           "RD": _radius_of_curvature, "NXT": _next, "END": _next}


def len_to_system(text):
    """Parse the text of a ``.len`` file (or an open file)."""
    if hasattr(text, "read"):
        text = text.read()
    st = _LenParse()
    for line in text.splitlines():
        opcode, *words = line.split() or [None]
        action = ACTIONS.get(opcode)
        if action is not None:
            action(st, words)
        elif opcode is not None and opcode not in IGNORED:
            st.unhandled.append(line.strip())
    st.system.oslo_unhandled = st.unhandled
    return st.system
