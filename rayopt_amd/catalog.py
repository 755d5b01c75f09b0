"""Glass catalogues from Zemax ``.agf``, OSLO ``.glc`` and CODE V ``.xml``
files and refractiveindex.info pages.

The reference keeps its glasses in an SQLite library filled by parsers
(rayopt/library.py, rayopt/zemax.py:186-268); the data files are the
vendors'.  Here a catalogue is read straight from the ``.agf`` text the user
points to and glass *names* (``N-BK7``, ``schott/N-BK7``) then resolve in
``Material.make`` -- hence in YAML prescriptions and in the ``.zmx`` /
``.len`` importers -- to :class:`rayopt_amd.model.DispersionGlass` with the
same dispersion formulas and coefficient order as the reference's
``agf_to_material``.  Host-side setup only: what reaches the GPU is
``n(lambda)`` per surface in the packed table.

``NM name formula glasscode nd vd exclude status``; ``CD`` the dispersion
coefficients; ``GC`` comment; ``ED`` expansion/density; ``LD`` the valid
wavelength range in micrometres.
"""
import codecs
import os

import numpy as np

from .model import DispersionGlass

# .agf formula number -> dispersion formula (rayopt/zemax.py:231-235).  The
# reference lists "sellmeier2", "extended1" and "sellmeier5" but has no
# formula under those names (it warns and cannot evaluate them); they are
# served here by the published Zemax forms: Extended = Schott with more
# inverse powers, Sellmeier 5 = five squared Sellmeier terms.
AGF_FORMULAS = (
    "schott", "sellmeier_squared", "herzberger", "sellmeier2", "conrady",
    "sellmeier_squared", "handbook_of_optics1", "handbook_of_optics2",
    "sellmeier_squared_offset", "schott", "sellmeier_squared", "extended2",
    "hikari")

def _number(text):
    try:
        return float(text)
    except ValueError:
        return float("nan")


def _open_text(path):
    with open(path, "rb") as f:
        head = f.read(4)
    enc = "utf-16" if head.startswith((codecs.BOM_UTF16_LE,
                                       codecs.BOM_UTF16_BE)) else "latin1"
    with open(path, encoding=enc) as f:
        return f.read()


def parse_agf(text):
    """``{name: DispersionGlass}`` from the text of an ``.agf`` catalogue."""
    glasses, glass = {}, None
    for line in text.splitlines():
        line = line.strip()
        if not line or line.startswith("!"):
            continue
        cmd, _, rest = line.partition(" ")
        args = rest.split()
        if cmd == "NM":
            formula = int(_number(args[1]))
            if not 1 <= formula <= len(AGF_FORMULAS):
                raise ValueError("glass %s: unknown .agf dispersion formula "
                                 "%d" % (args[0], formula))
            glass = DispersionGlass(AGF_FORMULAS[formula - 1], [],
                                    name=args[0])
            glass.glasscode = _number(args[2]) if len(args) > 2 else None
            # the catalogue's stated values (nd / vd themselves are
            # computed from the formula, model.Material)
            glass.nd_stated = _number(args[3]) if len(args) > 3 else None
            glass.vd_stated = _number(args[4]) if len(args) > 4 else None
            glass.status = int(_number(args[6])) if len(args) > 6 else None
            glasses[glass.name] = glass
        elif glass is None or cmd == "CC":
            continue
        elif cmd == "CD":
            glass.coefficients = np.array([_number(a) for a in args])
        elif cmd == "GC":
            glass.comment = rest.strip()
        elif cmd == "ED" and len(args) >= 3:
            glass.alpham3070, glass.alpha20300, glass.density = (
                _number(a) for a in args[:3])
        elif cmd == "LD" and len(args) >= 2:
            glass.lambda_min, glass.lambda_max = (_number(a)
                                                  for a in args[:2])
    return glasses


# OSLO .glc dispersion type -> formula (rayopt/oslo.py:201-205)
GLC_FORMULAS = {1: "schott", 2: "sellmeier_squared_transposed", 3: "conrady",
                6: "hikari"}


def parse_glc(text):
    """``{name: DispersionGlass}`` from the text of an OSLO ``.glc``
    catalogue: a header line ``version count name``, then one glass per line
    -- ``name nd vd density``, eight fields not used here, the dispersion
    type, the coefficient count and the coefficients
    (rayopt/oslo.py:169-205)."""
    glasses = {}
    lines = text.splitlines()
    for line in lines[1:]:
        fields = line.split()
        if len(fields) < 14:
            continue
        name = fields[0]
        kind, count = int(_number(fields[12])), int(_number(fields[13]))
        if kind not in GLC_FORMULAS:
            continue                      # no published formula: skip
        glass = DispersionGlass(
            GLC_FORMULAS[kind],
            [_number(f) for f in fields[14:14 + count]], name=name)
        glass.nd_stated, glass.vd_stated, glass.density = (
            _number(f) for f in fields[1:4])
        glasses[name] = glass
    return glasses


# refractiveindex.info page: DATA[].type "formula N" (rayopt/rii.py:80-90)
RII_FORMULAS = {1: "sellmeier_offset", 2: "sellmeier_squared_offset",
                3: "polynomial", 4: "refractiveindex_info", 5: "cauchy",
                6: "gas_offset", 7: "herzberger", 8: "retro", 9: "exotic"}


def parse_rii(text, name="rii"):
    """One refractiveindex.info database page (YAML with a ``DATA`` list
    holding a ``formula N`` entry: ``range``, ``coefficients``) as a
    :class:`DispersionGlass` (rayopt/rii.py:95-111)."""
    import yaml
    page = yaml.safe_load(text)
    glass = None
    for entry in page["DATA"]:
        kind = str(entry["type"])
        if kind.startswith("formula"):
            glass = DispersionGlass(
                RII_FORMULAS[int(kind.split()[1])],
                [_number(c) for c in str(entry["coefficients"]).split()],
                name=name)
            lo, hi = str(entry.get("range", "nan nan")).split()[:2]
            glass.lambda_min, glass.lambda_max = _number(lo), _number(hi)
    if glass is None:
        raise ValueError("refractiveindex.info page %r holds no dispersion "
                         "formula (tabulated data only)" % name)
    glass.comment = page.get("COMMENTS")
    glass.references = page.get("REFERENCES")
    return glass


# CODE V glass catalogue XML: EquationType -> formula (rayopt/codev.py:52-67)
CODEV_FORMULAS = {
    "Standard Sellmeier": "sellmeier",
    "Glass Manufacturer Sellmeier": "sellmeier_squared_offset",
    "Laurent": "schott", "Glass Manufacturer Laurent": "schott",
    "Herzberger": "herzberger", "Cauchy": "conrady"}


def parse_codev_xml(text):
    """``{name: DispersionGlass}`` from a CODE V glass catalogue in XML
    (``<Glasses><Glass>`` with ``GlassName`` carrying the catalogue ``ID`` as
    prefix, ``EquationType``, ``DispersionCoefficients/Coefficient``;
    rayopt/codev.py:32-67)."""
    import xml.etree.ElementTree as et
    root = et.fromstring(text)
    prefix = root.findtext("./ID") or ""
    glasses = {}
    for node in root.iterfind("./Glasses/Glass"):
        full = node.findtext("./GlassName")
        name = full[len(prefix):] if full.startswith(prefix) else full
        glass = DispersionGlass(
            CODEV_FORMULAS[node.findtext("./EquationType")],
            [float(c.text) for c in node.iterfind(
                "./DispersionCoefficients/Coefficient")], name=name)
        glass.comment = node.findtext("./NumericName")
        glasses[name] = glass
    return glasses


class GlassCatalogs:
    """The catalogues loaded in this process, by name (file stem, lower
    case).  ``find("N-BK7")`` searches all of them in loading order,
    ``find("schott/N-BK7")`` one."""
    def __init__(self):
        self.catalogs = {}

    def load(self, path, name=None):
        stem, ext = os.path.splitext(os.path.basename(path))
        name = (name or stem).lower()
        parse = {".glc": parse_glc, ".xml": parse_codev_xml}.get(
            ext.lower(), parse_agf)
        self.catalogs[name] = parse(_open_text(path))
        return self.catalogs[name]

    def add(self, name, glasses):
        self.catalogs[name.lower()] = dict(glasses)

    def clear(self):
        self.catalogs.clear()

    def find(self, spec):
        parts = str(spec).split("/")
        if len(parts) == 2:
            books = [self.catalogs.get(parts[0].lower(), {})]
        elif len(parts) == 1:
            books = list(self.catalogs.values())
        else:
            return None
        key = parts[-1]
        for book in books:
            if key in book:
                return book[key]
            for name, glass in book.items():      # Zemax names: any case
                if name.upper() == key.upper():
                    return glass
        return None


catalogs = GlassCatalogs()
load_agf = load_glc = load = catalogs.load
