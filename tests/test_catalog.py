"""Glass catalogues from .agf text (rayopt_amd/catalog.py) against the
reference's agf_to_material (rayopt/zemax.py:230-268) and published indices."""
import numpy as np
import pytest

import rayopt_amd as ra
from rayopt_amd import catalog
from oracle import refshim

# a small catalogue in .agf form; coefficients are the vendors' published
# ones (Schott N-BK7 / N-SF6 Sellmeier, the 1992 Schott-formula BK7), the
# remaining entries exercise the other formula numbers
AGF = """CC test catalogue
! a comment line
NM N-BK7 2 517642.251 1.5168 64.17 0 1
GC Schott borosilicate crown
ED 7.1 8.3 2.51 -0.0009 0
CD 1.03961212 0.00600069867 0.231792344 0.0200179144 1.01046945 103.560653
TD 1.86e-06 1.31e-08 -1.37e-11 4.34e-07 6.27e-10 0.17 20
OD 1 1 1 1 2.3 2.3
LD 0.3 2.5
IT 0.3 0.05 25
NM N-SF6 2 805254.337 1.80518 25.36 0 1
CD 1.77931763 0.0133714182 0.338149866 0.0617533621 2.08734474 174.01759
LD 0.37 2.5
NM BK7OLD 1 517642 1.5168 64.17 0 0
CD 2.2718929 -0.010108077 0.010592509 0.00020816965 -7.6472538e-06 4.9240991e-07
NM CONRADYGLASS 5 0 1.52 60 0 0
CD 1.50 0.01 0.0005
NM HERZ 3 0 1.5 60 0 0
CD 1.50 0.004 0.0001 -0.002 0.00001 0.0
NM HOO1 7 0 1.5 60 0 0
CD 2.25 0.01 0.02 0.005
NM HOO2 8 0 1.5 60 0 0
CD 1.9 0.35 0.02 0.005
NM SELL3 6 0 1.5 60 0 0
CD 1.0 0.006 0.23 0.02 1.0 100.0 0.01 0.03
NM SELL4 9 0 1.5 60 0 0
CD 0.2 0.9 0.006 0.9 100.0
NM EXT2 12 0 1.5 60 0 0
CD 2.27 -0.0101 0.0106 0.000208 -7.6e-06 4.9e-07 1e-05 1e-07
NM HIKARI 13 0 1.5 60 0 0
CD 2.27 -0.0101 1e-05 0.0106 0.000208 -7.6e-06 4.9e-07
"""

LINES = (587.5618e-9, 486.1327e-9, 656.2725e-9, 1014e-9, 404.7e-9)


@pytest.fixture
def book():
    catalog.catalogs.clear()
    catalog.catalogs.add("schott", catalog.parse_agf(AGF))
    yield catalog.catalogs
    catalog.catalogs.clear()


def test_published_indices(book):
    bk7 = book.find("N-BK7")
    assert bk7.typ == "sellmeier_squared" and bk7.nd_stated == 1.5168
    assert bk7.glasscode == 517642.251 and bk7.status == 1
    assert bk7.comment == "Schott borosilicate crown"
    assert (bk7.lambda_min, bk7.lambda_max) == (0.3, 2.5)
    assert bk7.density == 2.51
    nd, nf, nc = (bk7.refractive_index(l) for l in LINES[:3])
    assert nd == pytest.approx(1.5168, abs=2e-5)
    assert (nd - 1)/(nf - nc) == pytest.approx(64.17, abs=0.02)
    sf6 = book.find("schott/N-SF6")
    assert sf6.refractive_index(LINES[0]) == pytest.approx(1.80518, abs=2e-5)
    old = book.find("BK7OLD")
    assert old.typ == "schott"
    assert old.refractive_index(LINES[0]) == pytest.approx(1.5168, abs=2e-5)
    assert book.find("n-bk7") is bk7                  # any case, as Zemax
    assert book.find("ohara/N-BK7") is None and book.find("NOPE") is None


@pytest.mark.skipif(not refshim.available(), reason="no /root/reference")
def test_every_formula_matches_the_reference_parser(book):
    """Each glass block through the reference's agf_to_material and through
    parse_agf: same formula name, coefficients and n(lambda)."""
    refshim.load()
    import importlib
    ref_zemax = importlib.import_module("rayopt.zemax")
    blocks, cur = {}, None
    for line in AGF.splitlines():
        if line.startswith("NM "):
            cur = line.split()[1]
            blocks[cur] = ""
        if cur and not line.startswith(("CC", "!")):
            blocks[cur] += line + "\n"
    assert len(blocks) == 11
    for name, block in blocks.items():
        mine = book.find(name)
        theirs = ref_zemax.agf_to_material(block)
        assert theirs.typ == mine.typ and theirs.name == mine.name
        assert np.array_equal(theirs.coefficients, mine.coefficients)
        for l in LINES:
            assert mine.refractive_index(l) == pytest.approx(
                float(theirs.refractive_index(l)), rel=1e-15), (name, l)


def test_formulas_the_reference_lacks():
    """Sellmeier 2, Extended and Sellmeier 5 have no formula in the
    reference; the published forms are used."""
    g = catalog.parse_agf(
        "NM S2 4 0 1.5 60\nCD 0.1 1.1 0.09 0.01 10.0\n"
        "NM EXT 10 0 1.5 60\nCD 2.27 -0.0101 0.0106 0.000208 -7.6e-06 "
        "4.9e-07 1e-09 1e-10\n"
        "NM S5 11 0 1.5 60\nCD 0.6 0.004 0.4 0.01 0.2 0.02 0.9 100. 0.1 "
        "150.\n")
    w = 0.5875618
    c = g["S2"].coefficients
    assert g["S2"].refractive_index(w*1e-6)**2 == pytest.approx(
        1 + c[0] + c[1]*w**2/(w**2 - c[2]**2) + c[3]/(w**2 - c[4]**2))
    c = g["EXT"].coefficients
    assert g["EXT"].refractive_index(w*1e-6)**2 == pytest.approx(
        c[0] + c[1]*w**2 + sum(c[2 + i]*w**(-2*(i + 1)) for i in range(6)))
    c = g["S5"].coefficients.reshape(5, 2)
    assert g["S5"].refractive_index(w*1e-6)**2 == pytest.approx(
        1 + sum(k*w**2/(w**2 - l) for k, l in c))
    with pytest.raises(ValueError):
        catalog.parse_agf("NM BAD 14 0 1.5 60\n")


def test_names_resolve_in_prescriptions_and_zmx(book, tmp_path):
    s = ra.system_from_yaml("""
elements:
- {material: air}
- {roc: 50, distance: 10, material: N-BK7, radius: 10}
- {roc: -50, distance: 4, material: schott/N-SF6, radius: 10}
- {distance: 40, material: air, radius: 10}
""")
    l = s.wavelengths[0]
    assert s.refractive_index(l, 1) == book.find("N-BK7").refractive_index(l)
    assert s.refractive_index(l, 2) == book.find("N-SF6").refractive_index(l)
    from rayopt_amd.zemax import zmx_to_system
    z = zmx_to_system("UNIT MM\nSURF 0\n  DISZ INFINITY\nSURF 1\n  CURV "
                      "0.02\n  DISZ 4\n  GLAS N-BK7 0 0 1.5 60\n  DIAM 10\n"
                      "SURF 2\n  DISZ 90\n  DIAM 10\nSURF 3\n  DIAM 1\n")
    assert z[2].material is book.find("N-BK7")
    with pytest.raises(KeyError):
        ra.Material.make("N-LAK9")
    # utf-16 files (Zemax writes them) and latin1 files both load
    for enc, name in (("utf-16", "u16"), ("latin1", "l1")):
        path = tmp_path / (name + ".agf")
        path.write_bytes(AGF.encode(enc))
        got = catalog.load_agf(str(path))
        assert len(got) == 11 and name in catalog.catalogs.catalogs
        assert ra.Material.make(name + "/HIKARI").typ == "hikari"


GLC = """1.05 3 TESTCAT
BK7X 1.5168 64.17 2.51 0 0 0 0 0 0 0 0 1 6 2.2718929 -0.010108077 0.010592509 0.00020816965 -7.6472538e-06 4.9240991e-07 1 2 0.3 2.5
SELLT 1.5168 64.17 2.51 0 0 0 0 0 0 0 0 2 6 1.03961212 0.231792344 1.01046945 0.00600069867 0.0200179144 103.560653
CONR 1.52 60 2.5 0 0 0 0 0 0 0 0 3 3 1.50 0.01 0.0005
ODD 1.52 60 2.5 0 0 0 0 0 0 0 0 4 3 1.50 0.01 0.0005
"""


def test_glc_catalogue(tmp_path):
    g = catalog.parse_glc(GLC)
    assert sorted(g) == ["BK7X", "CONR", "SELLT"]      # type 4: no formula
    assert g["BK7X"].refractive_index(LINES[0]) == pytest.approx(
        1.5168, abs=2e-5)
    assert g["SELLT"].refractive_index(LINES[0]) == pytest.approx(
        1.5168, abs=2e-5)
    assert g["SELLT"].density == 2.51 and g["SELLT"].vd_stated == 64.17
    path = tmp_path / "mycat.glc"
    path.write_text(GLC)
    try:
        catalog.load(str(path))
        assert ra.Material.make("mycat/CONR").typ == "conrady"
    finally:
        catalog.catalogs.clear()
    if not refshim.available():
        return
    refshim.load()
    import importlib
    ref_oslo = importlib.import_module("rayopt.oslo")
    for line in GLC.splitlines()[1:4]:
        theirs = ref_oslo.glc_to_material(line)
        mine = g[theirs.name]
        assert theirs.typ == mine.typ
        assert np.array_equal(theirs.coefficients, mine.coefficients)
        for l in LINES:
            assert mine.refractive_index(l) == pytest.approx(
                float(theirs.refractive_index(l)), rel=1e-15)


RII_PAGE = """
REFERENCES: "a reference"
COMMENTS: "fused silica, 20 C"
DATA:
  - type: formula 1
    range: 0.21 6.7
    coefficients: 0 0.6961663 0.0684043 0.4079426 0.1162414 0.8974794 9.896161
  - type: tabulated k
    data: |
        0.21 1e-9
"""

CODEV_XML = """<Catalog><Name>Test</Name><ID>T_</ID><Glasses>
<Glass><GlassName>T_SILICA</GlassName><NumericName>458678</NumericName>
<Availability>1</Availability><EquationType>Standard Sellmeier</EquationType>
<DispersionCoefficients><Coefficient>0.6961663</Coefficient>
<Coefficient>0.0684043</Coefficient><Coefficient>0.4079426</Coefficient>
<Coefficient>0.1162414</Coefficient><Coefficient>0.8974794</Coefficient>
<Coefficient>9.896161</Coefficient></DispersionCoefficients></Glass>
<Glass><GlassName>T_LAUR</GlassName><NumericName>517642</NumericName>
<Availability>1</Availability><EquationType>Laurent</EquationType>
<DispersionCoefficients><Coefficient>2.2718929</Coefficient>
<Coefficient>-0.010108077</Coefficient><Coefficient>0.010592509</Coefficient>
<Coefficient>0.00020816965</Coefficient><Coefficient>-7.6472538e-06</Coefficient>
<Coefficient>4.9240991e-07</Coefficient></DispersionCoefficients></Glass>
</Glasses></Catalog>"""


def test_rii_page_and_codev_catalogue(tmp_path):
    silica = catalog.parse_rii(RII_PAGE, "SiO2|Malitson")
    assert silica.typ == "sellmeier_offset"
    assert (silica.lambda_min, silica.lambda_max) == (0.21, 6.7)
    assert silica.refractive_index(LINES[0]) == pytest.approx(1.45846,
                                                              abs=2e-5)
    with pytest.raises(ValueError):
        catalog.parse_rii("DATA:\n  - type: tabulated nk\n    data: '1 1 0'\n")
    book = catalog.parse_codev_xml(CODEV_XML)
    assert sorted(book) == ["LAUR", "SILICA"] and book["LAUR"].typ == "schott"
    assert book["SILICA"].refractive_index(LINES[0]) == \
        silica.refractive_index(LINES[0])
    path = tmp_path / "cv.xml"
    path.write_text(CODEV_XML)
    try:
        catalog.load(str(path))
        assert ra.Material.make("cv/LAUR").refractive_index(LINES[0]) == \
            pytest.approx(1.5168, abs=2e-5)
    finally:
        catalog.catalogs.clear()
    if not refshim.available():
        return
    refshim.load()
    import importlib
    import xml.etree.ElementTree as et
    import yaml
    ref_rii = importlib.import_module("rayopt.rii")
    page = yaml.safe_load(RII_PAGE)
    page.update(BOOK="SiO2", PAGE="Malitson")
    theirs = ref_rii.rii_to_material(yaml.dump(page))
    assert theirs.typ == silica.typ
    assert np.array_equal(theirs.coefficients, silica.coefficients)
    ref_codev = importlib.import_module("rayopt.codev")
    for node in et.fromstring(CODEV_XML).iterfind("./Glasses/Glass"):
        theirs = ref_codev.codevxml_to_material(et.tostring(node))
        mine = book[theirs.name]
        assert theirs.typ == mine.typ
        assert np.array_equal(theirs.coefficients, mine.coefficients)
        for l in LINES:
            assert mine.refractive_index(l) == pytest.approx(
                float(theirs.refractive_index(l)), rel=1e-15)
