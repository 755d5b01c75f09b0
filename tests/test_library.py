"""The glass library behind Material.make(name) (rayopt_amd/library.py)
against the reference's (rayopt/library.py:50-185, rayopt/material.py:104-115)
on the database that ships with rayopt."""
import os
import sqlite3

import numpy as np
import pytest

import rayopt_amd as ra
from rayopt_amd.library import Library, BUILTIN
from rayopt_amd.model import Material
from rayopt_amd.prescriptions import COOKE_INDICES
from oracle import refshim

LAMBDAS = (486.13e-9, 587.56e-9, 656.27e-9, 1.0e-6)

# the reference's own fixture, rayopt/test/test_raytrace.py:30-57
REFERENCE_COOKE = """
description: 'oslo cooke triplet example 50mm f/4 20deg'
wavelengths: [587.56e-9, 656.27e-9, 486.13e-9]
object: {angle_deg: 20, pupil: {radius: 6.25, aim: True}}
image: {type: finite, pupil: {radius: 0, update_radius: True}}
stop: 5
elements:
- {material: air}
- {roc: 21.25, distance: 5.0, material: SCHOTT-SK|N-SK16, radius: 6.5}
- {roc: -158.65, distance: 2.0, material: air, radius: 6.5}
- {roc: -20.25, distance: 6.0, material: SCHOTT-F|N-F2, radius: 5.0}
- {roc: 19.6, distance: 1.0, material: air, radius: 5.0}
- {material: air, radius: 4.75}
- {roc: 141.25, distance: 6.0, material: SCHOTT-SK|N-SK16, radius: 6.5}
- {roc: -17.285, distance: 2.0, material: air, radius: 6.5}
- {distance: 42.95, radius: 0.364}
"""


@pytest.fixture(autouse=True)
def fresh_library(monkeypatch):
    monkeypatch.delenv("RAYOPT_LIBRARY", raising=False)
    Library.reset()
    yield
    Library.reset()


def builtin_only():
    lib = Library(db=None)
    lib.path = None             # whatever is installed on this box: ignore
    return lib


def test_builtin_table_ships_with_the_package():
    assert os.path.exists(BUILTIN)
    lib = builtin_only()
    names = lib.names("glass")
    assert len(names) > 800 and "SCHOTT-BK|N-BK7" in names
    bk7 = lib.get("material", "schott-bk|n-bk7")          # any case
    assert bk7.refractive_index(587.56e-9) == pytest.approx(1.5168, abs=2e-4)
    with pytest.raises(KeyError):
        lib.get("material", "UNOBTAINIUM")
    with pytest.raises(KeyError):
        lib.get("material", "SCHOTT-BK|N-BK7", catalog="organic")
    assert lib.get("material", "SCHOTT-BK|N-BK7", "glass", "rii") is not None


def test_reference_cooke_fixture_loads_by_name_without_rayopt():
    """rayopt/test/test_raytrace.py:30-57 -- catalogue glass names -- through
    Material.make -> Library -> built-in formulas; the indices are the ones
    read from the reference with its catalogue (prescriptions.COOKE_INDICES,
    printed by tests/golden/make_golden.py)."""
    Library._one = builtin_only()
    s = ra.system_from_yaml(REFERENCE_COOKE)
    assert s[1].material.name == "SCHOTT-SK|N-SK16"
    for l, want in COOKE_INDICES.items():
        assert s[1].material.refractive_index(l) == pytest.approx(
            want["sk16"], rel=1e-15)
        assert s[3].material.refractive_index(l) == pytest.approx(
            want["f2"], rel=1e-15)
    # air is the basic material in both
    assert s[0].material.refractive_index(587.56e-9) == pytest.approx(
        COOKE_INDICES[587.56e-9]["air"], rel=1e-15)


@pytest.mark.skipif(not refshim.available(), reason="no /root/reference")
def test_same_answers_as_the_reference_library():
    """Every 7th material of the reference's database, by name, through both
    libraries: same n(lambda) where the reference can evaluate it."""
    ro = refshim.load()
    path = refshim.library_db()
    db = sqlite3.connect("file:%s?mode=ro" % path, uri=True)
    rows = db.execute("select m.name, c.name from material m join catalog c "
                      "on c.id = m.catalog_id order by m.id").fetchall()[::7]
    mine_db = Library(db=path, builtin=False)
    mine_tsv = builtin_only()
    try:
        from rayopt.library import Library as RefLibrary
        ref = RefLibrary("sqlite:///%s" % path)
    except Exception as err:           # SQLAlchemy missing / too new
        pytest.skip("reference library not usable here: %r" % (err,))
    checked = 0
    for name, cat in rows:
        try:
            with np.errstate(all="ignore"):
                theirs = ref.get("material", name, cat)
                want = [float(theirs.refractive_index(l)) for l in LAMBDAS]
        except Exception:
            continue                   # formulas the reference cannot evaluate
        if not np.isfinite(want).all():
            continue
        if not len(theirs.coefficients):
            # a page with tabulated n, k only: the reference hands out a
            # material without dispersion (n = 1); here that is an error
            with pytest.raises(KeyError):
                mine_db.get("material", name, cat)
            continue
        for lib in (mine_db, mine_tsv):
            got = [lib.get("material", name, cat).refractive_index(l)
                   for l in LAMBDAS]
            np.testing.assert_allclose(got, want, rtol=1e-14, err_msg=name)
        checked += 1
    assert checked > 100


def test_user_database_is_found_and_filled(tmp_path, monkeypatch):
    """$RAYOPT_LIBRARY names the database; Library.load adds catalogue files
    to it (Zemax .agf here) under the reference's schema."""
    agf = tmp_path / "mini.agf"
    agf.write_text(
        "CC test catalogue\n"
        "NM X-BK7 2 517642 1.5168 64.17 0 1\n"
        "CD 1.03961212 0.00600069867 0.231792344 0.0200179144 1.01046945 "
        "103.560653 0 0 0 0\nLD 0.3 2.5\n"
        "NM X-F2 2 620364 1.62004 36.37 0 1\n"
        "CD 1.34533359 0.00997743871 0.209073176 0.0470450767 0.937357162 "
        "111.886764 0 0 0 0\nLD 0.32 2.5\n")
    dbfile = str(tmp_path / "mine.sqlite")
    lib = Library(db=dbfile, builtin=False)
    assert lib.load(str(agf)) == 2
    assert lib.load(str(agf)) == 0                    # unchanged: skipped
    assert lib.load(str(agf), mode="reload") == 2
    tables = {r[0] for r in sqlite3.connect(dbfile).execute(
        "select name from sqlite_master where type = 'table'")}
    assert {"catalog", "material"} <= tables
    g = lib.get("material", "x-bk7", "mini", "zemax")
    assert g.refractive_index(587.56e-9) == pytest.approx(1.5168, abs=1e-4)
    with pytest.raises(KeyError):
        lib.get("material", "X-BK7", "other")
    monkeypatch.setenv("RAYOPT_LIBRARY", dbfile)
    Library.reset()
    assert Library.find_db() == dbfile
    assert Material.make("mini/X-F2").refractive_index(587.56e-9) == \
        pytest.approx(1.62004, abs=1e-4)
    assert Material.make("zemax/mini/X-F2").refractive_index(5e-7) == \
        Material.make("mini/X-F2").refractive_index(5e-7)
    assert Material.make("mini/X-F2") is Material.make("mini/X-F2")  # cached
