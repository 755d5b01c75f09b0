"""The packer's caches (rayopt_amd/pack.py) against every way a caller can
change a System between two propagate() calls -- including in-place edits of
the arrays elements and materials hand out, which no attribute stamp sees.
The reference re-reads e.offset / rot_normal / aspherics / the material on
every call (rayopt/system.py:459-464, elements.py:156-175,283-289)."""
import os
import sys

import numpy as np
import pytest

import rayopt_amd as ra
from rayopt_amd.pack import pack_system
from rayopt_amd.prescriptions import DOUBLE_GAUSS, TORTURE
from oracle import refshim

from fake_engine import OracleEngine

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "tools"))
import fuzz_pack                                            # noqa: E402

needs_reference = pytest.mark.skipif(not refshim.available(),
                                     reason="no reference on this box")


@pytest.mark.parametrize("block", range(4))
def test_cached_table_is_the_table_packed_from_scratch(block):
    """>= 2000 mutation steps over 4 blocks, every step checked."""
    steps = 0
    for seed in range(30*block, 30*(block + 1)):
        with np.errstate(all="ignore"):
            steps += fuzz_pack.sequence(seed, 30)
    assert steps > 500


def _image_row(system, y, u):
    g = ra.GeometricTrace(system, engine=OracleEngine())
    g.rays_given(y, u)
    g.propagate(clip=False)
    return np.array(g.y[-1])


def test_offset_edited_in_place_moves_the_image():
    """The round-2 review's repro: ``system[3].offset[1] += .3`` between two
    propagate() calls must not re-trace the old geometry."""
    s = ra.system_from_yaml(DOUBLE_GAUSS)
    y, u = ra.bundles.disc_bundle(500, 10., 2., 3)
    before = _image_row(s, y, u)
    s[3].offset[1] += .3
    after = _image_row(s, y, u)
    assert not np.array_equal(before, after, equal_nan=True)
    # and the table says what the element says
    l = s.wavelengths[0]
    table, _ = pack_system(s, l, s.refractive_index(l, 0))
    assert np.array_equal(table["offset"][3], s[3].offset)


def test_set_path_into_an_array_is_seen():
    """The advisor's repro: PathVariable's mechanism,
    ``set_path((2, "offset", 0), .5)``."""
    s = ra.system_from_yaml(DOUBLE_GAUSS)
    l = s.wavelengths[0]
    n0 = s.refractive_index(l, 0)
    pack_system(s, l, n0)
    s.set_path((2, "offset", 0), .5)
    table, _ = pack_system(s, l, n0)
    assert table["offset"][2][0] == .5
    assert np.array_equal(table["offset"][2], s[2].offset)


def test_rotation_and_coefficients_edited_in_place_are_seen():
    from rayopt_amd import model
    s = ra.system_from_yaml(TORTURE)
    l = s.wavelengths[0]
    n0 = s.refractive_index(l, 0)
    j = next(k for k, e in enumerate(s) if e.rotated)
    pack_system(s, l, n0)
    s[j].rot_normal[0, 1] += 1e-3
    table, _ = pack_system(s, l, n0)
    assert np.array_equal(table["rot"][j], s[j].rot_normal.ravel())
    gas = model.GasFormula([5792105e-8, 167917e-8], [238.0185, 57.362])
    s[0].material = gas
    n_a = pack_system(s, l, gas.refractive_index(l))[1]
    gas.b[0] *= 1.5
    n_b = pack_system(s, l, gas.refractive_index(l))[1]
    assert n_a[0] != n_b[0]


@needs_reference
def test_in_place_offset_matches_the_reference():
    """Live reference: the same in-place edit on a rayopt.System and on ours
    gives the same image row (rayopt/system.py:461)."""
    ro = refshim.load()
    rs = ro.system_from_yaml(DOUBLE_GAUSS)
    ms = ra.system_from_yaml(DOUBLE_GAUSS)
    y, u = ra.bundles.disc_bundle(300, 10., 2., 5)
    for step in range(3):
        rt = ro.GeometricTrace(rs)
        rt.rays_given(y, u)
        rt.propagate()
        got = _image_row(ms, y, u)
        assert np.array_equal(got, rt.y[-1], equal_nan=True), step
        for sysm in (rs, ms):
            sysm[3].offset[1] += .3
            sysm[5].offset[2] -= .01


def test_tables_of_a_polychromatic_batch_are_the_single_tables():
    """pack_tables (one pass over the elements' notes for all wavelengths, a
    whole-table cache entry per wavelength) returns what pack_system returns
    wavelength by wavelength -- before and after an edit of the system."""
    import rayopt_amd as ra
    from rayopt_amd import prescriptions as P
    from rayopt_amd.pack import pack_system, pack_tables
    s = ra.system_from_yaml(P.COOKE % dict(
        air="air", sk16="SCHOTT-SK|N-SK16", f2="SCHOTT-F|N-F2"))
    ls = [587.56e-9, 656.27e-9, 486.13e-9]
    n0 = [s.refractive_index(l, 0) for l in ls]
    for edit in (None, "curvature", "offset"):
        if edit == "curvature":
            s[2].curvature *= 1.01
        elif edit == "offset":
            s[3].offset[1] += 1e-3
        for _ in range(2):          # second round: every table from its cache
            tables, ns = pack_tables(s, ls, n0, 1, None)
            for k, (l, n) in enumerate(zip(ls, n0)):
                t1, n1 = pack_system(ra.system_from_dict(s.dict()), l, n, 1,
                                     None) if edit is None else \
                    pack_system(s, l, n, 1, None)
                if edit is None:
                    assert tables[k].tobytes() == \
                        pack_system(s, l, n, 1, None)[0].tobytes()
                else:
                    assert tables[k].tobytes() == t1.tobytes()
                    assert np.array_equal(ns[k], n1, equal_nan=True)
    # three different tables (dispersion), one geometry
    assert len({tables[k].tobytes() for k in range(3)}) == 3
