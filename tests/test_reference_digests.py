"""Full-size traces against the reference itself, by digest.

tests/golden/digests.json holds SHA-256 digests of what the UNMODIFIED
reference computed for the BASELINE configs at sizes it finishes in seconds
(C1 10^4, C2 10^6 rays x 3 wavelengths, C3 10^6 rays in five fields clipped
and unclipped, C4 2*10^4 rays through six Newton-solved aspheres, the
tilted / folded torture system 2*10^5 rays): every value of y, u, i, t of
every traced row (generator: tests/golden/make_digests.py).  Equal digests =
every one of up to 1.2*10^8 values equal to the reference's, bit for bit --
no oracle in between.  Here: the numpy oracle, the C oracle and the host build
of the kernel arithmetic; under -m gpu: the device."""
import json
import os
import sys

import numpy as np
import pytest

import rayopt_amd as ra
from rayopt_amd.pack import pack_system

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import digest_cases as dc  # noqa: E402

with open(os.path.join(os.path.dirname(__file__), "golden",
                       "digests.json")) as f:
    DIGESTS = json.load(f)
CASES = {c["name"]: c for c in dc.cases()}
NAMES = sorted(CASES)
# the asphere case is ~6 s in the numpy oracle (masked vector Newton) and the
# 10^6-ray ones ~3 s each: kept, the whole file stays under a minute on CPU


def rows_from(arrays, L):
    def rows_of(k, j):
        if j >= L:
            raise IndexError
        return arrays[k][j - 1]
    return rows_of


def check_inputs(name):
    case, want = CASES[name], DIGESTS[name]
    y, u = case["rays"]()
    if dc.digest_inputs(y, u) != want["inputs"]:
        pytest.skip("this host builds other launch rays than the recording "
                    "host did (numpy / libm differ): the digest of the "
                    "results cannot be compared")
    return case, want, y, u


@pytest.mark.parametrize("name", NAMES)
def test_inputs_reproduce(name):
    case, want = CASES[name], DIGESTS[name]
    assert dc.digest_inputs(*case["rays"]()) == want["inputs"], \
        "launch rays differ from the recording host's"


@pytest.mark.parametrize("engine", ["numpy oracle", "C oracle",
                                    "kernel arithmetic, host build"])
@pytest.mark.parametrize("name", NAMES)
def test_cpu_restatements_equal_the_reference_by_digest(name, engine,
                                                        hostemu):
    from oracle import trace_numpy as tn, build_c
    if engine != "C oracle" and (CASES[name].get("heavy") or
                                 "unclipped" in name):
        pytest.skip("the 10^7-ray case and the second 10^6-ray double-Gauss "
                    "case: C oracle only (seconds instead of minutes)")
    case, want, y, u = check_inputs(name)
    system = ra.system_from_yaml(case["yaml"])
    table, ns = pack_system(system, case["l"],
                            system.refractive_index(case["l"], 0))
    with np.errstate(all="ignore"):
        if engine == "numpy oracle":
            Y, U, I, T = tn.propagate(table, y, u, clip=case["clip"])
        elif engine == "C oracle":
            Y, U, I, T = build_c.propagate(table, y, u, clip=case["clip"])
        else:
            Y, U, I, T = hostemu(table, y, u, 1, len(table), case["clip"], 1)
    got = dc.digest_rows(rows_from(dict(y=Y, u=U, i=I, t=T), len(system)))
    assert got == want["results"], name
    assert int(np.isnan(U[-1][:, 0]).sum()) == want["dead_at_image"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_device_equals_the_reference_by_digest(name, arith):
    case, want, y, u = check_inputs(name)
    system = ra.system_from_yaml(case["yaml"])
    aspheric = "aspherics" in case["yaml"]
    if arith == "default" and not aspheric:
        pytest.skip("no aspheric element: one arithmetic")
    g = ra.GeometricTrace(system)
    g.rays_given(y, u, case["l"])
    g.propagate(clip=case["clip"])
    arrays = {"y": g.y, "u": g.u, "i": g.i, "t": g.t}
    L = len(system)
    if arith == "default":
        # the shipped arithmetic for aspheres: every value within the 1e-8
        # contract of the trace whose digest IS the reference's (checked by
        # the other parametrisation), identical NaN masks
        from conftest import assert_parity, RTOL_ASPHERE
        exact = ra.GeometricTrace(system, engine=ra.Engine(),
                                  exact_asphere=True)
        exact.rays_given(y, u, case["l"])
        exact.propagate(clip=case["clip"])
        for k in "yuit":
            assert_parity(np.asarray(arrays[k][1:]),
                          np.asarray(getattr(exact, k)[1:]), RTOL_ASPHERE,
                          "%s.%s" % (name, k))
        assert int(np.isnan(np.asarray(g.u[-1])[:, 0]).sum()) == \
            want["dead_at_image"]
        return

    def rows_of(k, j):
        if j >= L:
            raise IndexError
        return np.asarray(arrays[k][j])
    assert dc.digest_rows(rows_of) == want["results"], name
    assert int(np.isnan(np.asarray(g.u[-1])[:, 0]).sum()) == \
        want["dead_at_image"]


@pytest.mark.gpu
def test_three_wavelengths_in_one_launch_equal_the_reference_by_digest():
    """BASELINE config C2 as ONE trace -- rays_given(y, u, l=[l1, l2, l3]):
    three ray groups, each marched through its own surface table -- against
    the digests of the reference's three separate 10^6-ray traces."""
    names = ["C2_cooke_1e6_588nm", "C2_cooke_1e6_656nm", "C2_cooke_1e6_486nm"]
    ls = [587.56e-9, 656.27e-9, 486.13e-9]
    case, want, y, u = check_inputs(names[0])
    # one System whose glasses give exactly the three index sets
    from rayopt_amd import prescriptions as P
    system = ra.system_from_yaml(P.COOKE % dict(
        air="air", sk16="SCHOTT-SK|N-SK16", f2="SCHOTT-F|N-F2"))
    for j, el in enumerate(system):
        for l in ls:
            ref = ra.system_from_yaml(P.cooke(l))
            if el.material is not None:
                assert el.material.refractive_index(l) == \
                    ref[j].material.refractive_index(l)
    g = ra.GeometricTrace(system)
    g.rays_given(y, u, l=ls)
    g.propagate(clip=True)
    n, L = len(y), len(system)
    for k, name in enumerate(names):
        arrays = {"y": g.y, "u": g.u, "i": g.i, "t": g.t}

        def rows_of(key, j):
            if j >= L:
                raise IndexError
            return np.asarray(arrays[key][j])[k*n:(k + 1)*n]
        assert dc.digest_rows(rows_of) == DIGESTS[name]["results"], name


def _generated(make_trace):
    gen, want = dc.GENERATED, DIGESTS[dc.GENERATED["name"]]
    yp = gen["pupil"]()
    if dc.digest_inputs(yp, np.array(gen["fields"])) != want["inputs"]:
        pytest.skip("other pupil points than on the recording host")
    system = ra.system_from_yaml(gen["yaml"])
    g = make_trace(system)
    g.rays_fields(np.array(gen["fields"]), yp, gen["z"], gen["a"], gen["l"])
    L = len(system)
    for what in ("first trace: generation fused into it",
                 "re-trace: launch rays rebuilt, row 0 not read",
                 "re-trace again"):
        g.propagate(clip=gen["clip"])
        arrays = {"y": g.y, "u": g.u, "i": g.i, "t": g.t}

        def rows_of(k, j):
            if j >= L:
                raise IndexError
            return np.asarray(arrays[k][j])
        assert dc.digest_inputs(np.asarray(g.y[0]), np.asarray(g.u[0])) == \
            want["launch"], what
        assert dc.digest_rows(rows_of) == want["results"], what
    return g


def test_generated_bundles_equal_the_reference_by_digest_on_the_double():
    """rays_fields (five fields x 2*10^5 pupil points) + propagate against
    what the reference computes from System.aim + rays_given + propagate:
    launch rays and every traced value, by digest (engine double: the numpy
    oracle's generation and trace)."""
    from fake_engine import OracleEngine
    _generated(lambda system: ra.GeometricTrace(system,
                                                engine=OracleEngine()))


@pytest.mark.gpu
def test_generated_bundles_equal_the_reference_by_digest_on_the_device():
    """The same on the device: rays built by the kernel (first trace), rebuilt
    by every re-trace -- bit for bit the reference's bundles and results."""
    g = _generated(lambda system: ra.GeometricTrace(system))
    assert int(np.isnan(np.asarray(g.u[-1])[:, 0]).sum()) == \
        DIGESTS[dc.GENERATED["name"]]["dead_at_image"]
