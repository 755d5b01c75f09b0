import ctypes
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

# Even aspheres run on one of two arithmetics (rt_set_option
# "exact_asphere"): the shipped default -- FMA / rcp / rsq Newton solve, 1e-8
# contract, identical NaN masks -- and the bit-for-bit restatement of scipy's
# iteration.  Tests that compare aspheric results with the reference take the
# ``arith`` argument and run in BOTH ("default": contract tolerance + masks,
# "exact": ==); every other test runs with the exact one, so that "bit
# identical" means the reference's bits everywhere (the fixture below sets it
# per test; nothing is forced process-wide).
ARITHMETICS = ("exact", "default")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
if GOLDEN not in sys.path:      # digest_cases: the seeded bundles of the
    sys.path.insert(1, GOLDEN)  # full-size cases, shared with bench_legs.py

# tolerances of the parity contract (BASELINE.json north_star)
RTOL_SPHERICAL = 1e-10   # plane / sphere / conic closed form
RTOL_ASPHERE = 1e-8      # iterated even aspheres


def pytest_configure(config):
    config.addinivalue_line(
        "markers", "gpu: needs a real MI355X (run by gpurun / the driver)")


def pytest_generate_tests(metafunc):
    if "arith" in metafunc.fixturenames:
        metafunc.parametrize("arith", ARITHMETICS)


@pytest.fixture(autouse=True)
def _asphere_arithmetic(request, monkeypatch):
    """Contexts created during the test start on the test's arithmetic
    (rt_create reads RT_MI355_EXACT_ASPHERE) and the process-wide engines are
    switched to it."""
    params = getattr(getattr(request.node, "callspec", None), "params", {})
    exact = params.get("arith", "exact") == "exact"
    monkeypatch.setenv("RT_MI355_EXACT_ASPHERE", "1" if exact else "0")
    try:
        from rayopt_amd import engine as _engine
        for eng in list(_engine._engines.values()):
            if getattr(eng, "ctx", None):
                eng.set_option("exact_asphere", 1 if exact else 0)
    except Exception:
        pass
    yield


@pytest.fixture
def arith(request):
    return request.param if hasattr(request, "param") else \
        request.node.callspec.params["arith"]


def same_or_contract(got, want, aspheric, arith, what=""):
    """The comparison of a device result with the reference's / the oracle's:
    bit for bit, except aspheric systems on the default arithmetic -- those
    to the 1e-8 contract with identical NaN masks (assert_parity)."""
    if aspheric and arith == "default":
        assert_parity(np.asarray(got), np.asarray(want), RTOL_ASPHERE, what)
    else:
        assert np.array_equal(np.asarray(got), np.asarray(want),
                              equal_nan=True), what


def golden_names():
    return sorted(os.path.basename(p)[:-4]
                  for p in glob.glob(os.path.join(GOLDEN, "*.npz"))
                  if not os.path.basename(p).startswith(
                      ("kat_", "aim_", "consumers_", "analysis_",
                       "adversarial_")))


def consumer_golden_names():
    return sorted(os.path.basename(p)[:-4]
                  for p in glob.glob(os.path.join(GOLDEN, "consumers_*.npz")))


def load_consumer_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        g = {k: z[k] for k in z.files}
    g["yaml"] = str(g["yaml"])
    for k in ("l", "radius", "rms_mean", "rms_ref", "rms_mid",
              "refocus_shift"):
        g[k] = float(g[k])
    g["ref"] = int(g["ref"])
    g["clip"] = bool(g["clip"])
    g["finite_object"] = bool(g["finite_object"])
    g["w"] = g["w"] if g["w"].size else None
    return g


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        g = {k: z[k] for k in z.files}
    g["yaml"] = str(g["yaml"])
    g["l"] = float(g["l"])
    g["clip"] = bool(g["clip"])
    g["start"] = int(g["start"])
    g["stop"] = None if int(g["stop"]) == -999 else int(g["stop"])
    return g


def blas_follows_fma_chain():
    """Does THIS host's np.dot of an (N,3) array with a 3x3 matrix sum the
    inner index in order with fused multiply-adds, as the kernel does
    (rt_math.h: rt_dot3)?  Checked against exact rational arithmetic.  True
    for numpy's OpenBLAS on x86-64 with FMA3 (where the goldens were made);
    where it is, the numpy oracle -- and the reference -- are matched bit for
    bit on tilted elements too, elsewhere to ~1e-14."""
    from fractions import Fraction as F
    rng = np.random.default_rng(11)
    r = np.linalg.qr(rng.normal(size=(3, 3)))[0]
    y = rng.normal(size=(37, 3))*10
    for m in (r, r.T):
        got = np.dot(y, m)
        for i in range(len(y)):
            for j in range(3):
                acc = float(F(y[i, 0])*F(m[0, j]))
                acc = float(F(acc) + F(y[i, 1])*F(m[1, j]))
                acc = float(F(acc) + F(y[i, 2])*F(m[2, j]))
                if acc != got[i, j]:
                    return False
    return True


def case_rtol(g):
    return RTOL_ASPHERE if "aspherics" in g["yaml"] else RTOL_SPHERICAL


def assert_parity(got, want, rtol, what=""):
    """NaN masks must be identical; finite entries must agree to ``rtol``
    relative, with an absolute floor of ``rtol`` x the largest finite
    magnitude of the same surface row (components that are ~0 by symmetry
    have no meaningful relative error)."""
    got = np.asarray(got)
    want = np.asarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    for j in range(want.shape[0]):
        a, b = got[j], want[j]
        nan_a, nan_b = np.isnan(a), np.isnan(b)
        assert np.array_equal(nan_a, nan_b), \
            "%s row %d: NaN mask differs at %d entries" % (
                what, j, (nan_a != nan_b).sum())
        inf_b = np.isinf(b)
        assert np.array_equal(a[inf_b], b[inf_b]), "%s row %d: inf" % (what, j)
        fin = np.isfinite(b)
        if not fin.any():
            continue
        scale = np.abs(b[fin]).max()
        err = np.abs(a[fin] - b[fin])
        tol = rtol*np.maximum(np.abs(b[fin]), scale)
        worst = (err/np.maximum(np.abs(b[fin]), scale)).max()
        assert (err <= tol).all(), "%s row %d: rel err %.3g > %.1g" % (
            what, j, worst, rtol)


def build_hostemu():
    """g++ build of the kernel's arithmetic header (tests/hostemu), rebuilt
    when the sources are newer; returns the library path."""
    src = os.path.join(ROOT, "tests", "hostemu", "hostemu.cpp")
    lib = os.path.join(ROOT, "tests", "hostemu", "libhostemu.so")
    hdrs = [os.path.join(ROOT, "rayopt_amd", "csrc", "rt_math.h"),
            os.path.join(ROOT, "rayopt_amd", "csrc", "rt_aim.h"),
            os.path.join(ROOT, "include", "rt_mi355.h")]
    if (not os.path.exists(lib) or
            os.path.getmtime(lib) < max(os.path.getmtime(f)
                                        for f in [src] + hdrs)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off",
                               "-fPIC", "-shared", "-o", lib, src])
    return lib


@pytest.fixture(scope="session")
def hostemu():
    """The kernel arithmetic compiled for the host, as callables."""
    dll = ctypes.CDLL(build_hostemu())
    dll.emu_trace.restype = ctypes.c_int

    def trace(table, y0, u0, start, stop, clip, rays_per_lane=2):
        y0 = np.ascontiguousarray(y0, dtype=float)
        u0 = np.ascontiguousarray(u0, dtype=float)
        table = np.ascontiguousarray(table)
        n = y0.shape[0]
        rows = stop - start
        Y = np.full((rows, n, 3), -7.)
        U = np.full((rows, n, 3), -7.)
        I = np.full((rows, n, 3), -7.)
        T = np.full((rows, n), -7.)
        rc = dll.emu_trace(
            ctypes.c_void_p(table.ctypes.data), start, stop, int(clip),
            rays_per_lane, ctypes.c_void_p(y0.ctypes.data),
            ctypes.c_void_p(u0.ctypes.data), ctypes.c_int64(n),
            ctypes.c_void_p(Y.ctypes.data), ctypes.c_void_p(U.ctypes.data),
            ctypes.c_void_p(I.ctypes.data), ctypes.c_void_p(T.ctypes.data))
        assert rc == 0
        return Y, U, I, T

    def generate(fields, pupil, s0):
        fields = np.ascontiguousarray(fields)
        pupil = np.ascontiguousarray(pupil, dtype=float)
        s0 = np.ascontiguousarray(s0)
        n = len(fields)*len(pupil)
        Y, U = np.empty((n, 3)), np.empty((n, 3))
        rc = dll.emu_generate(
            ctypes.c_void_p(fields.ctypes.data), len(fields),
            ctypes.c_void_p(pupil.ctypes.data), ctypes.c_int64(len(pupil)),
            ctypes.c_void_p(s0.ctypes.data), ctypes.c_void_p(Y.ctypes.data),
            ctypes.c_void_p(U.ctypes.data))
        assert rc == 0
        return Y, U
    trace.generate = generate
    return trace
