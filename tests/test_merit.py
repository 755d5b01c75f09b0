"""The optimiser front end (rayopt_amd/merit.py) against the reference's
rayopt/optimize.py, and SpotOperand (one batched trace + one grouped device
reduction per merit evaluation)."""
import numpy as np
import pytest

import rayopt_amd as ra
from rayopt_amd import merit
from rayopt_amd.prescriptions import SINGLET, COOKE
from oracle import refshim

DISPERSIVE_COOKE = COOKE % dict(air=1.0, sk16="1.62041/60.32",
                                f2="1.62004/36.37")


def oracle_trace(system):
    from fake_engine import OracleEngine
    return ra.GeometricTrace(system, engine=OracleEngine())


def bundle(n=400):
    return ra.bundles.disc_bundle(n, 7.5, 0., 0)


def test_paths():
    s = ra.system_from_yaml(SINGLET)
    assert s.get_path((1, "curvature")) == s[1].curvature
    s.set_path((2, "curvature"), -0.01)
    assert s[2].curvature == -0.01
    s.set_path(("wavelengths", 0), 500e-9)
    assert s.wavelengths[0] == 500e-9
    assert s.get_path(("object", "finite")) is False or \
        s.get_path(("object", "finite")) == 0


def test_operand_terms():
    s = ra.system_from_yaml(SINGLET)
    v = np.array([1., 3.])
    op = merit.FuncOp(s, lambda sys: v, weight=2., offset=1.)
    (f,) = op.get_objective()
    assert np.array_equal(f(op.get()), [0., 4.])
    assert not list(op.get_equality()) and not list(op.get_inequality())
    op = merit.FuncOp(s, lambda sys: v, min=0.5, max=2.5, offset=.5)
    lo, hi = op.get_inequality()
    assert np.array_equal(lo(v), [0., 2.]) and np.array_equal(hi(v), [2., 0.])
    assert not list(op.get_objective())
    op = merit.FuncOp(s, lambda sys: v, min=1., max=1.)
    (eq,) = op.get_equality()
    assert np.array_equal(eq(v), v)          # the reference's v - offset
    with pytest.raises(AssertionError):
        merit.PathVariable(s, (1, "curvature"))      # unbounded, no scale
    var = merit.PathVariable(s, (1, "curvature"), bounds=(0., .1))
    assert var.scale == pytest.approx(.1) and var.init == s[1].curvature


def make_problem(mod, system, trace_of, constrained):
    y, u = bundle()

    def spot(sys):
        sys.update()
        t = trace_of(sys)
        t.rays_given(y, u)
        t.propagate()
        return t.rms()

    variables = [
        mod.PathVariable(system, (1, "curvature"), bounds=(0.005, 0.05)),
        mod.PathVariable(system, (2, "curvature"), bounds=(-0.05, -0.002)),
    ]
    operands = [mod.FuncOp(system, spot, weight=1.)]
    if constrained:     # keep the power: c1 - c2 >= 0.03, and c1 <= 0.03
        operands.append(mod.FuncOp(
            system, lambda sys: sys[1].curvature - sys[2].curvature,
            min=0.03))
        operands.append(mod.FuncOp(system, lambda sys: sys[1].curvature,
                                   max=0.03))
    return variables, operands


@pytest.mark.skipif(not refshim.available(), reason="no /root/reference")
@pytest.mark.parametrize("constrained", [False, True])
def test_optimize_takes_the_reference_iterates(constrained):
    """Same variables and operands through rayopt.optimize.optimize on the
    reference's objects and through merit.optimize on ours (oracle-backed
    engine double): same iterates, optimum, result attributes."""
    ro = refshim.load()
    import importlib
    ref_opt = importlib.import_module("rayopt.optimize")
    theirs = ro.system_from_yaml(SINGLET)
    theirs.update()
    mine = ra.system_from_yaml(SINGLET)
    rv, rp = make_problem(ref_opt, theirs, ro.GeometricTrace, constrained)
    mv, mp = make_problem(merit, mine, oracle_trace, constrained)
    f0 = mp[0].get()[0]**2
    r = ref_opt.optimize(rv, rp, trace=True)
    m = merit.optimize(mv, mp, trace=True)
    assert m.success == r.success and m.nit == r.nit and m.nfev == r.nfev
    np.testing.assert_allclose(m.x, r.x, rtol=1e-9)
    assert m.fun == pytest.approx(r.fun, rel=1e-9)
    assert m.fun < f0 or constrained
    np.testing.assert_allclose(m.trace_x, r.trace_x, rtol=1e-9)
    assert len(m.trace_v) == len(r.trace_v) == len(m.trace_x)
    assert [k for k, _ in m.trace_f] == [k for k, _ in r.trace_f]
    np.testing.assert_allclose(m.trace_f[0][1], r.trace_f[0][1], rtol=1e-8)
    # accept / reject write the optimum / the starting values back
    start = [theirs[1].curvature, theirs[2].curvature]
    m.accept()
    r.accept()
    assert mine[1].curvature == pytest.approx(theirs[1].curvature, rel=1e-9)
    if constrained:
        assert mine[1].curvature - mine[2].curvature >= 0.03 - 1e-9
        assert mine[1].curvature <= 0.03 + 1e-9
    m.reject()
    assert mine[1].curvature == pytest.approx(
        ra.system_from_yaml(SINGLET)[1].curvature)
    del start


@pytest.mark.skipif(not refshim.available(), reason="no /root/reference")
@pytest.mark.parametrize("constrained", [False, True])
def test_variables_and_operands_are_interchangeable(constrained):
    """The contract between optimize() and its arguments is structural: the
    reference's own Variable/Operand objects run through merit.optimize, and
    merit's through rayopt.optimize.optimize -- same optimum either way."""
    ro = refshim.load()
    import importlib
    ref_opt = importlib.import_module("rayopt.optimize")
    theirs = ro.system_from_yaml(SINGLET)
    theirs.update()
    rv, rp = make_problem(ref_opt, theirs, ro.GeometricTrace, constrained)
    want = ref_opt.optimize(rv, rp)
    # reference objects, our optimiser
    theirs2 = ro.system_from_yaml(SINGLET)
    theirs2.update()
    rv2, rp2 = make_problem(ref_opt, theirs2, ro.GeometricTrace, constrained)
    got = merit.optimize(rv2, rp2)
    np.testing.assert_allclose(got.x, want.x, rtol=1e-9)
    assert got.nit == want.nit
    # our objects, the reference's optimiser
    mine = ra.system_from_yaml(SINGLET)
    mv, mp = make_problem(merit, mine, oracle_trace, constrained)
    back = ref_opt.optimize(mv, mp)
    np.testing.assert_allclose(back.x, want.x, rtol=1e-9)
    assert back.nit == want.nit


def test_variable_from_callables():
    box = {"v": 2.}
    var = merit.Variable(None, bounds=(0., 4.), getter=lambda: box["v"],
                         setter=lambda x: box.update(v=x))
    assert var.init == 2. and var.scale == 4.
    var.set(3.)
    assert var.get() == 3.
    with pytest.raises(NotImplementedError):
        merit.Variable(None, scale=1.)


def test_each_point_is_traced_once():
    """merit, constraints and callback at the same x share one evaluation."""
    system = ra.system_from_yaml(SINGLET)
    calls = []
    variables, operands = make_problem(merit, system, oracle_trace, True)
    inner = operands[0].func
    operands[0].func = lambda sys: (calls.append(1), inner(sys))[1]
    res = merit.optimize(variables, operands, trace=True,
                         options=dict(maxiter=5))
    assert len(calls) == res.nevaluations
    assert res.nevaluations <= res.nfev + res.nit + 3*res.nit + 3


def spot_operand(system, trace=None, **kw):
    fields = np.c_[np.zeros(3), [0., .7, 1.]]
    return merit.SpotOperand(system, fields, nrays=10,
                             distribution="hexapolar", clip=False,
                             weight=1., trace=trace, **kw)


def test_spot_operand_host_logic():
    system = ra.system_from_yaml(DISPERSIVE_COOKE)
    op = spot_operand(system, oracle_trace(system))
    v = op.get()
    assert v.shape == (9,) and np.isfinite(v).all()
    # one wavelength: the scalar path, one bundle each
    single = spot_operand(system, oracle_trace(system),
                          wavelengths=[system.wavelengths[1]])
    np.testing.assert_allclose(single.get(), v[3:6], rtol=1e-12)
    # agrees with one reference-style rays_point + rms per bundle
    t = oracle_trace(system)
    import rayopt_amd.aiming as aiming
    real = aiming.GeometricTrace
    for w, l in enumerate(system.wavelengths):
        for f, field in enumerate(op.fields):
            t.rays_point(field, wavelength=l, nrays=10,
                         distribution="hexapolar", filter=False)
            assert t.rms() == pytest.approx(v[3*w + f], rel=1e-9)
    assert aiming.GeometricTrace is real


@pytest.mark.gpu
def test_spot_operand_optimisation_gpu():
    """Refocus + bend the rear element of the triplet for the polychromatic
    spot over three fields: every merit evaluation is one trace of
    3 wavelengths x 3 fields that keeps only the image row."""
    system = ra.system_from_yaml(DISPERSIVE_COOKE)
    system[-1].distance += 0.3                       # defocus the start
    op = merit.SpotOperand(system, np.c_[np.zeros(3), [0., .7, 1.]],
                           nrays=600, distribution="hexapolar", clip=False,
                           weight=1.)
    before = op.get()
    assert before.shape == (9,) and np.isfinite(before).all()
    variables = [
        merit.PathVariable(system, (-1, "distance"), bounds=(40., 46.)),
        merit.PathVariable(system, (7, "curvature"), bounds=(-0.08, -0.04)),
    ]
    res = merit.optimize(variables, [op], options=dict(maxiter=30))
    res.accept()
    after = op.get()
    assert np.square(after).sum() < 0.9*np.square(before).sum()
    assert np.square(after).sum() == pytest.approx(res.fun, rel=1e-6)
    assert len(op.kernel_ms) >= res.nevaluations and max(op.kernel_ms) < 50.
    # the operand equals the statistics of the downloaded image row
    t = op.trace
    t.rays_points(op.fields, wavelength=system.wavelengths, nrays=600,
                  distribution="hexapolar")
    P, A = t.rays_per_field, t.rays_alive_per_field
    spots = np.asarray(t.y[-1])[:, :2].reshape(3, 3, P, 2)[:, :, :A]
    want = np.sqrt(np.square(spots - spots.mean(2, keepdims=True))
                   .sum(3).mean(2)).ravel()
    np.testing.assert_allclose(after, want, rtol=1e-10)


@pytest.mark.gpu
def test_polychromatic_example_runs():
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), "examples", "optimize_polychromatic.py")
    spec = importlib.util.spec_from_file_location("optimize_polychromatic",
                                                  path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    before, after, res = mod.main(nrays=300, verbose=False)
    assert np.square(after).sum() < np.square(before).sum()
    assert np.isfinite(after).all()


def _batched_problem(system, trace):
    op = merit.SpotOperand(system, np.c_[np.zeros(3), [0., .7, 1.]],
                           nrays=10, distribution="hexapolar", clip=False,
                           weight=1., trace=trace)
    variables = [
        merit.PathVariable(system, (-1, "distance"), bounds=(40., 46.)),
        merit.PathVariable(system, (6, "curvature"), bounds=(0.002, 0.012)),
        merit.PathVariable(system, (7, "curvature"), bounds=(-0.08, -0.04)),
    ]
    return variables, op


def _batched_gradient_checks(make_trace):
    """get_variants == get() per variant; the one-launch gradient equals
    forward differences from separate evaluations; optimize(jac='batched')
    reaches the optimum of the default (scipy finite differences) run with a
    fraction of the operand evaluations."""
    import copy
    system = ra.system_from_yaml(DISPERSIVE_COOKE)
    system[-1].distance += 0.3
    variables, op = _batched_problem(system, make_trace(system))
    variants = []
    for k in range(3):
        s = copy.deepcopy(system)
        s[7].curvature *= 1 + 1e-3*k
        variants.append(s)
    many = op.get_variants(variants)
    assert many.shape == (3, 9)
    for k, s in enumerate(variants):
        one = merit.SpotOperand(s, op.fields, nrays=10, clip=False,
                                distribution="hexapolar", weight=1.,
                                trace=make_trace(s)).get()
        np.testing.assert_allclose(many[k], one, rtol=1e-12)
    problem = merit._Problem(variables, [op])
    assert problem.batchable()
    x = problem.start.copy()
    g = problem.gradient(x, 1e-5)
    f0 = problem.merit(x)
    assert problem.merit_batched(x) == pytest.approx(f0, rel=1e-12)
    for k in range(3):
        xk = x.copy()
        xk[k] += 1e-5
        assert g[k] == pytest.approx((problem.merit(xk) - f0)/1e-5,
                                     rel=1e-6, abs=1e-9)
    problem.apply(problem.current)
    slow = merit.optimize(variables, [op], options=dict(maxiter=30))
    slow.reject()
    fast = merit.optimize(variables, [op], options=dict(maxiter=30),
                          jac="batched")
    assert fast.fun == pytest.approx(slow.fun, rel=1e-3)
    assert fast.nevaluations < 0.6*slow.nevaluations
    # constraints or foreign operands: refused, not silently ignored
    with pytest.raises(ValueError):
        merit.optimize(variables, [op, merit.FuncOp(
            system, lambda s: s[7].curvature, max=0.)], jac="batched")


def test_batched_gradient_host_logic():
    _batched_gradient_checks(oracle_trace)


@pytest.mark.gpu
def test_batched_gradient_gpu():
    _batched_gradient_checks(lambda s: ra.GeometricTrace(s))


def test_path_variable_methods_resolve_at_call_time():
    """Round-2 advisor finding: get/set of a PathVariable are methods, not
    closures over the constructor's arguments -- a subclass that overrides
    them is honoured, and a variable deep-copied with its system (or whose
    system is rebound) reads and writes THAT system
    (rayopt/optimize.py:46-56 resolves self.system at call time)."""
    import copy
    from rayopt_amd.merit import PathVariable
    s = ra.system_from_yaml(ra.prescriptions.SINGLET)

    class Clamped(PathVariable):
        def set(self, value):
            PathVariable.set(self, min(value, .05))

    v = Clamped(s, (1, "curvature"), bounds=(-1, 1))
    v.set(.5)
    assert s[1].curvature == .05 and v.get() == .05
    s2, v2 = copy.deepcopy((s, v))
    v2.set(.01)
    assert s2[1].curvature == .01 and s[1].curvature == .05
    other = ra.system_from_yaml(ra.prescriptions.SINGLET)
    v.system = other
    v.set(.02)
    assert other[1].curvature == .02 and s[1].curvature == .05
