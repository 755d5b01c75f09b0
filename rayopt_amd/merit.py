"""The optimiser as a caller of the fast trace.

Same public names, argument meaning and result object as the reference's
``rayopt/optimize.py`` (``Variable``, ``PathVariable``, ``Operand``,
``FuncOp``, ``optimize``; rayopt/optimize.py:24-161), so merit functions
written for rayopt run unchanged; the difference is what an operand costs.
:class:`SpotOperand` evaluates the RMS spot of every field at every
wavelength with ONE batched trace that stores only the image row
(``propagate(keep=[-1])``) and ONE device reduction (``rt_spot_stats``):
a few milliseconds and ``6 x fields x wavelengths`` doubles over PCIe per
merit evaluation, where the reference traces bundle after bundle on the host.

The minimiser itself is scipy's, driven exactly as the reference drives it
(normalised variables ``x = value/scale``, sum of squared weighted operands,
``eq``/``ineq`` constraint dictionaries, ``maxiter=100, eps=1e-5`` unless
overridden), so with the same operands both take the same iterates.
"""
import numpy as np
from scipy.optimize import minimize


class Variable:
    """A degree of freedom: ``get()`` / ``set(value)``, box ``bounds`` and a
    ``scale`` that normalises it for the minimiser (default: the width of the
    box, which then has to be finite; rayopt/optimize.py:24-43)."""
    def __init__(self, system, bounds=(-np.inf, np.inf), scale=None,
                 init=None):
        self.system = system
        self.bounds = bounds
        if scale is None:
            scale = bounds[1] - bounds[0]
            assert np.isfinite(scale), "give a scale or finite bounds"
        self.scale = scale
        self.init = self.get() if init is None else init

    def get(self):
        raise NotImplementedError

    def set(self, value):
        raise NotImplementedError


class PathVariable(Variable):
    """The attribute / item reached by ``system.get_path(path)``, e.g.
    ``(1, "curvature")`` (rayopt/optimize.py:46-55)."""
    def __init__(self, system, path, *args, **kwargs):
        self.path = path
        super().__init__(system, *args, **kwargs)

    def get(self):
        return self.system.get_path(self.path)

    def set(self, value):
        self.system.set_path(self.path, value)


class Operand:
    """A vector-valued quantity ``get()`` of the system and how it enters the
    problem (rayopt/optimize.py:58-84): with a ``weight`` it adds
    ``sum((weight*(v - offset))**2)`` to the merit; ``min`` / ``max`` bound
    ``v - offset`` from below / above; ``min == max`` makes it an equality
    (as in the reference the equality is ``v - offset == 0`` whatever the
    common value of ``min`` and ``max`` is)."""
    def __init__(self, system, weight=None, offset=0, min=None, max=None):
        self.system = system
        self.weight = weight
        self.offset = offset
        self.min = min
        self.max = max

    def get(self):
        raise NotImplementedError

    def get_objective(self):
        if self.weight:
            yield lambda v: self.weight*(v - self.offset)

    def get_equality(self):
        if self.min is not None and self.min == self.max:
            yield lambda v: v - self.offset

    def get_inequality(self):
        if self.min is not None:
            yield lambda v: v - self.offset - self.min
        if self.max is not None:
            yield lambda v: self.max - (v - self.offset)


class FuncOp(Operand):
    """``func(system)`` flattened (rayopt/optimize.py:87-93)."""
    def __init__(self, system, func, *args, **kwargs):
        super().__init__(system, *args, **kwargs)
        self.func = func

    def get(self):
        return np.atleast_1d(self.func(self.system)).ravel()


class SpotOperand(Operand):
    """RMS spot radius of every field (rows of ``fields``, fractional object
    coordinates) at every wavelength, flattened wavelength-major.

    One evaluation = aiming of all fields (batched on the GPU, optional) +
    one fused trace of ``wavelengths x fields x bundle`` rays that keeps only
    the image row + one grouped device reduction.  ``lost="omit"`` (default)
    measures the rays that arrive, ``"nan"`` gives NaN for a bundle that
    lost a ray like the reference's ``rms()``; non-finite values are
    replaced by ``penalty`` so a minimiser can step out of a bad region."""
    def __init__(self, system, fields, wavelengths=None, nrays=200,
                 distribution="hexapolar", clip=True, aim=True, lost="omit",
                 penalty=1e3, trace=None, **kwargs):
        super().__init__(system, **kwargs)
        from .geometric_trace import GeometricTrace
        self.fields = np.atleast_2d(np.asarray(fields, dtype=float))
        self.wavelengths = wavelengths
        self.nrays = nrays
        self.distribution = distribution
        self.clip = clip
        self.aim = aim
        self.lost = lost
        self.penalty = penalty
        self.trace = GeometricTrace(system) if trace is None else trace
        self.kernel_ms = []

    def get(self):
        l = self.wavelengths
        if l is None:
            l = self.system.wavelengths
        if np.ndim(l) == 1 and len(l) == 1:
            l = l[0]
        t = self.trace
        t.rays_points(self.fields, wavelength=l, nrays=self.nrays,
                      distribution=self.distribution, clip=self.clip,
                      aim=self.aim, keep=[-1])
        self.kernel_ms.append(t.kernel_ms())
        r = t.rms_fields(lost=self.lost).ravel()
        return np.where(np.isfinite(r), r, self.penalty)


class _Problem:
    """Variables + operands -> the callables scipy needs.  The operand
    vectors of the most recent points are kept, so the merit, the constraint
    functions and the callback evaluated at the same ``x`` trace once."""
    def __init__(self, variables, operands):
        assert variables
        assert operands
        self.variables = variables
        self.operands = operands
        self.scale = np.array([v.scale for v in variables])
        self.current = np.array([v.get() for v in variables])
        self.start = np.array([v.init for v in variables])/self.scale
        self.bounds = np.array([v.bounds for v in variables]) \
            / self.scale[:, None]
        self.objective, self.equality, self.inequality = [], [], []
        for k, op in enumerate(operands):
            self.objective += [(k, f) for f in op.get_objective()]
            self.equality += [(k, f) for f in op.get_equality()]
            self.inequality += [(k, f) for f in op.get_inequality()]
        assert self.objective, "no operand has a weight"
        self._seen = {}
        self._room = len(variables) + 1
        self.evaluations = 0

    def apply(self, values):
        for value, var in zip(values, self.variables):
            var.set(value)

    def values(self, x):
        key = tuple(np.asarray(x, dtype=float))
        hit = self._seen.pop(key, None)
        if hit is None:
            self.apply(np.asarray(x)*self.scale)
            hit = [op.get() for op in self.operands]
            self.evaluations += 1
        self._seen[key] = hit                # most recently used last
        while len(self._seen) > self._room:
            self._seen.pop(next(iter(self._seen)))
        return hit

    def _stack(self, terms, x):
        v = self.values(x)
        return np.concatenate([f(v[k]) for k, f in terms])

    def merit(self, x):
        return np.square(self._stack(self.objective, x)).sum()

    def constraints(self):
        out = []
        if self.equality:
            out.append({"type": "eq",
                        "fun": lambda x: self._stack(self.equality, x)})
        if self.inequality:
            out.append({"type": "ineq",
                        "fun": lambda x: self._stack(self.inequality, x)})
        return out


def optimize(variables, operands, callback=None, tol=1e-4, options={},
             trace=False, **kwargs):
    """Minimise the summed squares of the weighted operands over the
    variables, subject to the operands' bounds (rayopt/optimize.py:96-161).

    Returns scipy's ``OptimizeResult`` with, as in the reference,
    ``accept()`` / ``reject()`` (write the optimum / the starting values into
    the system) and, with ``trace=True``, ``trace_x`` (variables per
    iteration, unscaled), ``trace_v`` (operand vectors) and ``trace_f``
    (``(operand index, weighted terms per iteration)``).  Extra keywords go
    to ``scipy.optimize.minimize``.  ``nevaluations`` counts the distinct
    points the operands were evaluated at."""
    problem = _Problem(variables, operands)
    path_x, path_v, path_f = [], [], []

    def each_iteration(x):
        if trace:
            v = problem.values(x)
            path_x.append(x*problem.scale)
            path_v.append(v)
            path_f.append([f(v[k]) for k, f in problem.objective])
        if callback:
            return callback(x)

    settings = dict(maxiter=100, eps=1e-5)
    settings.update(options)
    result = minimize(problem.merit, problem.start, bounds=problem.bounds,
                      constraints=problem.constraints(),
                      callback=each_iteration, tol=tol, options=settings,
                      **kwargs)
    result.accept = lambda: problem.apply(result.x*problem.scale)
    result.reject = lambda: problem.apply(problem.current)
    result.trace_x = np.array(path_x)
    result.trace_v = path_v
    result.trace_f = [(k, np.array([step[j] for step in path_f]))
                      for j, (k, f) in enumerate(problem.objective)]
    result.nevaluations = problem.evaluations
    return result
