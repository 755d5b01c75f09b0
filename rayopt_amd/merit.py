"""The optimiser as a caller of the fast trace.

Same public names, argument meaning and result object as the reference's
``rayopt/optimize.py`` (``Variable``, ``PathVariable``, ``Operand``,
``FuncOp``, ``optimize``; rayopt/optimize.py:24-161), so merit functions
written for rayopt run unchanged; the difference is what an operand costs.
:class:`SpotOperand` evaluates the RMS spot of every field at every
wavelength with ONE batched trace that stores only the image row
(``propagate(keep=[-1])``) and ONE device reduction (``rt_row_stats``, one
pass over the image row):
a few milliseconds and ``10 x fields x wavelengths`` doubles over PCIe per
merit evaluation, where the reference traces bundle after bundle on the host.

The minimiser itself is scipy's, driven exactly as the reference drives it
(normalised variables ``x = value/scale``, sum of squared weighted operands,
``eq``/``ineq`` constraint dictionaries, ``maxiter=100, eps=1e-5`` unless
overridden), so with the same operands both take the same iterates.
"""
import numpy as np
from scipy.optimize import minimize


# --------------------------------------------------------------------------
# variables and operands
#
# The contract with optimize() is structural, so that objects written for the
# reference (rayopt/optimize.py:24-93) can be handed to this optimize() and
# the classes below to the reference's: a *variable* is anything with
# ``get()``, ``set(value)``, ``bounds``, ``scale`` and ``init``; an *operand*
# anything with ``get()`` and the three iterables ``get_objective()``,
# ``get_equality()``, ``get_inequality()`` of callables that map the operand
# vector to residuals (objective terms are squared and summed; equalities are
# driven to zero, inequalities kept >= 0).
# --------------------------------------------------------------------------

class Affine:
    """The residual map ``v -> gain*(v - offset) - shift`` of an operand
    vector.  Every role an operand can play is one of these: a weighted
    objective term (gain = weight), an equality (1, shift 0), a lower bound
    ``v - offset >= lo`` (1, shift lo), an upper bound ``v - offset <= hi``
    (-1, shift -hi)."""
    __slots__ = ("gain", "offset", "shift")

    def __init__(self, gain, offset, shift=0):
        self.gain, self.offset, self.shift = gain, offset, shift

    def __call__(self, v):
        return self.gain*(v - self.offset) - self.shift

    def __repr__(self):
        return "Affine(%r*(v - %r) - %r)" % (self.gain, self.offset,
                                             self.shift)


def _abstract(what):
    def method(self, *args):
        raise NotImplementedError("%s.%s: subclass %s, or pass the callable"
                                  % (type(self).__name__, what,
                                     type(self).__name__))
    method.__name__ = what
    return method


class _Described:
    """repr with the public attributes: what optimisation logs show."""
    def __repr__(self):
        public = ", ".join(
            "%s=%r" % (k, v) for k, v in sorted(vars(self).items())
            if k != "system" and not k.startswith("_") and not callable(v))
        return "%s(%s)" % (type(self).__name__, public)


class Variable(_Described):
    """A degree of freedom of ``system`` (same constructor as
    rayopt/optimize.py:24): box ``bounds``, a ``scale`` that normalises it
    for the minimiser -- by default the width of the box, which then has to
    be finite -- and the value ``init`` the search starts from (default: the
    present one).  Either subclass it with ``get``/``set`` or pass the two
    callables."""
    get = _abstract("get")
    set = _abstract("set")

    def __init__(self, system, bounds=(-np.inf, np.inf), scale=None,
                 init=None, getter=None, setter=None):
        lower, upper = bounds
        width = upper - lower
        if scale is None and not np.isfinite(width):
            raise AssertionError(
                "Variable: without a scale the bounds must be finite "
                "(got %r)" % (bounds,))
        if getter is not None:
            self.get = getter
        if setter is not None:
            self.set = setter
        self.system, self.bounds = system, bounds
        self.scale = width if scale is None else scale
        self.init = init if init is not None else self.get()


class PathVariable(Variable):
    """The attribute or item ``system.get_path(path)`` reaches, e.g.
    ``(1, "curvature")`` (rayopt/optimize.py:46).  ``get`` / ``set`` are
    ordinary methods resolved at call time against ``self.system`` and
    ``self.path``: a subclass may override them (clamp, transform), and a
    variable whose ``system`` is rebound -- or deep-copied together with its
    system -- follows."""
    def __init__(self, system, path, bounds=(-np.inf, np.inf), scale=None,
                 init=None):
        self.path = tuple(path) if isinstance(path, list) else path
        self.system = system        # get() below reads it
        Variable.__init__(self, system, bounds, scale, init)

    def get(self):
        return self.system.get_path(self.path)

    def set(self, value):
        self.system.set_path(self.path, value)


class Operand(_Described):
    """A vector-valued quantity ``get()`` of the system and the roles it
    plays (same constructor as rayopt/optimize.py:58): a ``weight`` adds
    ``sum((weight*(v - offset))**2)`` to the merit; ``min`` / ``max`` bound
    ``v - offset`` from below / above; ``min == max`` pins ``v - offset`` to
    ZERO whatever the common value is -- the reference's behaviour, kept."""
    get = _abstract("get")

    def __init__(self, system, weight=None, offset=0, min=None, max=None):
        self.system, self.weight, self.offset = system, weight, offset
        self.min, self.max = min, max

    def get_objective(self):
        return [Affine(self.weight, self.offset)] if self.weight else []

    def get_equality(self):
        pinned = self.min is not None and self.min == self.max
        return [Affine(1, self.offset)] if pinned else []

    def get_inequality(self):
        # lower: v - offset - min >= 0;  upper: max - (v - offset) >= 0
        return [Affine(sign, self.offset, sign*limit)
                for sign, limit in ((1, self.min), (-1, self.max))
                if limit is not None]


class FuncOp(Operand):
    """``func(system)`` as a flat vector (rayopt/optimize.py:87)."""
    def __init__(self, system, func, weight=None, offset=0, min=None,
                 max=None):
        Operand.__init__(self, system, weight, offset, min, max)
        self.func = func

    def get(self):
        return np.ravel(self.func(self.system))


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
        r = t.rms_fields(lost=self.lost).ravel()
        # after the reduction has been waited for: the trace's events are
        # complete by then and the query costs no round trip of its own
        self.kernel_ms.append(t.kernel_ms())
        return np.where(np.isfinite(r), r, self.penalty)

    def get_variants(self, systems):
        """The operand for V variants of the system (same number of
        elements) in ONE aiming launch, ONE trace and ONE reduction:
        ``(V, wavelengths*fields)``, row v equal to what :meth:`get` gives on
        ``systems[v]``.  This is what makes a finite-difference gradient cost
        one evaluation instead of one per variable (``optimize(...,
        jac="batched")``).  Group ``v*W + w`` of the batch is variant v at
        wavelength w: its own surface table, its own aimed pupil per field."""
        from ._lib import AIM_ARGS_DTYPE
        from .aiming import start_pupil, entrance_pupil
        from .launch import aim_seeds, field_frames
        from .pack import pack_system
        from .pupil import pupil_distribution
        l = self.wavelengths
        if l is None:
            l = self.system.wavelengths
        l = np.atleast_1d(np.asarray(l, dtype=float))
        fields, nf, nw, nv = self.fields, len(self.fields), len(l), \
            len(systems)
        engine = self._variants_engine()
        ref, yp, weight = pupil_distribution(self.distribution, self.nrays)
        alive = len(yp)
        pad = -alive % 64
        yp = np.concatenate([yp, np.full((pad, 2), np.nan)])
        if weight is None:
            weight = np.ones(alive)/alive
        weight = np.concatenate([weight, np.zeros(pad)])
        groups = [(s, li) for s in systems for li in l]
        tables = np.stack([
            pack_system(s, li, s.refractive_index(li, 0))[0]
            for s, li in groups])
        engine.upload_system(tables)
        starts = [start_pupil(s, li) if self.aim else entrance_pupil(s, li)
                  for s, li in groups]
        if self.aim:
            args = np.zeros((), dtype=AIM_ARGS_DTYPE)
            args["stop"], args["rim"] = self.system.stop, False
            args["maxiter"], args["tol"] = 60, 1e-9
            seeds = np.concatenate([
                aim_seeds(s, fields, z0, a0, g)
                for g, ((s, li), (z0, a0)) in enumerate(zip(groups, starts))])
            z, a, status = engine.aim_pupil(seeds, args)
            z, a = z.reshape(len(groups), nf), a.reshape(len(groups), nf, 2, 2)
            failed = status.reshape(len(groups), nf).any(1)
        else:
            z = [np.broadcast_to(z0, (nf,)) for z0, _ in starts]
            a = [a0 for _, a0 in starts]
            failed = np.zeros(len(groups), dtype=bool)
        z = np.where(np.isfinite(z), z, 0.) if self.aim else z
        frames = np.concatenate([
            field_frames(s, fields, z[g], np.nan_to_num(a[g]) if self.aim
                         else a[g])
            for g, (s, li) in enumerate(groups)])
        engine.generate_rays(frames, yp)
        engine.set_weights(np.tile(weight, len(frames)))
        mask = np.zeros(len(self.system), dtype=np.uint8)
        mask[0] = mask[-1] = 1
        engine.set_keep_rows(mask)
        engine.trace(1, 0, self.clip)
        self.kernel_ms.append(engine.kernel_ms())
        # count and spread of every bundle in one pass (rt_row_stats)
        stats = engine.row_stats(len(self.system) - 1, len(yp), len(frames))
        stats = stats[:, (0, 2, 3, 4)]      # count, centroid, spread
        r = np.sqrt(stats[:, 3])
        if self.lost == "nan":
            r = np.where(stats[:, 0] < alive, np.nan, r)
        r = r.reshape(len(groups), nf)
        r[failed] = np.nan                   # a variant that cannot be aimed
        r = np.where(np.isfinite(r), r, self.penalty)
        return r.reshape(nv, nw*nf)

    def _variants_engine(self):
        """A second context for variant batches, so the operand's own trace
        keeps its rays and results."""
        if getattr(self, "_vengine", None) is None:
            base = self.trace.engine
            from .engine import Engine
            self._vengine = Engine(self.trace._device) \
                if isinstance(base, Engine) else type(base)()
        return self._vengine


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

    # -- finite differences over system variants, one launch -----------------
    def batchable(self):
        """True if the forward-difference gradient of the merit can be taken
        from ONE batched evaluation: every variable is a path into the same
        system, every weighted operand can evaluate system variants."""
        system = self.operands[0].system
        return (not self.equality and not self.inequality and
                all(getattr(v, "path", None) is not None and
                    v.system is system for v in self.variables) and
                all(hasattr(self.operands[k], "get_variants") and
                    self.operands[k].system is system
                    for k, _ in self.objective))

    def gradient(self, x, eps=1e-5):
        """Forward differences ``(f(x + eps e_k) - f(x))/eps`` (backward
        where the step would leave the box), all ``len(x) + 1`` points
        evaluated as variants of the system in one launch per operand."""
        import copy
        x = np.asarray(x, dtype=float)
        system = self.operands[0].system
        nvar = len(self.variables)
        if getattr(self, "_copies", None) is None or \
                len(self._copies) != nvar + 1:
            self._copies = [copy.deepcopy(system) for _ in range(nvar + 1)]
        steps = np.where(x + eps > self.bounds[:, 1], -eps, eps)
        points = np.tile(x, (nvar + 1, 1))
        points[np.arange(1, nvar + 1), np.arange(nvar)] += steps
        for clone, point in zip(self._copies, points):
            for var, value in zip(self.variables, point*self.scale):
                clone.set_path(var.path, value)
        values = {k: self.operands[k].get_variants(self._copies)
                  for k in {k for k, _ in self.objective}}
        self.evaluations += 1
        merits = np.array([
            np.square(np.concatenate([f(values[k][p])
                                      for k, f in self.objective])).sum()
            for p in range(nvar + 1)])
        self._batched = (tuple(x), merits[0])
        return (merits[1:] - merits[0])/steps

    def merit_batched(self, x):
        """The merit at ``x``, from the gradient batch if it was there."""
        hit = getattr(self, "_batched", None)
        if hit is not None and hit[0] == tuple(np.asarray(x, dtype=float)):
            return hit[1]
        return self.merit(x)

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
    points the operands were evaluated at.  ``jac="batched"`` (extension):
    forward-difference gradients whose ``len(variables) + 1`` points are
    traced as variants of the system in one launch
    (:meth:`SpotOperand.get_variants`); a gradient then costs about one merit
    evaluation instead of one per variable."""
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
    fun = problem.merit
    if isinstance(kwargs.get("jac"), str) and kwargs["jac"] == "batched":
        # extension: the whole finite-difference gradient from one launch
        if not problem.batchable():
            raise ValueError("jac='batched' needs PathVariables of one "
                             "system, operands with get_variants() and no "
                             "constraints")
        step = settings.pop("eps")
        kwargs["jac"] = lambda x: problem.gradient(x, step)
        fun = problem.merit_batched
    result = minimize(fun, problem.start, bounds=problem.bounds,
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
