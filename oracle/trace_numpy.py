"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the sequential geometric trace.

A numpy restatement of the reference algorithm for the hot path
``GeometricTrace.propagate`` (quartiq/rayopt).  It is the *checker* for the
HIP engine and the ``cpu_baseline`` ("port") leg of ``bench.py``; it is never
imported by the product package ``rayopt_amd`` and is not a fallback.

Parity pinning: this oracle is validated (tests/test_oracle_golden.py)
  * against golden vectors produced by running the unmodified reference in
    the build container (tests/golden/*.npz, generator
    tests/golden/make_golden.py), which include the reference's own
    Cooke-triplet fixture and its pinned ``rms`` known-answer
    (rayopt/test/test_raytrace.py:36-44,189-199), and
  * directly against the in-place imported reference whenever
    ``/root/reference`` is present (oracle/refshim.py).

It deliberately performs the same whole-array numpy operations, in the same
order, as the reference (AoS ``(N,3)`` float64 arrays, one temporary per
ufunc), so (i) its results are bit-identical to the reference's -- for every
surface type, tilted elements and iterated aspheres included -- and (ii) its
timing is representative of the reference's numpy path.  It consumes the same ``rt_surface`` table as the GPU
kernel (a numpy structured array, see include/rt_mi355.h).

The one place where the reference is not numpy is the even-asphere intercept:
a Python loop over rays calling ``scipy.optimize.newton``
(rayopt/elements.py:333-349).  That algorithm lives in SciPy (not vendored;
1.15.3 in the build container; ``setup.py`` pins no version); its scalar
Newton-Raphson branch is restated here in masked, vectorised form
(``newton_intercept``) and checked ray-for-ray, bit for bit, against the real
thing; its derivative is a BLAS dot in the reference (np.dot of a (1,3) with a
(3,1) array), i.e. a chain of fused multiply-adds, reproduced with ``fma``.
"""
import numpy as np

F_ROTATED, F_CURVED, F_CONIC, F_ASPH, F_ALT, F_REFRACT, F_MIRROR = (
    0x01, 0x02, 0x04, 0x08, 0x10, 0x20, 0x40)


def _two_sum(a, b):
    s = a + b
    bb = s - a
    return s, (a - (s - bb)) + (b - bb)


def _two_product(a, b):
    """a*b = p + e exactly (Dekker / Veltkamp, no hardware fma needed)."""
    p = a*b
    c = 134217729.                       # 2^27 + 1
    ah = a*c
    ah = ah - (ah - a)
    al = a - ah
    bh = b*c
    bh = bh - (bh - b)
    bl = b - bh
    return p, ((ah*bh - p) + ah*bl + al*bh) + al*bl


def fma(a, b, c):
    """Correctly rounded a*b + c on arrays, from IEEE + - * alone (Boldo and
    Melquiond, "Emulation of a FMA and correctly rounded sums", 2008: the
    small terms are added with rounding to odd, which makes the final
    rounding that of the exact sum).  The reference reaches fused
    multiply-adds through BLAS -- np.dot of a (1,3) with a (3,1) array in
    Interface.intercept's fprime (rayopt/elements.py:342) sums
    fma(q2, u2, fma(q1, u1, q0*u0)); checked against exact rational
    arithmetic -- and numpy itself has no fma.  Finite, non-extreme operands
    (no overflow / underflow of the partial terms), NaN in NaN out."""
    a, b, c = np.broadcast_arrays(np.asarray(a, float), np.asarray(b, float),
                                  np.asarray(c, float))
    uh, ul = _two_product(a, b)
    th, tl = _two_sum(c, uh)
    v, err = _two_sum(tl, ul)
    # round-to-odd of tl + ul: if inexact and v came out even, its odd
    # neighbour on the side of the error is the one
    bits = v.view(np.int64) if v.flags.c_contiguous else \
        np.ascontiguousarray(v).view(np.int64)
    inexact = (err != 0) & np.isfinite(v)
    even = (bits & 1) == 0
    toward = np.where(err > 0, np.inf, -np.inf)
    v = np.where(inexact & even, np.nextafter(v, toward), v)
    out = th + v
    # infinities (and anything the error-free steps turned into NaN): the
    # plain expression has the right special value
    return np.where(np.isfinite(out), out, a*b + c)


def _rot(s):
    return np.asarray(s["rot"], dtype=float).reshape(3, 3)


def to_normal(s, *v):
    """TransformMixin.to_normal: y @ R.T (rayopt/elements.py:156-163,174)."""
    if int(s["flags"]) & F_ROTATED:
        r = _rot(s).T
        return tuple(np.dot(vi, r) for vi in v)
    return v


def from_normal(s, *v):
    """TransformMixin.from_normal: y @ R (rayopt/elements.py:171-172)."""
    if int(s["flags"]) & F_ROTATED:
        r = _rot(s)
        return tuple(np.dot(vi, r) for vi in v)
    return v


def surface_sag(s, xyz):
    """Spheroid.surface_sag (rayopt/elements.py:440-455)."""
    flags = int(s["flags"])
    e = xyz[..., 2].copy()
    if not flags & (F_CURVED | F_ASPH):
        return e
    xy = xyz[..., :2]
    r2 = np.einsum("...i,...i", xy, xy)
    if flags & F_CURVED:
        c = float(s["c"])
        e -= c*r2/(1 + np.sqrt(1 - float(s["kc2"])*r2))
    if flags & F_ASPH:
        d = 0.
        for ai in s["asph"][:int(s["nasph"])][::-1]:
            d += float(ai)
            d *= r2
        e -= d
    return e


def surface_normal(s, xyz):
    """Spheroid.surface_normal (rayopt/elements.py:457-475)."""
    flags = int(s["flags"])
    q = np.zeros_like(xyz)
    q[..., 2] = 1
    if not flags & (F_CURVED | F_ASPH):
        return q
    xy = xyz[..., :2]
    r2 = np.einsum("...i,...i", xy, xy)
    e = 0.
    if flags & F_CURVED:
        e -= float(s["c"])/np.sqrt(1 - float(s["kc2"])*r2)
    if flags & F_ASPH:
        d = 0.
        for di in s["dasph"][:int(s["nasph"])][::-1]:
            d *= r2
            d += float(di)
        e -= d
    q[..., :2] = xy*e[..., None]
    return q


# Census of the iteration (tests of rt_newton_census only): when this is a
# list, newton_intercept appends per call (iterates, dead) -- the iterates
# every ray went through until its result was decided (an iterate that is NaN
# decides it: every later one is NaN as well and the solver ends in NaN), and
# which rays arrived with a NaN direction.
ITERATES = None


def newton_intercept(s, y, u, tol=1e-7, maxiter=5):
    """Interface.intercept (rayopt/elements.py:333-349).

    Per ray: scipy.optimize.newton(func=sag(y + s u), fprime=normal(y + s u).u,
    x0=-y_z/u_z, tol=1e-7, maxiter=5), RuntimeError -> NaN.  Scalar
    Newton-Raphson semantics of scipy 1.15.3 (_zeros_py.py): fval == 0 returns
    the current iterate; fder == 0 raises; p = p0 - fval/fder; np.isclose(p,
    p0, rtol=0, atol=tol) returns p; exhausting maxiter raises.
    """
    p0 = -y[:, 2]/u[:, 2]
    out = np.full(p0.shape, np.nan)
    live = np.ones(p0.shape, dtype=bool)
    iterates = np.zeros(p0.shape, dtype=np.int64)
    with np.errstate(all="ignore"):
        for itr in range(maxiter):
            if not live.any():
                break
            idx = np.nonzero(live)[0]
            iterates[idx] += 1
            yi, ui, pi = y[idx], u[idx], p0[idx]
            xyz = yi + pi[:, None]*ui
            fval = surface_sag(s, xyz)
            zero = fval == 0
            out[idx[zero]] = pi[zero]
            # np.dot(normal (1,3), ui.T (3,1)) (:342): a BLAS call, which
            # sums in order with fused multiply-adds
            q = surface_normal(s, xyz)
            fder = fma(q[:, 2], ui[:, 2],
                       fma(q[:, 1], ui[:, 1], q[:, 0]*ui[:, 0]))
            dzero = (fder == 0) & ~zero      # "Derivative was zero" -> NaN
            p = pi - fval/fder
            fin = np.isfinite(p) & np.isfinite(pi)
            close = np.where(fin, np.abs(p - pi) <= tol, p == pi)
            conv = close & ~zero & ~dzero
            out[idx[conv]] = p[conv]
            # (a NaN iterate: nothing changes any more -- out stays NaN)
            done = zero | dzero | conv | np.isnan(p)
            p0[idx] = p
            live[idx[done]] = False
    if ITERATES is not None:
        ITERATES.append((iterates, np.isnan(u[:, 0])))
    return out


def intercept(s, y, u):
    """Spheroid.intercept (rayopt/elements.py:477-501)."""
    flags = int(s["flags"])
    if flags & F_ASPH:
        return newton_intercept(s, y, u)
    c = float(s["c"])
    if not flags & F_CURVED:
        return -y[:, 2]/u[:, 2]
    if not flags & F_CONIC:
        uy = (u*y).sum(1)
        uu = 1.
        yy = np.square(y).sum(1)
    else:
        k = np.array([(1, 1, float(s["kw"]))])
        uy = (u*y*k).sum(1)
        uu = (np.square(u)*k).sum(1)
        yy = (np.square(y)*k).sum(1)
    d = c*uy - u[:, 2]
    e = c*uu
    f = c*yy - 2*y[:, 2]
    g = np.sqrt(np.square(d) - e*f)
    if flags & F_ALT:
        g *= -1
    return -(d + g)/e


def clip(s, y, u):
    """Element.clip (rayopt/elements.py:206-209)."""
    good = np.square(y[:, :2]).sum(1) <= float(s["radius2"])
    return np.where(good[:, None], u, np.nan)


def refract(s, y, u0):
    """Interface.refract, Spencer & Murty (rayopt/elements.py:351-369)."""
    mu = float(s["mu"])
    if mu == 1:
        return u0
    r = surface_normal(s, y)
    r2 = np.square(r).sum(1)
    muf = abs(mu)
    a = muf*(u0*r).sum(1)/r2
    if mu == -1:
        return u0 - 2*a[:, None]*r
    b = (mu**2 - 1)/r2
    g = -a + np.sign(mu)*np.sqrt(np.square(a) - b)
    return muf*u0 + g[:, None]*r


def element_propagate(s, y0, u0, do_clip):
    """Interface.propagate (rayopt/elements.py:306-315)."""
    t = intercept(s, y0, u0)
    y = y0 + t[:, None]*u0
    if do_clip:
        u0 = clip(s, y, u0)
    u = u0
    if float(s["mu"]):
        u = refract(s, y, u0)
    return y, u, t*float(s["n0"])


def propagate(table, y, u, start=1, stop=None, clip=False):
    """GeometricTrace.propagate + System.propagate.

    (rayopt/geometric_trace.py:72-80, rayopt/system.py:459-464).  ``y, u`` are
    rows ``start-1`` of the trace, (N,3), in the normal frame of element
    ``start-1``.  Returns ``Y, U, I`` of shape (rows, N, 3) and ``T`` of shape
    (rows, N) for elements ``start .. stop-1``.
    """
    idx = range(len(table))[start:stop]
    y = np.array(y, dtype=float)
    u = np.array(u, dtype=float)
    ny = len(idx)
    Y = np.empty((ny,) + y.shape)
    U = np.empty_like(Y)
    I = np.empty_like(Y)
    T = np.empty((ny, y.shape[0]))
    with np.errstate(all="ignore"):
        y, u = from_normal(table[idx.start - 1], y, u)
        for row, j in enumerate(idx):
            s = table[j]
            y, i = to_normal(s, y - s["offset"], u)
            y, u, t = element_propagate(s, y, i, clip)
            Y[row], U[row], I[row], T[row] = y, u, i, t
            y, u = from_normal(s, y, u)
    return Y, U, I, T
