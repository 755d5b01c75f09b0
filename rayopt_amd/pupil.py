"""Pupil sampling patterns: normalised aperture coordinates (x sagittal,
y meridional) inside the unit circle, with the index of the centre/reference
ray and optional quadrature weights -- the patterns and conventions of
rayopt's ``pupil_distribution`` (rayopt/utils.py:117-199), the generator of
the pupil grids that ``GeometricTrace.rays_fields`` expands on the GPU.
"""
import numpy as np
from numpy.polynomial import legendre as L


def _grid_in_circle(n, stagger=False):
    side = int(np.sqrt(n*4/np.pi))
    g = np.mgrid[-1:1:1j*side, -1:1:1j*side]
    if stagger:
        g[0] += (np.arange(side) % 2.)*(2./side)
    g = g.reshape(2, -1)
    inside = (g**2).sum(0) <= 1
    return np.concatenate([[[0., 0.]], g[:, inside].T])


def _legendre(k):
    return L.Legendre.basis(k)


def _radau_nodes(n):
    """Gauss-Radau nodes/weights on [-1,1], node -1 first."""
    p = _legendre(n - 1)
    q = (p + _legendre(n)).convert(kind=np.polynomial.Polynomial)
    quotient = q // np.polynomial.Polynomial((1., 1.))
    x = np.r_[-1., np.sort(quotient.roots().real)]
    w = (1 - x)/(n*p(x))**2
    return x, w


def _lobatto_nodes(n):
    """Gauss-Lobatto nodes/weights on [-1,1], end points included."""
    p = _legendre(n - 1)
    x = np.r_[-1., np.sort(p.deriv().roots().real), 1.]
    w = 2/(n*(n - 1)*p(x)**2)
    return x, w


def _disc_quadrature(x, w):
    """Product rule on the unit disc from radial nodes on [-1,1] and a
    uniform half-turn of azimuths."""
    n = len(x)
    r = ((x + 1.)/2.)**.5
    phi = np.pi*((np.arange(n) + .5)/n - .5)
    if r[0] == 0.:
        rs = np.r_[r[0], np.repeat(r[1:], n)]
        ws = np.r_[w[0], np.repeat(w[1:]/n, n)]/2
        ps = np.r_[0, np.tile(phi, n - 1)]
    else:
        rs = np.repeat(r, n)
        ws = np.repeat(w/n, n)/2
        ps = np.tile(phi, n)
    return rs, ps, ws


# --------------------------------------------------------------------------
# the patterns: name -> function(nrays, rng) -> (ref, xy, weight)
#
# A pattern is a set of normalised pupil coordinates, the index of its
# reference ray and, for the quadrature rules, weights.  Ray counts, ordering
# and the reference index are part of the contract with the reference
# (rayopt/utils.py:117-199): Analysis indexes into the bundles it requests
# (rayopt/analysis.py:236-249 splits a "tee" fan at ``ref``).
# --------------------------------------------------------------------------

def _fan(lo, n, axis):
    """n points from lo to 1 along one pupil axis (0 sagittal, 1 meridional)."""
    xy = np.zeros((n, 2))
    xy[:, axis] = np.linspace(lo, 1, n)
    return xy


def _half_meridional(n, rng):
    return 0, _fan(0, n, 1), None


def _meridional(n, rng):
    return 0, _fan(-1, n - n % 2 + 1, 1), None


def _sagittal(n, rng):
    n -= n % 2
    return n//2, _fan(-1, n + 1, 0), None


def _cross(n, rng):
    n -= n % 4
    arm = n//2 + 1
    return n//4, np.r_[_fan(-1, arm, 1), _fan(-1, arm, 0)], None


def _tee(n, rng):
    half = (n - 2)//3
    return (2*half + 1,
            np.r_[_fan(-1, 2*half + 1, 1), _fan(0, half + 1, 0)], None)


def _random(n, rng):
    rng = np.random.default_rng() if rng is None else rng
    r, phi = rng.random((2, n))
    z = np.exp(2j*np.pi*phi)*np.sqrt(r)
    return 0, np.r_[[[0., 0.]], np.c_[z.real, z.imag]], None


def _hexapolar(n, rng):
    rings = int(np.sqrt(n/3. - 1/12.) - 1/2.)
    parts = [np.zeros((1, 2))]
    for ring in range(1, rings + 1):         # 6, 12, 18 ... points per ring
        a = np.linspace(0, 2*np.pi, 6*ring, endpoint=False)
        # (sin a * ring)/rings, in this order: the reference's rounding
        parts.append(np.c_[np.sin(a)*ring/rings, np.cos(a)*ring/rings])
    return 0, np.concatenate(parts), None


def _quadrature(nodes):
    def pattern(n, rng):
        r, phi, weight = _disc_quadrature(*nodes(int(np.sqrt(n) + 1)))
        return 0, np.c_[r*np.cos(phi), r*np.sin(phi)], weight
    return pattern


PATTERNS = {
    "half-meridional": _half_meridional, "meridional": _meridional,
    "sagittal": _sagittal, "cross": _cross, "tee": _tee, "random": _random,
    "square": lambda n, rng: (0, _grid_in_circle(n), None),
    "triangular": lambda n, rng: (0, _grid_in_circle(n, stagger=True), None),
    "hexapolar": _hexapolar,
    "radau": _quadrature(_radau_nodes), "lobatto": _quadrature(_lobatto_nodes),
}


_PATTERN_CACHE = {}


def pupil_distribution(distribution, nrays, rng=None):
    """Return ``(ref, xy, weight)``: index of the reference ray, (n,2)
    coordinates, weights (None unless a quadrature rule).  The deterministic
    patterns are computed once per ``(distribution, nrays)`` -- a merit
    function asks for the same one at every evaluation -- and handed out as
    copies."""
    if nrays == 1:
        return 0, np.zeros((1, 2)), None
    try:
        pattern = PATTERNS[distribution]
    except KeyError:
        raise ValueError("unknown ray distribution", distribution) from None
    if distribution == "random":
        return pattern(nrays, rng)
    key = distribution, nrays
    if key not in _PATTERN_CACHE:
        if len(_PATTERN_CACHE) > 64:
            _PATTERN_CACHE.clear()
        _PATTERN_CACHE[key] = pattern(nrays, rng)
    ref, xy, weight = _PATTERN_CACHE[key]
    return ref, xy.copy(), None if weight is None else weight.copy()
