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


def pupil_distribution(distribution, nrays, rng=None):
    """Return ``(ref, xy, weight)``: index of the reference ray, (n,2)
    coordinates, weights (None unless a quadrature rule)."""
    d, n = distribution, nrays
    ref, weight = 0, None
    lin = np.linspace
    if n == 1:
        xy = np.zeros((1, 2))
    elif d == "half-meridional":
        xy = np.c_[np.zeros(n), lin(0, 1, n)]
    elif d == "meridional":
        n -= n % 2
        xy = np.c_[np.zeros(n + 1), lin(-1, 1, n + 1)]
    elif d == "sagittal":
        n -= n % 2
        ref = n//2
        xy = np.c_[lin(-1, 1, n + 1), np.zeros(n + 1)]
    elif d == "cross":
        n -= n % 4
        ref = n//4
        h = n//2 + 1
        xy = np.r_[np.c_[np.zeros(h), lin(-1, 1, h)],
                   np.c_[lin(-1, 1, h), np.zeros(h)]]
    elif d == "tee":
        n = (n - 2)//3
        ref = 2*n + 1
        xy = np.r_[np.c_[np.zeros(2*n + 1), lin(-1, 1, 2*n + 1)],
                   np.c_[lin(0, 1, n + 1), np.zeros(n + 1)]]
    elif d == "random":
        rng = np.random.default_rng() if rng is None else rng
        r, phi = rng.random((2, n))
        z = np.exp(2j*np.pi*phi)*np.sqrt(r)
        xy = np.r_[[[0., 0.]], np.c_[z.real, z.imag]]
    elif d == "square":
        xy = _grid_in_circle(n)
    elif d == "triangular":
        xy = _grid_in_circle(n, stagger=True)
    elif d == "hexapolar":
        rings = int(np.sqrt(n/3. - 1/12.) - 1/2.)
        parts = [np.zeros((1, 2))]
        for i in range(1, rings + 1):
            a = lin(0, 2*np.pi, 6*i, endpoint=False)
            parts.append(np.c_[np.sin(a)*i/rings, np.cos(a)*i/rings])
        xy = np.concatenate(parts)
    elif d in ("radau", "lobatto"):
        k = int(np.sqrt(n) + 1)
        x, w = _radau_nodes(k) if d == "radau" else _lobatto_nodes(k)
        r, p, weight = _disc_quadrature(x, w)
        xy = np.c_[r*np.cos(p), r*np.sin(p)]
    else:
        raise ValueError("unknown ray distribution", d)
    return ref, xy, weight
