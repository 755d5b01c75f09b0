"""Host post-processing of the per-ray OPD the device produced: resampling of
the scattered exit-pupil samples onto a regular grid and the FFT that turns
the pupil function into a point spread function.  These operate on n x n
grids (n ~ 4 sqrt(N)), not on ray batches; conventions follow
rayopt/geometric_trace.py:132-169."""
import numpy as np
from scipy.interpolate import griddata


def resample_pupil(x, y, t, n):
    """Linear interpolation of the finite samples ``t(x, y)`` onto an n x n
    grid spanning the largest |coordinate|; NaN outside the convex hull."""
    ok = np.isfinite(x) & np.isfinite(y) & np.isfinite(t)
    if not ok.any():
        raise ValueError("no rays made it through")
    x, y, t = x[ok], y[ok], t[ok]
    half = max(np.fabs(x).max(), np.fabs(y).max())
    gx, gy = np.mgrid[-1:1:1j*n, -1:1:1j*n]*half
    gt = griddata((x, y), t, (gx, gy), method="linear", fill_value=np.nan)
    return gx, gy, gt


def psf_from_opd(gx, opd, pad, radius, wavelength):
    """|FFT|^2 of the unit-amplitude pupil function exp(-2 pi i opd), zero
    padded ``pad`` times; returns image-plane coordinates p, q and the PSF
    normalised to unit sum (constant amplitude across the pupil assumed)."""
    inside = np.isfinite(opd)
    count = np.count_nonzero(inside)
    pupil = np.where(inside, np.exp(-2j*np.pi*opd), 0)/count**.5
    shape = tuple(k*pad for k in pupil.shape)
    amp = np.fft.fft2(pupil, shape)
    psf = (amp*amp.conj()).real/amp.size
    step = gx[1, 0] - gx[0, 0]
    freq = np.fft.fftfreq(shape[0], step/(wavelength*radius))
    p, q = np.broadcast_arrays(freq[:, None], freq)
    return p, q, psf
