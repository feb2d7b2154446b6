"""LSF taps on the host (setup only; the convolution itself runs in the HIP tile kernel)."""
from __future__ import annotations

import numpy as np


def gaussian_taps(fwhm_pixels, normalize: bool = False) -> np.ndarray:
    """Taps the reference obtains from ``Gaussian1DKernel(stddev=float(FWHM)/2.355)``
    (core/voigt_model.py:462-464; the literal 2.355 is the reference's, trap T3): 8 sigma rounded
    up to the next odd integer, sampled at integer pixel offsets.

    ``normalize=False`` gives the raw samples (what astropy 4.x stores in ``kernel.array``);
    ``normalize=True`` divides by their sum (what astropy >= 5 stores).  Parity is defined on the
    taps actually passed to the engine -- they are data at the C ABI (SURVEY trap T2)."""
    sigma = float(fwhm_pixels) / 2.355
    size = int(np.ceil(8 * sigma))
    if size % 2 == 0:
        size += 1
    j = np.arange(size, dtype=np.float64) - size // 2
    taps = (1.0 / (np.sqrt(2 * np.pi) * sigma)) * np.exp(-0.5 * j ** 2 / sigma ** 2)
    return taps / taps.sum() if normalize else taps
