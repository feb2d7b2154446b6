#!/opt/conda/bin/python3.9
"""Golden values of the reference's host helpers next to the hot path (build container only, as make_golden.py):

    PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 tests/golden/make_golden_host.py

``mean_fwhm_pixels`` (core/voigt_model.py:33-58) on three wavelength grids -> tests/golden/host_helpers.npz (data only:
the grids' parameters, the FWHM values in km/s and the reference's results)."""
import os
import sys
import types

import numpy as np

np.asscalar = lambda a: np.asarray(a).item()          # astropy 4.3.1 compat (see make_golden.py)
np.alen = lambda a: len(np.asarray(a))
if not hasattr(np, "rank"):
    np.rank = lambda a: np.ndim(a)
for _name in ("emcee", "corner"):
    if _name not in sys.modules:
        _m = types.ModuleType(_name)
        _m.__version__ = "placeholder-not-installed"
        _m.EnsembleSampler = None
        sys.modules[_name] = _m
sys.path.insert(0, "/root/reference/src")
from rbvfit.core.voigt_model import mean_fwhm_pixels   # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def grids():
    """(name, wavelength grid): linear (C1's), log-spaced (constant velocity pixels), and an irregular one."""
    rng = np.random.default_rng(21)
    yield "linear", np.linspace(3755.0, 3795.0, 4096)
    yield "loglam", 1200.0 * np.exp(np.arange(3000) * (2.5 / 299792.458))
    yield "irregular", np.cumsum(rng.uniform(0.01, 0.05, 777)) + 5000.0


def main():
    out = {}
    fwhm = np.array([6.5, 18.0, 2.394991274145626, 150.0])
    out["fwhm_kms"] = fwhm
    for name, w in grids():
        out[f"{name}__wave"] = w
        out[f"{name}__pixels"] = np.array([mean_fwhm_pixels(float(f), w) for f in fwhm])
    np.savez_compressed(os.path.join(HERE, "host_helpers.npz"), **out)
    print({k: v for k, v in out.items() if k.endswith("__pixels")})


if __name__ == "__main__":
    main()
