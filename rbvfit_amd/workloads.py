"""Synthetic workloads of SURVEY.md section 8(d) / BASELINE.json ``configs`` (C0..C4).

Product-side setup used by ``bench.py`` and the full-size GPU tests: physics tables from
``rbvfit_amd.model``, clean spectra from the engine itself, seeded Gaussian noise, the reference's
traditional bounds, walker cloud around the truth (``_initialize_walkers``, vfit_mcmc.py:442-466,
with a 1e-3 spread).  No oracle here.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

from . import _lib as L
from .engine import Engine
from .model import FitConfiguration, VoigtModel
from .vfit import set_bounds

MGII = [(0.348, "MgII", [2796.35, 2803.53], 2)]
MULTI = [(0.348, "MgII", [2796.35, 2803.53], 3),
         (0.348, "FeII", [2600.17, 2586.65, 2382.77], 3),
         (0.348, "CIV", [1548.20, 1550.77], 2)]
STRESS = [(0.348 + 0.01 * i, "MgII", [2796.35, 2803.53], 8) for i in range(4)]


def cos_like_kernel() -> np.ndarray:
    """Synthetic tabulated 'COS-like' LSF of C3 (the real COS tables need linetools + network)."""
    j = np.arange(-50, 51, dtype=np.float64)
    g = np.exp(-0.5 * (j / 2.2) ** 2) / (np.sqrt(2 * np.pi) * 2.2)
    lor = (6.0 / np.pi) / (j ** 2 + 36.0)
    return (0.85 * g + 0.15 * lor) * (1.0 + 0.002 * j)


def _theta_random(seed, C):
    rng = np.random.default_rng(seed)
    return np.concatenate([rng.uniform(12.8, 13.8, C), rng.uniform(8, 35, C), rng.uniform(-120, 120, C)])


SPECS = {
    # name: (physics, instruments [(name, lo, hi, P, FWHM, tabulated?)], default W, theta_true, seed)
    "C0": (MGII, [("G", 3755.0, 3795.0, 4096, "6.5", False)], 50, np.array([13.5, 13.2, 15.0, 25.0, -40.0, 20.0]), 11),
    "C1": (MGII, [("G", 3755.0, 3795.0, 4096, "6.5", False)], 512, np.array([13.5, 13.2, 15.0, 25.0, -40.0, 20.0]), 11),
    "C2": (MULTI, [("G", 2050.0, 3800.0, 16384, "6.5", False)], 1024, None, 12),
    "C3": (MULTI, [("A", 2050.0, 2925.0, 8192, "6.5", True), ("B", 2925.0, 3800.0, 8192, "2.5", False)], 2048, None, 13),
    "C4": (STRESS, [("G", 3700.0, 3900.0, 65536, "6.5", False)], 4096, None, 14),
}


@dataclass
class Workload:
    name: str
    engine: Engine
    thetas: np.ndarray
    theta_true: np.ndarray
    lb: np.ndarray
    ub: np.ndarray
    pixels: List[int]
    n_lines: int
    tables: list
    spectra: list = field(default_factory=list)     # [(wave, flux, error)] per instrument

    @property
    def ndim(self):
        return self.theta_true.size

    @property
    def algorithmic_bytes_per_eval(self) -> int:
        """read wave, flux, inv_sigma2 once per walker-eval, theta, write lnprob."""
        return sum(24 * p for p in self.pixels) + 8 * self.ndim + 8

    @property
    def algorithmic_flops_per_eval(self) -> float:
        """SURVEY 8(d) secondary roof: per (line, pixel) 8 (x) + H (12 wing / 150 core) + 1, per
        pixel 30 (exp) + 2K + 4; "core" = |x| < 8 Doppler widths at theta_true."""
        total = 0.0
        for data, (wave, _, _) in zip(self.tables, self.spectra):
            th = self.theta_true
            b = th[np.asarray(data.b_indices)]
            v = th[np.asarray(data.v_indices)]
            lam_obs = np.asarray(data.atomic_lambda0) * np.asarray(data.z_factors) * (1.0 + v / 299792.458)
            x = 299792.458 * (lam_obs[:, None] / wave[None, :] - 1.0) / b[:, None]
            ncore = int(np.count_nonzero(np.abs(x) < 8.0))
            L, P = x.shape
            K = 0 if data.taps is None else len(data.taps)
            total += 9.0 * L * P + 150.0 * ncore + 12.0 * (L * P - ncore) + P * (30.0 + 2.0 * K + 4.0)
        return total


def make_workload(name: str, walkers: Optional[int] = None, device_id: int = 0, pixels: Optional[int] = None,
                  walker_seed: int = 1) -> Workload:
    physics, insts, W0, theta_true, seed = SPECS[name]
    W = int(walkers if walkers is not None else W0)
    cfg = FitConfiguration()
    for z, ion, trans, nc in physics:
        cfg.add_system(z, ion, trans, nc)
    C = cfg.total_components
    if theta_true is None:
        theta_true = _theta_random(seed, C)
    rng = np.random.default_rng(seed)
    _, lb, ub = set_bounds(theta_true[:C], theta_true[C:2 * C], theta_true[2 * C:])
    eng = Engine(device_id)
    eng.set_bounds(lb, ub)
    tables, pix, spectra = [], [], []
    for iname, lo, hi, P, fwhm, tabulated in insts:
        P = int(pixels) if pixels else P
        # raw (astropy-4.x) Gaussian taps, as in the golden fixtures and the SURVEY 8(a) anchors
        model = VoigtModel(cfg, FWHM=fwhm, kernel_taps=cos_like_kernel() if tabulated else None, normalize_kernel=False)
        data = model.compile().data
        wave = np.linspace(lo, hi, P)
        err = np.full(P, 0.05)
        ones = np.ones(P)
        idx = eng.add_instrument(wave, ones, 1.0 / err ** 2, np.log(1.0 / err ** 2), **data.engine_kwargs())
        clean = eng.model_flux(idx, theta_true)[0]
        flux = clean + rng.normal(0.0, 0.05, P)
        eng.update_spectrum(idx, flux, 1.0 / err ** 2, np.log(1.0 / err ** 2))
        tables.append(data); pix.append(P); spectra.append((wave, flux, err))
    wr = np.random.default_rng(walker_seed)
    thetas = theta_true + 1e-3 * wr.standard_normal((W, theta_true.size))
    thetas = np.clip(thetas, lb + 1e-10, ub - 1e-10)
    return Workload(name, eng, thetas, theta_true, lb, ub, pix, tables[0].n_lines, tables, spectra)
