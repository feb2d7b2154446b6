"""Curve-of-growth grid as ONE GPU batch.

Mirrors ``compute_cog`` / ``set_one_absorber`` / ``compute_ewlist_from_voigt`` of the reference
(src/rbvfit/compute_cog.py:23-183): a single-line, single-component model at z = 0 with no LSF,
evaluated on ``linspace(lam_rest - 5, lam_rest + 5, 1000)`` for every (log N, b) pair, equivalent
width by trapezoidal integration of 1 - flux.  The reference loops ``model_flux`` over the grid
one theta at a time; here the whole (N, b) grid is one ``model_flux`` batch and the integral is a reduction
on the device behind it.  Plotting is not part
of the accelerated path.
"""
from __future__ import annotations

import numpy as np

from . import atomic
from .model import FitConfiguration, VoigtModel


class compute_cog:
    def __init__(self, lam_guess, Nlist, blist, ion: str = "auto", device_id: int = 0):
        st = atomic.lookup(lam_guess, "closest")
        self.st = {"wave": float(st["wave"]), "fval": float(st["fval"]), "gamma": float(st["gamma"]), "name": st["name"]}
        wave_val = self.st["wave"]
        config = FitConfiguration()
        config.add_system(z=0.0, ion=(self.st["name"].split()[0] if ion == "auto" else ion),
                          transitions=[wave_val], components=1)
        model = VoigtModel(config, FWHM=None)                       # no LSF for COG calculations (:164)
        self.model_compiled = model.compile(device_id=device_id)
        self.Nlist = np.array(Nlist, dtype=np.float64)
        self.blist = np.array(blist, dtype=np.float64)
        wave = np.linspace(wave_val - 5.0, wave_val + 5.0, 1000)     # set_one_absorber grid (:50)
        NN, BB = np.meshgrid(self.Nlist, self.blist, indexing="ij")
        theta = np.stack([NN.ravel(), BB.ravel(), np.zeros(NN.size)], axis=1)    # [N, b, v=0] (:53)
        # one batch of len(N) * len(b) rows; the trapezoidal integral of 1 - flux (:56-60) is formed on the GPU behind the model
        # launch (vp_model_flux_rowsum): one double per (N, b) pair crosses PCIe instead of a 1000-pixel row
        ew = self.model_compiled.equivalent_width(theta, wave)
        self.Wlist = ew.reshape(self.Nlist.size, self.blist.size)    # (:171-183)
        self.wave = wave
