"""Host-side ensemble driver with fully batched proposals.

emcee / zeus are not vendored by the reference and are absent from this image, so this module
provides the walker loop the reference delegates to them (vfit_mcmc.py:408-440, 536-540): the
affine-invariant stretch move of Goodman & Weare (2010) in emcee's red-blue form -- the ensemble is
split in two halves and each half is updated in ONE batched lnprob call against the other half.
With emcee installed, ``EnsembleSampler(nwalkers, ndim, fitter.lnprob, vectorize=True)`` works the
same way; this driver only removes the dependency.

Also mirrors ``vfit._initialize_walkers`` (vfit_mcmc.py:442-466).
"""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np


def initialize_walkers(theta, lb, ub, nwalkers: int, perturbation: float, lnprob: Callable, rng,
                       max_attempts: int = 1000) -> np.ndarray:
    """theta + perturbation * N(0,1), clipped to (lb+1e-10, ub-1e-10); rows whose lnprob is not
    finite are redrawn (batched) up to ``max_attempts`` times."""
    theta, lb, ub = (np.asarray(a, dtype=np.float64) for a in (theta, lb, ub))
    pos = np.clip(theta + perturbation * rng.standard_normal((nwalkers, theta.size)), lb + 1e-10, ub - 1e-10)
    lp = np.asarray(lnprob(pos), dtype=np.float64)
    for _ in range(max_attempts):
        bad = ~np.isfinite(lp)
        if not bad.any():
            return pos
        pos[bad] = np.clip(theta + perturbation * rng.standard_normal((bad.sum(), theta.size)), lb + 1e-10, ub - 1e-10)
        lp[bad] = lnprob(pos[bad])
    raise RuntimeError(f"Could not initialize {int((~np.isfinite(lp)).sum())} walkers after {max_attempts} attempts")


class StretchMoveSampler:
    """``lnprob`` must accept a (n, D) array and return (n,) values (-inf allowed, NaN rejected like
    emcee does).  ``chain`` has shape (nsteps, nwalkers, D) and ``lnprobability`` (nsteps, nwalkers)."""

    def __init__(self, nwalkers: int, ndim: int, lnprob: Callable, a: float = 2.0, seed: Optional[int] = None):
        if nwalkers % 2 or nwalkers < 2 * ndim:
            raise ValueError("nwalkers must be even and at least 2*ndim (as emcee requires)")
        self.nwalkers, self.ndim, self.lnprob, self.a = nwalkers, ndim, lnprob, float(a)
        self.rng = np.random.default_rng(seed)
        self.chain = None
        self.lnprobability = None
        self.naccepted = np.zeros(nwalkers, dtype=np.int64)
        self.nsteps = 0
        self.n_lnprob_calls = 0

    def _eval(self, pos):
        lp = np.asarray(self.lnprob(pos), dtype=np.float64)
        self.n_lnprob_calls += 1
        if np.any(np.isnan(lp)):
            raise ValueError("Probability function returned NaN")
        return lp

    def run_mcmc(self, p0, nsteps: int, lnprob0=None):
        pos = np.array(p0, dtype=np.float64)
        if pos.shape != (self.nwalkers, self.ndim):
            raise ValueError(f"initial state must have shape ({self.nwalkers}, {self.ndim})")
        lp = self._eval(pos) if lnprob0 is None else np.array(lnprob0, dtype=np.float64)
        chain = np.empty((nsteps, self.nwalkers, self.ndim))
        lnps = np.empty((nsteps, self.nwalkers))
        half = self.nwalkers // 2
        halves = (np.arange(half), np.arange(half, self.nwalkers))
        for it in range(nsteps):
            for k in (0, 1):
                S, C = halves[k], halves[1 - k]
                n = S.size
                zz = ((self.a - 1.0) * self.rng.random(n) + 1.0) ** 2 / self.a        # g(z) ~ 1/sqrt(z) on [1/a, a]
                partners = pos[C[self.rng.integers(0, C.size, n)]]
                prop = partners - (partners - pos[S]) * zz[:, None]
                lp_new = self._eval(prop)                                            # ONE batched call
                lnq = (self.ndim - 1.0) * np.log(zz) + lp_new - lp[S]
                accept = np.log(self.rng.random(n)) < lnq
                pos[S[accept]] = prop[accept]
                lp[S[accept]] = lp_new[accept]
                self.naccepted[S[accept]] += 1
            chain[it], lnps[it] = pos, lp
        self.chain = chain if self.chain is None else np.concatenate([self.chain, chain])
        self.lnprobability = lnps if self.lnprobability is None else np.concatenate([self.lnprobability, lnps])
        self.nsteps += nsteps
        return pos, lp

    @property
    def acceptance_fraction(self):
        return self.naccepted / max(self.nsteps, 1)

    def get_chain(self, discard: int = 0, flat: bool = False):
        c = self.chain[discard:]
        return c.reshape(-1, self.ndim) if flat else c


class DeviceStretchSampler:
    """Same interface as ``StretchMoveSampler``, but the whole loop runs on the GPU through
    ``vp_stretch_run`` (include/rbvfit_amd.h): positions, lnprob, proposals and accept/reject stay
    in HBM, one lnprob batch per half-ensemble, Philox4x32-10 draws keyed by (seed, step, half,
    walker) -- successive ``run_mcmc`` calls continue the same stream.  ``engine`` is the
    ``rbvfit_amd.Engine`` that holds the bounds and instruments (``vfit.engine``)."""

    def __init__(self, nwalkers: int, ndim: int, engine, a: float = 2.0, seed: Optional[int] = None):
        if nwalkers % 2 or nwalkers < 2 * ndim:
            raise ValueError("nwalkers must be even and at least 2*ndim (as emcee requires)")
        self.nwalkers, self.ndim, self.engine, self.a = nwalkers, ndim, engine, float(a)
        self.seed = int(np.random.SeedSequence(seed).generate_state(1, dtype=np.uint64)[0])
        self.chain = None
        self.lnprobability = None
        self.naccepted = np.zeros(nwalkers, dtype=np.int64)
        self.nsteps = 0

    def run_mcmc(self, p0, nsteps: int, lnprob0=None, store: bool = True):
        pos = np.array(p0, dtype=np.float64)
        if pos.shape != (self.nwalkers, self.ndim):
            raise ValueError(f"initial state must have shape ({self.nwalkers}, {self.ndim})")
        pos, lp, chain, lnps, self.naccepted = self.engine.stretch_run(
            pos, nsteps, lnprob=lnprob0, a=self.a, seed=self.seed, step0=self.nsteps, store_chain=store,
            naccepted=self.naccepted)
        if store:
            self.chain = chain if self.chain is None else np.concatenate([self.chain, chain])
            self.lnprobability = lnps if self.lnprobability is None else np.concatenate([self.lnprobability, lnps])
        self.nsteps += nsteps
        return pos, lp

    @property
    def acceptance_fraction(self):
        return self.naccepted / max(self.nsteps, 1)

    def get_chain(self, discard: int = 0, flat: bool = False):
        c = self.chain[discard:]
        return c.reshape(-1, self.ndim) if flat else c
