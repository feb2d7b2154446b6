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


class EnsembleSliceSampler:
    """Ensemble slice sampling with the differential move (Karamanis, Beutler & Peacock 2021) -- the
    algorithm of ``zeus.EnsembleSampler``, which the reference offers as ``sampler='zeus'``
    (vfit_mcmc.py:425-440) and which is absent from this image.  Fully batched: every stepping-out
    or shrinking round is ONE lnprob call over the walkers still active in it, so the batches are
    ragged (W/2, then fewer and fewer rows) -- the call shape SURVEY 3.1 describes for zeus.

    Per iteration the ensemble is split at random into two halves; each walker of the active half
    slices along eta = mu * 2.38/sqrt(2 D) * (X_l - X_m), l != m drawn from the other half.  ``mu``
    is tuned during ``tune_steps`` iterations towards equal numbers of expansions and contractions.
    ``lnprob`` maps (n, D) -> (n,)."""

    def __init__(self, nwalkers: int, ndim: int, lnprob: Callable, mu: float = 1.0, maxsteps: int = 10000,
                 tune: bool = True, tolerance: float = 0.05, patience: int = 5, seed: Optional[int] = None):
        if nwalkers % 2 or nwalkers < 2 * ndim:
            raise ValueError("nwalkers must be even and at least 2*ndim (as zeus requires)")
        self.nwalkers, self.ndim, self.lnprob = nwalkers, ndim, lnprob
        self.mu, self.maxsteps, self.tune = float(mu), int(maxsteps), bool(tune)
        self.tolerance, self.patience, self._good = float(tolerance), int(patience), 0
        self.rng = np.random.default_rng(seed)
        self.chain = None
        self.lnprobability = None
        self.nsteps = 0
        self.n_lnprob_calls = 0
        self.n_lnprob_evals = 0
        self.batch_sizes = []                       # rows per lnprob call (diagnostic: the ragged shapes)
        self.mu_history = []

    def _eval(self, pos):
        lp = np.asarray(self.lnprob(pos), dtype=np.float64)
        self.n_lnprob_calls += 1
        self.n_lnprob_evals += len(pos)
        self.batch_sizes.append(len(pos))
        if np.any(np.isnan(lp)):
            raise ValueError("Probability function returned NaN")
        return lp

    def run_mcmc(self, p0, nsteps: int, lnprob0=None):
        pos = np.array(p0, dtype=np.float64)
        if pos.shape != (self.nwalkers, self.ndim):
            raise ValueError(f"initial state must have shape ({self.nwalkers}, {self.ndim})")
        lp = self._eval(pos) if lnprob0 is None else np.array(lnprob0, dtype=np.float64)
        if not np.all(np.isfinite(lp)):
            raise ValueError("initial walkers must have finite lnprob")
        chain = np.empty((nsteps, self.nwalkers, self.ndim))
        lnps = np.empty((nsteps, self.nwalkers))
        half = self.nwalkers // 2
        gamma0 = 2.38 / np.sqrt(2.0 * self.ndim)
        rng = self.rng
        for it in range(nsteps):
            perm = rng.permutation(self.nwalkers)
            nexp = ncon = 0
            for S, C in ((perm[:half], perm[half:]), (perm[half:], perm[:half])):
                n = S.size
                # two distinct partners from the complementary half
                l = rng.integers(0, C.size, n)
                m = (l + 1 + rng.integers(0, C.size - 1, n)) % C.size
                eta = self.mu * gamma0 * (pos[C[l]] - pos[C[m]])
                X = pos[S]
                Z0 = lp[S] - rng.exponential(size=n)
                L = -rng.random(n)
                R = L + 1.0
                J = np.floor(self.maxsteps * rng.random(n)).astype(np.int64)
                K = (self.maxsteps - 1) - J
                # stepping out, left then right: one batch per round over the walkers still expanding
                for edge, budget, sign in ((L, J, -1.0), (R, K, 1.0)):
                    act = np.arange(n)
                    while act.size:
                        act = act[budget[act] > 0]
                        if not act.size:
                            break
                        out = self._eval(X[act] + edge[act, None] * eta[act]) > Z0[act]
                        act = act[out]
                        edge[act] += sign
                        budget[act] -= 1
                        nexp += act.size
                # shrinking
                Xn, lpn = X.copy(), lp[S].copy()
                act = np.arange(n)
                while act.size:
                    Wd = L[act] + rng.random(act.size) * (R[act] - L[act])
                    prop = X[act] + Wd[:, None] * eta[act]
                    lpp = self._eval(prop)
                    ok = lpp > Z0[act]
                    Xn[act[ok]], lpn[act[ok]] = prop[ok], lpp[ok]
                    rej = act[~ok]
                    wr = Wd[~ok]
                    L[rej] = np.where(wr < 0, wr, L[rej])
                    R[rej] = np.where(wr < 0, R[rej], wr)
                    ncon += rej.size
                    act = rej
                pos[S], lp[S] = Xn, lpn
            if self.tune:
                ratio = 2.0 * max(1, nexp) / (max(1, nexp) + ncon)
                self.mu *= ratio
                self._good = self._good + 1 if abs(ratio - 1.0) < self.tolerance else 0
                if self._good >= self.patience:
                    self.tune = False
            self.mu_history.append(self.mu)
            chain[it], lnps[it] = pos, lp
        self.chain = chain if self.chain is None else np.concatenate([self.chain, chain])
        self.lnprobability = lnps if self.lnprobability is None else np.concatenate([self.lnprobability, lnps])
        self.nsteps += nsteps
        return pos, lp

    def get_chain(self, discard: int = 0, flat: bool = False):
        c = self.chain[discard:]
        return c.reshape(-1, self.ndim) if flat else c


class DeviceSliceSampler:
    """Ensemble slice sampling with the whole walker loop on the GPU (``vp_slice_run``,
    csrc/slice_kernels.h): the same move as ``EnsembleSliceSampler`` / zeus, but every stepping-out or
    shrinking round is a lnprob batch enqueued from the host without looking at its result -- the
    active walkers are compacted in HBM, and the host reads one word per group of rounds.  Draws are
    Philox4x32-10 keyed by (seed, step, half, walker, purpose); successive ``run_mcmc`` calls continue
    the stream and the tuned ``mu``.  ``engine`` is the ``rbvfit_amd.Engine`` (``vfit.engine``)."""

    def __init__(self, nwalkers: int, ndim: int, engine, mu: float = 1.0, maxsteps: int = 10000, tune: bool = True,
                 tolerance: float = 0.05, patience: int = 5, seed: Optional[int] = None):
        if nwalkers % 2 or nwalkers < 2 * ndim:
            raise ValueError("nwalkers must be even and at least 2*ndim (as zeus requires)")
        self.nwalkers, self.ndim, self.engine = nwalkers, ndim, engine
        self.mu, self.maxsteps, self.tune = float(mu), int(maxsteps), bool(tune)
        self.tolerance, self.patience = float(tolerance), int(patience)
        self.seed = int(np.random.SeedSequence(seed).generate_state(1, dtype=np.uint64)[0])
        self._tune_state = 1                  # 1 + consecutive in-tolerance iterations: carried from call to call
        self.chain = None
        self.lnprobability = None
        self.nsteps = 0
        self.n_lnprob_evals = 0
        self.mu_history = []

    def run_mcmc(self, p0, nsteps: int, lnprob0=None, store: bool = True):
        pos = np.array(p0, dtype=np.float64)
        if pos.shape != (self.nwalkers, self.ndim):
            raise ValueError(f"initial state must have shape ({self.nwalkers}, {self.ndim})")
        r = self.engine.slice_run(pos, nsteps, lnprob=lnprob0, mu=self.mu, tune=self._tune_state if self.tune else 0, tolerance=self.tolerance,
                                  patience=self.patience, maxsteps=self.maxsteps, seed=self.seed, step0=self.nsteps,
                                  store_chain=store)
        self.mu, self.tune, self._tune_state = r["mu"], r["tune"], r["tune_state"]
        self.mu_history.extend(r["mu_history"].tolist())
        self.n_lnprob_evals += r["n_evals"]
        if store:
            self.chain = r["chain"] if self.chain is None else np.concatenate([self.chain, r["chain"]])
            self.lnprobability = (r["chain_lnprob"] if self.lnprobability is None
                                  else np.concatenate([self.lnprobability, r["chain_lnprob"]]))
        self.nsteps += nsteps
        return r["pos"], r["lnprob"]

    def get_chain(self, discard: int = 0, flat: bool = False):
        c = self.chain[discard:]
        return c.reshape(-1, self.ndim) if flat else c


def gelman_rubin(chain) -> np.ndarray:
    """Potential scale reduction R-hat per parameter for a (nsteps, nwalkers, D) chain, walkers as the
    parallel chains -- the diagnostic the reference prints for zeus runs (vfit_mcmc.py:633-640)."""
    c = np.asarray(chain, dtype=np.float64)
    n, m = c.shape[0], c.shape[1]
    means = c.mean(axis=0)                                  # (m, D)
    W = c.var(axis=0, ddof=1).mean(axis=0)
    B = n * means.var(axis=0, ddof=1)
    var = (n - 1) / n * W + B / n
    return np.sqrt(var / W)
