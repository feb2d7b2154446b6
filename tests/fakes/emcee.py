"""Test double for the `emcee` package (absent from this image; SURVEY 3.1 describes its call shapes).

Only what rbvfit's walker loop touches (vfit_mcmc.py:408-423, 536-540): ``EnsembleSampler(nwalkers, ndim,
log_prob_fn, pool=None, vectorize=False)`` with the default red-blue StretchMove (two splits), ``run_mcmc``,
``get_chain`` / ``get_log_prob`` and ``acceptance_fraction``.  With ``vectorize=True`` the probability
function receives a whole (n, ndim) block and must return n floats: one full-ensemble call for the initial
state, then exactly TWO calls of nwalkers/2 rows per step; a NaN in the return raises
``ValueError("Probability function returned NaN")``.  Every call's shape is recorded in ``calls``."""
import numpy as np

__version__ = "3.1.0+rbvfit_amd.testdouble"


class State:
    def __init__(self, coords, log_prob):
        self.coords, self.log_prob = coords, log_prob


class EnsembleSampler:
    def __init__(self, nwalkers, ndim, log_prob_fn, pool=None, moves=None, args=None, kwargs=None, vectorize=False,
                 a=2.0, seed=None):
        if nwalkers % 2 or nwalkers < 2 * ndim:
            raise ValueError("emcee: nwalkers must be even and >= 2 * ndim")
        if pool is not None:
            raise ValueError("test double: pool is not supported (and must not be used with a HIP context)")
        self.nwalkers, self.ndim, self.log_prob_fn, self.vectorize, self.a = nwalkers, ndim, log_prob_fn, vectorize, a
        self.rng = np.random.default_rng(seed)
        self.calls = []                       # shape of every array handed to log_prob_fn
        self._chain, self._lp = [], []
        self.naccepted = np.zeros(nwalkers)
        self.iteration = 0

    def compute_log_prob(self, coords):
        if np.any(~np.isfinite(coords)):
            raise ValueError("At least one parameter value was infinite or NaN")
        if self.vectorize:
            self.calls.append(coords.shape)
            out = self.log_prob_fn(coords)
        else:
            out = []
            for row in coords:
                self.calls.append(row.shape)
                out.append(self.log_prob_fn(row))
        lp = np.array([float(v) for v in out])
        if lp.shape != (len(coords),):
            raise ValueError("log_prob_fn returned the wrong number of values")
        if np.any(np.isnan(lp)):
            raise ValueError("Probability function returned NaN")
        return lp

    def run_mcmc(self, initial_state, nsteps, progress=False, **kw):
        p = np.array(getattr(initial_state, "coords", initial_state), dtype=np.float64)
        if p.shape != (self.nwalkers, self.ndim):
            raise ValueError("incompatible input dimensions")
        lp = self.compute_log_prob(p)
        if not np.all(np.isfinite(lp)):
            raise ValueError("Initial state has a large condition number or non-finite log_prob")
        half = self.nwalkers // 2
        for _ in range(nsteps):
            idx = self.rng.permutation(self.nwalkers)           # emcee shuffles the split every step
            sets = [idx[:half], idx[half:]]
            for k in (0, 1):
                S, C = sets[k], sets[1 - k]
                zz = ((self.a - 1.0) * self.rng.random(half) + 1.0) ** 2 / self.a
                partner = p[C[self.rng.integers(0, half, half)]]
                q = partner - (partner - p[S]) * zz[:, None]
                lq = self.compute_log_prob(q)                   # ONE (W/2, ndim) call per split
                lnr = (self.ndim - 1.0) * np.log(zz) + lq - lp[S]
                acc = np.log(self.rng.random(half)) < lnr
                p[S[acc]], lp[S[acc]] = q[acc], lq[acc]
                self.naccepted[S[acc]] += 1
            self._chain.append(p.copy()); self._lp.append(lp.copy())
            self.iteration += 1
        return State(p, lp)

    @property
    def acceptance_fraction(self):
        return self.naccepted / max(self.iteration, 1)

    def get_chain(self, discard=0, flat=False, thin=1):
        c = np.array(self._chain)[discard::thin]
        return c.reshape(-1, self.ndim) if flat else c

    def get_log_prob(self, discard=0, flat=False, thin=1):
        c = np.array(self._lp)[discard::thin]
        return c.reshape(-1) if flat else c
