"""Test double for the `zeus` package (zeus-mcmc; absent from this image).

``EnsembleSampler(nwalkers, ndim, logprob_fn, vectorize=False, verbose=True)`` doing ensemble slice sampling
with the differential move: per step each half-ensemble runs stepping-out and shrinking loops that evaluate
ONLY the still-active walkers, so with ``vectorize=True`` the probability function sees ragged (n, ndim)
blocks, 1 <= n <= nwalkers/2, several per step.  Shapes are recorded in ``calls``."""
import numpy as np

__version__ = "2.5.4+rbvfit_amd.testdouble"


class EnsembleSampler:
    def __init__(self, nwalkers, ndim, logprob_fn, args=None, kwargs=None, pool=None, vectorize=False, verbose=True,
                 mu=1.0, maxsteps=10000, seed=None):
        if nwalkers % 2 or nwalkers < 2 * ndim:
            raise ValueError("zeus: nwalkers must be even and >= 2 * ndim")
        if pool is not None:
            raise ValueError("test double: pool is not supported")
        self.nwalkers, self.ndim, self.fn, self.vectorize = nwalkers, ndim, logprob_fn, vectorize
        self.mu, self.maxsteps = mu, maxsteps
        self.rng = np.random.default_rng(seed)
        self.calls, self._chain, self._lp = [], [], []
        self.neval = 0

    def _lnp(self, X):
        X = np.ascontiguousarray(X)
        self.neval += len(X)
        if self.vectorize:
            self.calls.append(X.shape)
            out = np.asarray(self.fn(X), dtype=np.float64)
        else:
            self.calls.extend([r.shape for r in X])
            out = np.array([float(self.fn(r)) for r in X])
        if out.shape != (len(X),):
            raise ValueError("logprob_fn returned the wrong number of values")
        if np.any(np.isnan(out)):
            raise ValueError("Log Probability returned NaN")
        return out

    def run_mcmc(self, start, nsteps, **kw):
        X = np.array(start, dtype=np.float64)
        if X.shape != (self.nwalkers, self.ndim):
            raise ValueError("Incompatible input dimensions")
        Z = self._lnp(X)
        if not np.all(np.isfinite(Z)):
            raise ValueError("Invalid walker initial positions")
        half, rng = self.nwalkers // 2, self.rng
        g0 = 2.38 / np.sqrt(2 * self.ndim)
        for _ in range(nsteps):
            perm = rng.permutation(self.nwalkers)
            nexp = ncon = 0
            for S, C in ((perm[:half], perm[half:]), (perm[half:], perm[:half])):
                pairs = np.array([rng.choice(half, 2, replace=False) for _ in range(half)])
                eta = self.mu * g0 * (X[C[pairs[:, 0]]] - X[C[pairs[:, 1]]])
                X0, Z0 = X[S].copy(), Z[S] - rng.exponential(size=half)
                L = -rng.random(half); R = L + 1.0
                J = np.floor(self.maxsteps * rng.random(half)).astype(int); K = self.maxsteps - 1 - J
                for edge, budget, sgn in ((L, J, -1.0), (R, K, 1.0)):       # stepping out: only walkers still expanding
                    act = np.arange(half)[budget > 0]
                    while act.size:
                        out = self._lnp(X0[act] + edge[act, None] * eta[act]) > Z0[act]
                        act = act[out]
                        edge[act] += sgn; budget[act] -= 1; nexp += act.size
                        act = act[budget[act] > 0]
                act = np.arange(half)                                    # shrinking: only walkers not yet accepted
                while act.size:
                    Wd = L[act] + rng.random(act.size) * (R[act] - L[act])
                    Y = X0[act] + Wd[:, None] * eta[act]
                    ZY = self._lnp(Y)
                    ok = ZY > Z0[act]
                    X[S[act[ok]]], Z[S[act[ok]]] = Y[ok], ZY[ok]
                    rej, wr = act[~ok], Wd[~ok]
                    L[rej] = np.where(wr < 0, wr, L[rej]); R[rej] = np.where(wr < 0, R[rej], wr)
                    ncon += rej.size
                    act = rej
            self.mu *= 2.0 * max(1, nexp) / (max(1, nexp) + ncon)
            self._chain.append(X.copy()); self._lp.append(Z.copy())

    def get_chain(self, discard=0, flat=False, thin=1):
        c = np.array(self._chain)[discard::thin]
        return c.reshape(-1, self.ndim) if flat else c

    def get_log_prob(self, discard=0, flat=False, thin=1):
        c = np.array(self._lp)[discard::thin]
        return c.reshape(-1) if flat else c
