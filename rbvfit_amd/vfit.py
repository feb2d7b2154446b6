"""GPU-backed mirror of the reference's fitter for the posterior part only.

Mirrors ``vfit.__init__`` (validation + _compile_models, vfit_mcmc.py:127-259), ``lnprior``
(:291-295), ``lnlike`` (:297-319), ``lnprob`` (:348-353) and ``set_bounds`` (non-ion branch,
:759-787).  The walker loop stays on the host (emcee/zeus with ``vectorize=True`` or
``rbvfit_amd.sampler``); every call evaluates the whole theta batch in one GPU pass.
"""
from __future__ import annotations

from typing import Dict

import numpy as np

from .engine import Engine
from .model import CompiledModelData, CompiledVoigtModel, VoigtModel, tables_from_rbvfit


def set_bounds(nguess, bguess, vguess, **kwargs):
    """Traditional bounds of the reference (vfit_mcmc.py:759-787): N +/- 2, b -/+ 40 clipped to
    [2, 150], v +/- 50; keyword overrides Nlow/blow/vlow/Nhi/bhi/vhi."""
    nguess, bguess, vguess = (np.asarray(a, dtype=np.float64) for a in (nguess, bguess, vguess))
    if kwargs.get("ions") is not None:
        raise NotImplementedError("ion-aware bounds are outside the accelerated path (and broken in the "
                                  "reference, SURVEY T16); pass explicit Nlow/Nhi/... overrides instead")
    Nlow, NHI = nguess - 2.0, nguess + 2.0
    blow, bHI = np.clip(bguess - 40.0, 2.0, None), np.clip(bguess + 40.0, None, 150.0)
    vlow, vHI = vguess - 50.0, vguess + 50.0
    Nlow = np.asarray(kwargs.get("Nlow", Nlow)); blow = np.asarray(kwargs.get("blow", blow))
    vlow = np.asarray(kwargs.get("vlow", vlow)); NHI = np.asarray(kwargs.get("Nhi", NHI))
    bHI = np.asarray(kwargs.get("bhi", bHI)); vHI = np.asarray(kwargs.get("vhi", vHI))
    lb = np.concatenate([Nlow, blow, vlow])
    ub = np.concatenate([NHI, bHI, vHI])
    return [lb, ub], lb, ub


def _tables_of(model) -> CompiledModelData:
    if isinstance(model, CompiledModelData):
        return model
    if isinstance(model, CompiledVoigtModel):
        return model.data
    if isinstance(model, VoigtModel):
        return model.compile().data
    if hasattr(model, "config") and hasattr(model, "compile"):      # a reference rbvfit VoigtModel
        return tables_from_rbvfit(model.compile(verbose=False))
    if hasattr(model, "data") or hasattr(model, "atomic_lambda0"):  # reference CompiledVoigtModel / data
        return tables_from_rbvfit(model)
    raise TypeError("instrument 'model' must be a VoigtModel / CompiledVoigtModel / CompiledModelData "
                    "(rbvfit_amd or rbvfit) or a callable model(theta, wave) -> flux")


def _is_host_callable(model) -> bool:
    """The reference's pass-through branch (vfit_mcmc.py:242-248): anything without ``.config`` / ``.compile`` is
    stored as the model function itself and called as ``model(theta, wave)`` (:304).  Here: a plain callable that
    carries no tables the GPU path could take (functions, lambdas, bound methods, functools.partial, objects with
    ``__call__``) -- evaluated on the host, row by row, exactly like that."""
    if isinstance(model, (CompiledModelData, CompiledVoigtModel, VoigtModel)):
        return False
    if hasattr(model, "config") and hasattr(model, "compile"):
        return False
    if hasattr(model, "data") or hasattr(model, "atomic_lambda0"):
        return False
    owner = getattr(model, "__self__", None)               # a bound model_flux of a compiled model: take its tables
    if owner is not None and (isinstance(owner, CompiledVoigtModel) or hasattr(owner, "data") or hasattr(owner, "atomic_lambda0")):
        return False
    return callable(model)


def _unwrap_bound(model):
    """``compiled.model_flux`` (what the reference's ``_compile_models`` itself stores) -> the compiled model."""
    owner = getattr(model, "__self__", None)
    if owner is not None and not isinstance(model, (CompiledModelData, CompiledVoigtModel, VoigtModel)) and \
            (isinstance(owner, CompiledVoigtModel) or hasattr(owner, "data") or hasattr(owner, "atomic_lambda0")):
        return owner
    return model


class vfit:
    def __init__(self, instrument_data: Dict, theta, lb, ub, no_of_Chain=50, no_of_steps=1000,
                 perturbation=1e-4, sampler="emcee", skip_initial_state_check=False, device_id: int = 0):
        self._validate_unified_instrument_data(instrument_data)
        self._validate_guesses(theta, lb, ub)
        self.theta = np.asarray(theta, dtype=np.float64)
        self.lb = np.asarray(lb, dtype=np.float64)
        self.ub = np.asarray(ub, dtype=np.float64)
        self.no_of_Chain = self.nwalkers = no_of_Chain
        self.no_of_steps = no_of_steps
        self.perturbation = perturbation
        self.skip_initial_state_check = skip_initial_state_check
        self.sampler_name = sampler.lower()
        if self.sampler_name not in ("emcee", "zeus"):
            raise ValueError(f"Unknown sampler '{sampler}'. Use 'emcee' or 'zeus'.")
        self.ndim = len(self.theta)
        self.multi_instrument = len(instrument_data) > 1
        self.engine = Engine(device_id)
        self.engine.set_bounds(self.lb, self.ub)
        self.instrument_data = {}
        self._host_instruments = []        # entries whose model is a user callable (vfit_mcmc.py:242-248): evaluated on the host
        self._n_gpu = 0
        for name, data in instrument_data.items():
            error = np.asarray(data["error"])
            entry = {
                "wave": np.asarray(data["wave"]),
                "flux": np.asarray(data["flux"]),
                "error": error,
                "inv_sigma2": 1.0 / (error ** 2),                  # vfit_mcmc.py:255 (dtype of error, T4)
                "log_inv_sigma2": np.log(1.0 / (error ** 2)),      # :256
            }
            model = _unwrap_bound(data["model"])
            if _is_host_callable(model):
                # the reference takes ANY callable (theta(D,), wave(P,)) -> flux(P,) verbatim; it cannot run on the GPU, so
                # its likelihood term is formed on the host per row and added to the GPU instruments' (lnprob below)
                entry["model"] = model
                entry["index"] = None
                self._host_instruments.append(entry)
            else:
                tables = _tables_of(model)
                if 3 * tables.total_components != self.ndim:
                    raise ValueError(f"instrument '{name}': model has {3 * tables.total_components} parameters, "
                                     f"theta has {self.ndim}")
                entry["tables"] = tables
                entry["index"] = self.engine.add_instrument(entry["wave"], entry["flux"], entry["inv_sigma2"],
                                                            entry["log_inv_sigma2"], **tables.engine_kwargs())
                self._n_gpu += 1
            self.instrument_data[name] = entry

    # -- validation (vfit_mcmc.py:199-229) -----------------------------------------------------
    @staticmethod
    def _validate_unified_instrument_data(instrument_data):
        if not isinstance(instrument_data, dict):
            raise TypeError("instrument_data must be a dictionary")
        if len(instrument_data) == 0:
            raise ValueError("instrument_data cannot be empty")
        required = {"model", "wave", "flux", "error"}
        for name, data in instrument_data.items():
            if not isinstance(data, dict):
                raise TypeError(f"instrument_data['{name}'] must be a dictionary")
            missing = required - set(data.keys())
            if missing:
                raise ValueError(f"instrument_data['{name}'] missing keys: {missing}")
            n = len(data["wave"])
            if len(data["flux"]) != n or len(data["error"]) != n:
                raise ValueError(f"instrument_data['{name}']: wave, flux, and error must have same length")

    @staticmethod
    def _validate_guesses(theta, lb, ub):
        theta, lb, ub = np.asarray(theta), np.asarray(lb), np.asarray(ub)
        if len(theta) != len(lb) or len(theta) != len(ub):
            raise ValueError("theta, lb, and ub must have the same length")
        if np.any(theta < lb) or np.any(theta > ub):
            raise ValueError("Initial guess theta must be within bounds lb and ub")

    # -- posterior ----------------------------------------------------------------------------
    def lnprior(self, theta):
        """(D,) -> float or (W, D) -> (W,): 0 inside the box, -inf outside (bounds inclusive)."""
        th = np.asarray(theta, dtype=np.float64)
        oob = np.any(th < self.lb, axis=-1) | np.any(th > self.ub, axis=-1)
        out = np.where(oob, -np.inf, 0.0)
        return float(out) if th.ndim == 1 else out

    def _host_lnlike_rows(self, rows):
        """Sum over the user-callable instruments of -0.5 sum((flux - model(theta, wave))**2 w - log w) for each row
        (vfit_mcmc.py:302-313, the callable invoked as ``data['model'](theta, data['wave'])``, :304).  ANY exception
        while a row is evaluated makes that row's likelihood -inf, as the reference's ``except Exception`` does
        (:317-319); NaN is not caught and propagates."""
        out = np.zeros(len(rows), dtype=np.float64)
        for i, th in enumerate(rows):
            try:
                total = 0.0
                for entry in self._host_instruments:
                    model_dat = entry["model"](th, entry["wave"])
                    total += -0.5 * np.sum((entry["flux"] - model_dat) ** 2 * entry["inv_sigma2"] - entry["log_inv_sigma2"])
                out[i] = total
            except Exception:
                out[i] = -np.inf
        return out

    def _prior_only(self, th2):
        return np.atleast_1d(self.lnprior(th2)).astype(np.float64)

    def lnprob(self, theta):
        """(D,) -> float, (W, D) -> (W,) float64: one GPU pass for the whole batch; instruments given as plain
        callables (the reference's pass-through branch) add their term on the host, for the rows inside the prior box
        only -- the reference does not evaluate any model for a row outside it (vfit_mcmc.py:350-351)."""
        th = np.asarray(theta, dtype=np.float64)
        if not self._host_instruments:
            out = self.engine.lnprob(th)
            return float(out[0]) if th.ndim == 1 else out
        th2 = np.atleast_2d(th)
        out = np.array(self.engine.lnprob(th2) if self._n_gpu else self._prior_only(th2), dtype=np.float64)
        inside = ~np.isneginf(self._prior_only(th2))
        if np.any(inside):
            out[inside] = out[inside] + self._host_lnlike_rows(th2[inside])
        return float(out[0]) if th.ndim == 1 else out

    def lnlike(self, theta):
        """Likelihood without the prior.  The engine fuses prior and likelihood; rows outside the
        box are evaluated here through a context-free detour: lnlike = lnprob where the prior is 0,
        and the model is evaluated explicitly for out-of-bounds rows."""
        th = np.atleast_2d(np.asarray(theta, dtype=np.float64))
        out = np.array(self.engine.lnprob(th), dtype=np.float64) if self._n_gpu else np.zeros(len(th))
        oob = np.isneginf(self._prior_only(th))
        if np.any(oob) and self._n_gpu:
            rows = th[oob]
            total = np.zeros(len(rows))
            for entry in self.instrument_data.values():
                if entry["index"] is None:
                    continue
                model = self.engine.model_flux(entry["index"], rows)
                total += -0.5 * np.sum((entry["flux"] - model) ** 2 * entry["inv_sigma2"]
                                       - entry["log_inv_sigma2"], axis=1)
            out[oob] = total
        if self._host_instruments:
            out = out + self._host_lnlike_rows(th)
        return float(out[0]) if np.asarray(theta).ndim == 1 else out

    __call__ = lnprob

    def close(self):
        self.engine.close()

    # -- batched finite-difference stencils -----------------------------------------
    def lnprob_and_grad(self, theta, eps: float = 1e-8):
        """lnprob(theta) and its forward-difference gradient from ONE batch of D+1 rows (the
        stencil scipy's L-BFGS-B builds serially for the reference, vfit_mcmc.py:355-360).  Steps
        that would leave the box are taken backwards."""
        theta = np.asarray(theta, dtype=np.float64)
        D = theta.size
        h = np.where(theta + eps > self.ub, -eps, eps)
        batch = np.vstack([theta[None, :], theta[None, :] + np.diag(h)])
        lp = self.lnprob(batch)
        return float(lp[0]), (lp[1:] - lp[0]) / h

    def optimize_guess(self, theta, eps: float = 1e-8):
        """Mirror of ``vfit.optimize_guess`` (vfit_mcmc.py:355-360): L-BFGS-B on -lnprob inside the
        bounds, with the finite-difference gradient evaluated as one GPU batch per iteration."""
        import scipy.optimize as op

        def nll(th):
            f, g = self.lnprob_and_grad(th, eps)
            if not np.isfinite(f):
                return np.inf, np.zeros_like(th)
            return -f, -g

        res = op.minimize(nll, np.asarray(theta, dtype=np.float64), jac=True, method="L-BFGS-B",
                          bounds=list(zip(self.lb, self.ub)))
        return res.x

    # -- quick fit --------------------------------------------------------------
    def chi2(self, theta):
        """Sum over instruments of sum(((flux - model)/error)**2) -- the objective of the reference's
        quick fit (quick_fit_interface.py:30-53).  (D,) -> float, (W, D) -> (W,); no prior (rows outside
        the box are evaluated too).  Obtained from the fused likelihood: chi2 = -2 lnlike + sum log w."""
        th = np.asarray(theta, dtype=np.float64)
        const = sum(float(np.sum(np.asarray(e["log_inv_sigma2"], dtype=np.float64))) for e in self.instrument_data.values())
        out = -2.0 * np.atleast_1d(self.lnlike(np.atleast_2d(th))) + const
        return float(out[0]) if th.ndim == 1 else out

    def estimate_parameter_errors(self, theta_best, theta_initial=None, delta_frac: float = 0.01):
        """Mirror of ``_estimate_parameter_errors`` (quick_fit_interface.py:87-128): curvature of chi2
        along each axis from central differences, sigma = 1/sqrt(d2chi2) -- the 2D+1 evaluations are
        ONE batch.  Non-positive curvature falls back to |theta_best - theta_initial|."""
        tb = np.asarray(theta_best, dtype=np.float64)
        ti = np.asarray(self.theta if theta_initial is None else theta_initial, dtype=np.float64)
        D = tb.size
        delta = np.maximum(np.maximum(np.abs(tb) * delta_frac, np.abs(ti) * delta_frac), 1e-6)
        batch = np.vstack([tb[None, :], tb[None, :] + np.diag(delta), tb[None, :] - np.diag(delta)])
        c = self.chi2(batch)
        d2 = (c[1:D + 1] - 2.0 * c[0] + c[D + 1:]) / delta ** 2
        with np.errstate(divide="ignore", invalid="ignore"):
            err = np.where(d2 > 0, np.sqrt(1.0 / d2), np.abs(tb - ti))
        return err

    def fit_quick(self, verbose: bool = False, eps: float = 1e-8):
        """Mirror of ``vfit.fit_quick`` (vfit_mcmc.py:362-406 -> quick_fit_interface.py:10-84):
        L-BFGS-B on chi2 inside the bounds (``maxfun=5000``), then curvature errors.  The
        finite-difference gradient scipy would build serially is one (D+1)-row GPU batch per
        iteration.  Returns (theta_best, theta_best_error) and stores them on the object."""
        import warnings
        import scipy.optimize as op
        self.mcmc_flag = False
        lb, ub = self.lb, self.ub

        def objective(th):
            h = np.where(th + eps > ub, -eps, eps)
            c = self.chi2(np.vstack([th[None, :], th[None, :] + np.diag(h)]))
            if not np.all(np.isfinite(c)):
                return 1e10, np.zeros_like(th)               # quick_fit_interface.py:51-53
            return float(c[0]), (c[1:] - c[0]) / h

        try:
            res = op.minimize(objective, np.asarray(self.theta, dtype=np.float64), jac=True, method="L-BFGS-B",
                              bounds=list(zip(lb, ub)), options={"maxfun": 5000})
            theta_best = res.x
            theta_err = self.estimate_parameter_errors(theta_best, self.theta)
            if not res.success:
                warnings.warn(f"Optimization may not have converged: {res.message}")
        except Exception as e:                                # quick_fit_interface.py:79-82
            warnings.warn(f"Minimize fitting failed: {e}")
            theta_best = np.array(self.theta, dtype=np.float64)
            theta_err = np.zeros_like(theta_best)
        self.theta_best, self.theta_best_error = theta_best, theta_err
        return theta_best, theta_err

    # -- walker loop (host) ----------------------------------------------------------------------
    def runmcmc(self, optimize: bool = False, verbose: bool = False, use_pool: bool = False, seed=None,
                sampler: str = "auto"):
        """Mirror of ``vfit.runmcmc`` (vfit_mcmc.py:492-561) reduced to the sampling itself: walker
        initialisation (:442-466) and ``no_of_steps`` ensemble steps with ONE batched GPU lnprob
        call per half-ensemble.  ``sampler``: 'emcee' (``vectorize=True``), 'host'
        (``rbvfit_amd.sampler.StretchMoveSampler``), 'device' (``DeviceStretchSampler``: the whole
        loop in HBM), 'zeus' / 'host-slice' (ensemble slice sampling: zeus itself or
        ``EnsembleSliceSampler``, ragged batches), 'device-slice' (``DeviceSliceSampler``: slice sampling
        with the ragged active sets compacted on the GPU) or 'auto' (the constructor's ``sampler=``
        choice: emcee/zeus when installed, else the matching host driver).  ``use_pool`` must stay False: a HIP
        context cannot be shared with forked workers, and batching replaces the Pool."""
        if use_pool:
            raise ValueError("use_pool=True is not supported: the batched GPU lnprob replaces the fork Pool")
        if optimize:                                   # vfit_mcmc.py:507-512
            self.theta = self.optimize_guess(self.theta)
        from .sampler import StretchMoveSampler, initialize_walkers
        rng = np.random.default_rng(seed)
        guesses = initialize_walkers(self.theta, self.lb, self.ub, self.no_of_Chain, self.perturbation,
                                     self.lnprob, rng)
        if sampler not in ("auto", "emcee", "zeus", "host", "host-slice", "device", "device-slice"):
            raise ValueError("sampler must be 'auto', 'emcee', 'zeus', 'host', 'host-slice', 'device' or 'device-slice'")
        if sampler == "auto":                          # the constructor's sampler= choice, as in the reference
            try:
                if self.sampler_name == "zeus":
                    import zeus  # noqa: F401
                    sampler = "zeus"
                else:
                    import emcee  # noqa: F401
                    sampler = "emcee"
            except ImportError:
                sampler = "host-slice" if self.sampler_name == "zeus" else "host"
        if sampler == "zeus":                          # vfit_mcmc.py:425-440
            import zeus
            sampler = zeus.EnsembleSampler(self.no_of_Chain, self.ndim, self.lnprob, vectorize=True, verbose=verbose)
            sampler.run_mcmc(guesses, self.no_of_steps)
        elif sampler == "host-slice":
            from .sampler import EnsembleSliceSampler
            sampler = EnsembleSliceSampler(self.no_of_Chain, self.ndim, self.lnprob, seed=seed)
            sampler.run_mcmc(guesses, self.no_of_steps)
        elif sampler == "emcee":
            import emcee
            sampler = emcee.EnsembleSampler(self.no_of_Chain, self.ndim, self.lnprob, vectorize=True)
            sampler.run_mcmc(guesses, self.no_of_steps, progress=verbose)
        elif sampler in ("device", "device-slice") and self._host_instruments:
            raise ValueError("sampler='device' / 'device-slice' keep the walker loop on the GPU; an instrument whose model is a "
                             "Python callable is evaluated on the host: use 'emcee', 'zeus', 'host' or 'host-slice'")
        elif sampler == "device-slice":                # zeus' move with the whole walker loop on the GPU (vp_slice_run)
            from .sampler import DeviceSliceSampler
            sampler = DeviceSliceSampler(self.no_of_Chain, self.ndim, self.engine, seed=seed)
            sampler.run_mcmc(guesses, self.no_of_steps)
        elif sampler == "device":                      # whole walker loop on the GPU (vp_stretch_run)
            from .sampler import DeviceStretchSampler
            sampler = DeviceStretchSampler(self.no_of_Chain, self.ndim, self.engine, seed=seed)
            sampler.run_mcmc(guesses, self.no_of_steps)
        else:
            sampler = StretchMoveSampler(self.no_of_Chain, self.ndim, self.lnprob, seed=seed)
            sampler.run_mcmc(guesses, self.no_of_steps)
        self.sampler = sampler
        self.mcmc_flag = True
        burn = int(0.2 * self.no_of_steps)
        self.samples = sampler.get_chain(discard=burn, flat=True)
        lo, med, hi = np.percentile(self.samples, [16, 50, 84], axis=0)     # compute_best_theta (:563-570)
        self.best_theta, self.low_theta, self.high_theta = med, lo, hi
        return sampler
