"""CPU ORACLE for the rbvfit lnprob hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A NumPy/SciPy restatement of the reference's algorithm for the one path this repository
accelerates: Voigt forward model + Gaussian log-likelihood.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it; the product
package ``rbvfit_amd`` never does (it fails loudly if its HIP library is missing).

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks every function below against
golden vectors produced by running the real reference in the build container
(``tests/golden/make_golden.py``; rbvfit 2.4.0, scipy 1.7.1, astropy 4.3.1).

Third-party arithmetic on the path that is not in /root/reference (un-vendored, lower-bound
pins only -- setup.cfg:30-37): ``scipy.special.wofz`` (Faddeeva w(z); S. G. Johnson's package:
ACM TOMS Algorithm 916 [Zaghloul & Ali 2011] near the real axis, a Gautschi/Poppe-Wijers
continued fraction elsewhere), ``scipy.ndimage.convolve1d`` and ``astropy.convolution.convolve``.
SciPy is present on the GPU box, so ``wofz`` is called directly here (the same routine the
reference calls); the two convolutions are restated in NumPy because astropy is not.

Each function cites the reference lines it follows (paths relative to
/root/reference/src/rbvfit/).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np
from scipy.special import wofz

# lsf_mode values (shared with the fixtures and with include/rbvfit_amd.h)
LSF_NONE = 0          # kernel is None                         (core/voigt_model.py:221)
LSF_SCIPY_NEAREST = 1  # ndimage.convolve1d(mode='nearest'), raw taps          (:224)
LSF_ASTROPY_EXTEND = 2  # astropy convolve(boundary='extend'), taps / sum(taps) (:227,230)


@dataclass
class OracleModelData:
    """Mirror of ``CompiledModelData`` (core/voigt_model.py:265-280) with the astropy kernel
    object replaced by its tap array + the dispatch branch it selects."""
    atomic_lambda0: np.ndarray
    atomic_gamma: np.ndarray      # float32 in the reference (rb_setline.py:44); any float here
    atomic_f: np.ndarray          # float32 in the reference (rb_setline.py:42)
    z_factors: np.ndarray
    N_indices: np.ndarray
    b_indices: np.ndarray
    v_indices: np.ndarray
    taps: np.ndarray
    lsf_mode: int = LSF_NONE
    voigt_method: str = "wofz"

    @property
    def n_lines(self) -> int:
        return len(self.atomic_lambda0)


def h_tepper_garcia(x: np.ndarray, a: np.ndarray) -> np.ndarray:
    """'fast' H(a,x): core/voigt_approx.py:69-86 (bug-compatible far wings)."""
    x2 = x * x
    G = np.exp(-x2)
    sqrt_pi = np.sqrt(np.pi)
    eps = np.maximum(1e-2, 100.0 * np.abs(a) / sqrt_pi)              # :74
    safe = np.maximum(x2, eps)                                       # :75
    numer = G * (4.0 * safe ** 2 + 7.0 * safe + 4.0) - 1.5           # :79
    denom = safe * (safe + 1.0) ** 2                                 # :80
    H_tg = G - (a / sqrt_pi) * numer / denom                         # :81
    H_core = G * (1.0 - 2.0 * a / sqrt_pi)                           # :84
    return np.where(x2 < eps, H_core, H_tg)                          # :86


def voigt_tau(lambda0, gamma, f, N_linear, b_values, wave_rest, voigt_method="wofz"):
    """(L,P) optical depths: core/voigt_model.py:100-159, same operation order and dtypes."""
    c_freq = 2.99792458e18                                           # :130
    atomic_constant = 4.48898479507e3                                # :131
    lambda0_bc = lambda0[:, np.newaxis]
    gamma_bc = gamma[:, np.newaxis]
    f_bc = f[:, np.newaxis]
    N_bc = N_linear[:, np.newaxis]
    b_bc = b_values[:, np.newaxis]
    b_f = b_bc / lambda0_bc * 1e13                                   # :142
    freq0 = c_freq / lambda0_bc                                      # :143
    freq = c_freq / wave_rest                                        # :144
    constant = atomic_constant / (freq0 * b_bc)                      # :146
    a = gamma_bc / (4 * np.pi * b_f)                                 # :149
    x = (freq - freq0) / b_f                                         # :150
    if voigt_method == "fast":
        H = h_tepper_garcia(x, a)                                    # :154
    else:
        H = wofz(x + 1j * a).real                                    # :156
    return N_bc * f_bc * constant * H                                # :158


def lsf_convolve(flux: np.ndarray, taps: np.ndarray, lsf_mode: int) -> np.ndarray:
    """LSF dispatch of core/voigt_model.py:220-230, restated without scipy.ndimage/astropy.

    Both third-party routines compute a TRUE convolution (kernel flipped) with the edge value
    replicated; for K taps (K odd, centre c=K//2):  out[p] = sum_j k[j] * f[clamp(p + c - j)].
    The astropy branch additionally divides the taps by their sum (normalize_kernel=True
    default).  Pinned by tests/golden/conv_semantics.npz (asymmetric kernel).
    """
    if lsf_mode == LSF_NONE or taps is None or len(taps) == 0:
        return flux
    k = np.asarray(taps, dtype=np.float64)
    K = k.size
    c = K // 2
    padded = np.concatenate([np.full(c, flux[0]), flux, np.full(K - 1 - c, flux[-1])])
    if lsf_mode == LSF_ASTROPY_EXTEND and np.isnan(flux.sum()):
        # astropy.convolution.convolve's default nan_treatment='interpolate' (convolve.py:318, 374-389 of astropy 4.3.1; the C
        # loop of _convolveNd_c): as soon as ANY sample is NaN every output is  top / bot  with
        #   top = sum of val * ker,  bot = sum of ker   over the window's non-NaN samples
        # -- a NaN sample is replaced by the kernel-weighted mean of its neighbours and the outputs around it are
        # renormalised; an output whose whole window is NaN keeps the input's value (NaN).  Pinned by
        # tests/golden/nan_semantics.npz and nan_wave_custom.npz (a NaN wavelength sample makes such a model pixel).
        win = np.lib.stride_tricks.sliding_window_view(padded, K)          # win[p, j] = padded[p + j]
        kf = k[::-1]                                                       # true convolution: kernel flipped
        ok = ~np.isnan(win)
        top = np.where(ok, win * kf, 0.0).sum(axis=1)
        bot = np.where(ok, kf, 0.0).sum(axis=1)
        with np.errstate(invalid="ignore", divide="ignore"):
            return np.where(bot == 0.0, flux, top / np.where(bot == 0.0, 1.0, bot))
    if lsf_mode == LSF_ASTROPY_EXTEND:
        k = k / k.sum()
    # np.convolve flips the kernel; 'valid' over the edge-padded signal gives P outputs
    return np.convolve(padded, k, mode="valid")


def model_flux(data: OracleModelData, theta: np.ndarray, wavelength: np.ndarray,
               return_unconvolved: bool = False) -> np.ndarray:
    """One theta -> flux(P): ``_evaluate_compiled_model`` core/voigt_model.py:162-261."""
    theta = np.asarray(theta, dtype=np.float64)
    N_linear = 10 ** theta[data.N_indices]                           # :192
    b_values = theta[data.b_indices]                                 # :193
    v_values = theta[data.v_indices]                                 # :194
    c = 299792.458                                                   # :197
    z_total = data.z_factors * (1 + v_values / c) - 1                # :200
    wave_rest = wavelength[np.newaxis, :] / (1 + z_total[:, np.newaxis])  # :203-204
    tau_all = voigt_tau(data.atomic_lambda0, data.atomic_gamma, data.atomic_f,
                        N_linear, b_values, wave_rest, data.voigt_method)  # :207-211
    tau_total = np.sum(tau_all, axis=0)                              # :214
    flux = np.exp(-tau_total)                                        # :217
    if return_unconvolved:
        return flux
    return lsf_convolve(flux, data.taps, data.lsf_mode)             # :220-230


@dataclass
class OracleInstrument:
    """One entry of ``vfit.instrument_data`` after ``_compile_models`` (vfit_mcmc.py:234-259)."""
    data: OracleModelData
    wave: np.ndarray
    flux: np.ndarray
    inv_sigma2: np.ndarray
    log_inv_sigma2: np.ndarray

    @classmethod
    def from_error(cls, data, wave, flux, error):
        error = np.asarray(error)
        return cls(data, np.asarray(wave), np.asarray(flux),
                   1.0 / (error ** 2), np.log(1.0 / (error ** 2)))   # vfit_mcmc.py:255-256 (T4)


def lnprior(theta, lb, ub) -> float:
    """vfit_mcmc.py:291-295 (bounds inclusive)."""
    if np.any(theta < lb) or np.any(theta > ub):
        return -np.inf
    return 0.0


def lnlike(theta, instruments: Sequence[OracleInstrument]) -> float:
    """vfit_mcmc.py:297-319 (no 2*pi term, T6)."""
    total = 0.0
    for inst in instruments:
        model_dat = model_flux(inst.data, theta, inst.wave)          # :304
        total += -0.5 * np.sum((inst.flux - model_dat) ** 2 * inst.inv_sigma2
                               - inst.log_inv_sigma2)                # :309-311
    return total


def lnprob(theta, lb, ub, instruments: Sequence[OracleInstrument]) -> float:
    """vfit_mcmc.py:348-353: out-of-bounds -> -inf WITHOUT evaluating the model."""
    theta = np.asarray(theta, dtype=np.float64)
    lp = lnprior(theta, lb, ub)
    if not np.isfinite(lp):
        return -np.inf
    return lp + lnlike(theta, instruments)


def lnprob_batch(thetas, lb, ub, instruments) -> np.ndarray:
    """What emcee does with pool=None: a serial map of lnprob over walker rows."""
    return np.array([lnprob(t, lb, ub, instruments) for t in np.atleast_2d(thetas)], dtype=np.float64)


def gaussian_taps(fwhm_pixels: float, normalize: bool = False) -> np.ndarray:
    """Taps of ``Gaussian1DKernel(stddev=FWHM/2.355)`` (core/voigt_model.py:462-464, literal
    2.355 = trap T3).  astropy's default size is 8 sigma rounded up to the next odd integer,
    sampled at integer offsets ('center' discretisation).  ``normalize=False`` reproduces
    astropy 4.3.1 (taps NOT normalised, trap T2); ``True`` divides by the sum."""
    sigma = float(fwhm_pixels) / 2.355
    size = int(np.ceil(8 * sigma))
    if size % 2 == 0:
        size += 1
    j = np.arange(size, dtype=np.float64) - size // 2
    amplitude = 1.0 / (np.sqrt(2 * np.pi) * sigma)        # astropy Gaussian1D model form
    taps = amplitude * np.exp(-0.5 * j ** 2 / sigma ** 2)
    return taps / taps.sum() if normalize else taps


# ---------------------------------------------------------------------------------------------
# fixture adapters (tests/golden/*.npz -> oracle objects)
# ---------------------------------------------------------------------------------------------
def data_from_fixture(z, inst: str) -> OracleModelData:
    g = lambda k: z[f"{inst}__{k}"]
    gamma = g("gamma")
    f = g("f")
    if bool(g("gamma_is_f32")):
        gamma = gamma.astype(np.float32)     # keep the reference's dtype inside the oracle (T1)
        f = f.astype(np.float32)
    return OracleModelData(g("lambda0"), gamma, f, g("zfac"), g("N_idx").astype(np.int64),
                           g("b_idx").astype(np.int64), g("v_idx").astype(np.int64),
                           g("taps"), int(g("lsf_mode")),
                           "fast" if int(g("voigt_method")) == 1 else "wofz")


def instruments_from_fixture(z):
    out = []
    for inst in [str(s) for s in z["instruments"]]:
        out.append(OracleInstrument(data_from_fixture(z, inst), z[f"{inst}__wave"], z[f"{inst}__flux"],
                                    z[f"{inst}__inv_sigma2"], z[f"{inst}__log_inv_sigma2"]))
    return out
