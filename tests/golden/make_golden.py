#!/opt/conda/bin/python3.9
"""Generate the golden input/output vectors under tests/golden/ from the REAL reference.

Runs ONLY in the build container (never on the GPU box, never at test time):

    PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 tests/golden/make_golden.py

It imports rongmon/rbvfit from /root/reference/src (read-only) and calls the reference's
own hot path -- ``VoigtModel(...).compile().model_flux`` (voigt_model.py:295-311) and
``vfit.lnprob`` (vfit_mcmc.py:348-353) -- on seeded inputs, then stores inputs AND outputs
as small .npz files.  Nothing of the reference's source is stored: fixtures are data only
(the compiled line tables, LSF taps, spectra, theta rows, flux rows, lnprob values).

In-process accommodations (none of them touches the reference):
  * astropy 4.3.1 (the only astropy in this image) references three numpy names that
    numpy 1.26 removed; they are re-added before the import.
  * ``rbvfit.vfit_mcmc`` imports ``emcee`` and ``corner`` at module top (vfit_mcmc.py:21,25).
    Neither is installed anywhere in the image and neither is used by lnprior/lnlike/lnprob,
    so two empty placeholder modules are registered in ``sys.modules`` for the import.

Versions that bind the fixtures: numpy 1.26.4, scipy 1.7.1 (scipy.special.wofz,
scipy.ndimage.convolve1d), astropy 4.3.1 (Gaussian1DKernel is NOT normalised in this
version -- trap T2 -- so fixtures store the exact taps the reference used).
"""
import os
import sys
import types
import json

import numpy as np

np.asscalar = lambda a: np.asarray(a).item()          # astropy 4.3.1 compat (see docstring)
np.alen = lambda a: len(np.asarray(a))
if not hasattr(np, "rank"):
    np.rank = lambda a: np.ndim(a)
for _name in ("emcee", "corner"):                       # import-only placeholders
    if _name not in sys.modules:
        _m = types.ModuleType(_name)
        _m.__version__ = "placeholder-not-installed"
        _m.EnsembleSampler = None
        sys.modules[_name] = _m

sys.path.insert(0, "/root/reference/src")
import scipy                                            # noqa: E402
import astropy                                          # noqa: E402
from scipy.special import wofz                          # noqa: E402
from astropy.convolution import Gaussian1DKernel, CustomKernel  # noqa: E402
from rbvfit.core.fit_configuration import FitConfiguration      # noqa: E402
from rbvfit.core.voigt_model import VoigtModel                  # noqa: E402
import rbvfit.vfit_mcmc as mc                                    # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
VERSIONS = dict(numpy=np.__version__, scipy=scipy.__version__, astropy=astropy.__version__,
                rbvfit="2.4.0 (/root/reference)")


def _kernel_info(kernel):
    """(taps, lsf_mode): 0 none, 1 scipy convolve1d mode='nearest' (raw taps),
    2 astropy convolve boundary='extend' (taps divided by their sum)."""
    if kernel is None:
        return np.zeros(0), 0
    if isinstance(kernel, Gaussian1DKernel):
        return np.asarray(kernel.array, dtype=np.float64), 1
    return np.asarray(kernel.array, dtype=np.float64), 2


def _tables(compiled):
    d = compiled.data
    taps, mode = _kernel_info(d.kernel)
    return dict(
        lambda0=np.asarray(d.atomic_lambda0, dtype=np.float64),
        gamma=np.asarray(d.atomic_gamma).astype(np.float64),     # float32 -> widened (T1)
        f=np.asarray(d.atomic_f).astype(np.float64),
        gamma_is_f32=np.array(np.asarray(d.atomic_gamma).dtype == np.float32),
        zfac=np.asarray(d.z_factors, dtype=np.float64),
        N_idx=np.asarray(d.N_indices, dtype=np.int32),
        b_idx=np.asarray(d.b_indices, dtype=np.int32),
        v_idx=np.asarray(d.v_indices, dtype=np.int32),
        taps=taps, lsf_mode=np.array(mode),
        voigt_method=np.array(0 if d.voigt_method == "wofz" else 1),
    )


def make_config(spec):
    cfg = FitConfiguration()
    for z, ion, trans, nc in spec:
        cfg.add_system(z=z, ion=ion, transitions=list(trans), components=nc)
    return cfg


def build_case(name, spec, instruments, theta_true, seed, n_theta, n_flux_rows,
               voigt_method="wofz", spread=1e-3, extra_thetas=None, noise=0.05,
               err_dtype=np.float64):
    """instruments: list of (inst_name, wave, fwhm_or_None, custom_taps_or_None)."""
    rng = np.random.default_rng(seed)
    cfg = make_config(spec)
    theta_true = np.asarray(theta_true, dtype=np.float64)
    C = theta_true.size // 3
    inst_data = {}
    compiled = {}
    for iname, wave, fwhm, custom in instruments:
        m = VoigtModel(cfg, FWHM=fwhm, voigt_method=voigt_method)
        if custom is not None:
            m.kernel = CustomKernel(np.asarray(custom, dtype=np.float64))
        cm = m.compile()
        compiled[iname] = cm
        clean = cm.model_flux(theta_true, wave)
        err = np.full(wave.size, noise).astype(err_dtype)
        flux = (clean + rng.normal(0.0, noise, wave.size)).astype(err_dtype)
        inst_data[iname] = dict(model=m, wave=wave, flux=flux, error=err)
    _, lb, ub = mc.set_bounds(theta_true[:C], theta_true[C:2 * C], theta_true[2 * C:])
    fit = mc.vfit(inst_data, theta_true, lb, ub)
    thetas = theta_true + spread * rng.standard_normal((n_theta, theta_true.size))
    thetas = np.clip(thetas, lb + 1e-10, ub - 1e-10)
    # a few wider excursions so the fixtures are not all within 1e-3 of the truth
    k = min(8, n_theta // 2)
    thetas[1:1 + k] = np.clip(theta_true + rng.uniform(-0.3, 0.3, (k, theta_true.size)) *
                              np.concatenate([np.full(C, 1.0), np.full(C, 10.0), np.full(C, 20.0)]),
                              lb + 1e-10, ub - 1e-10)
    thetas[0] = theta_true
    # out-of-bounds rows (prior -> -inf, model not evaluated): one below lb, one above ub
    if n_theta >= 6:
        thetas[-1] = theta_true
        thetas[-1, 0] = lb[0] - 0.5
        thetas[-2] = theta_true
        thetas[-2, -1] = ub[-1] + 1.0
    if extra_thetas is not None:
        thetas = np.vstack([thetas, np.asarray(extra_thetas, dtype=np.float64)])
    lnprob = np.array([fit.lnprob(t) for t in thetas], dtype=np.float64)
    out = dict(name=np.array(name), theta_true=theta_true, lb=lb, ub=ub, thetas=thetas,
               lnprob=lnprob, instruments=np.array([i[0] for i in instruments]),
               versions=np.array(json.dumps(VERSIONS)))
    for iname, wave, fwhm, custom in instruments:
        d = fit.instrument_data[iname]
        tb = _tables(compiled[iname])
        for k_, v_ in tb.items():
            out[f"{iname}__{k_}"] = v_
        out[f"{iname}__wave"] = d["wave"]
        out[f"{iname}__flux"] = d["flux"]                       # dtype preserved (T4)
        out[f"{iname}__error"] = d["error"]
        out[f"{iname}__inv_sigma2"] = d["inv_sigma2"]
        out[f"{iname}__log_inv_sigma2"] = d["log_inv_sigma2"]
        rows = [compiled[iname].model_flux(t, d["wave"]) for t in thetas[:n_flux_rows]]
        out[f"{iname}__model_flux"] = np.array(rows, dtype=np.float64)
    path = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{name}: D={theta_true.size} n_theta={len(thetas)} lnprob[0]={lnprob[0]!r} "
          f"finite={np.isfinite(lnprob).sum()} -> {os.path.getsize(path)/1024:.0f} KiB")
    return fit, thetas, lnprob


MGII = [(0.348, "MgII", [2796.35, 2803.53], 2)]
MULTI = [(0.348, "MgII", [2796.35, 2803.53], 3),
         (0.348, "FeII", [2600.17, 2586.65, 2382.77], 3),
         (0.348, "CIV", [1548.20, 1550.77], 2)]
STRESS = [(0.348 + 0.01 * i, "MgII", [2796.35, 2803.53], 8) for i in range(4)]


def theta_random(seed, C):
    rng = np.random.default_rng(seed)
    return np.concatenate([rng.uniform(12.8, 13.8, C), rng.uniform(8, 35, C), rng.uniform(-120, 120, C)])


def cos_like_kernel():
    """Synthetic tabulated 'COS-like' LSF of SURVEY 8(d) C3: 0.85 N(0,2.2) + 0.15 Lorentz(6),
    j=-50..50, skewed by (1+0.002 j).  NOT normalised on purpose (astropy path normalises)."""
    j = np.arange(-50, 51, dtype=np.float64)
    g = np.exp(-0.5 * (j / 2.2) ** 2) / (np.sqrt(2 * np.pi) * 2.2)
    lor = (6.0 / np.pi) / (j ** 2 + 36.0)
    return (0.85 * g + 0.15 * lor) * (1.0 + 0.002 * j)


def main():
    theta_c0 = np.array([13.5, 13.2, 15.0, 25.0, -40.0, 20.0])
    wave_c0 = np.linspace(3755.0, 3795.0, 4096)

    # ---- C0/C1: the bench configuration at full size ---------------------------------------
    build_case("c0_mgii", MGII, [("G", wave_c0, "6.5", None)], theta_c0, 11, 32, 6)
    build_case("c0_mgii_fast", MGII, [("G", wave_c0, "6.5", None)], theta_c0, 11, 12, 4,
               voigt_method="fast")
    build_case("c0_mgii_nolsf", MGII, [("G", wave_c0, None, None)], theta_c0, 11, 12, 4)
    # strong / saturated / narrow lines: large tau0, larger damping parameter
    theta_strong = np.array([15.4, 14.6, 4.0, 8.0, -45.0, 25.0])
    build_case("c0_mgii_strong", MGII, [("G", wave_c0, "2.2", None)], theta_strong, 15, 10, 10,
               spread=5e-2)
    # ragged pixel count, short spectrum (P < K) and asymmetric tabulated kernel
    build_case("ragged_1000", MGII, [("G", np.linspace(3760.0, 3790.0, 1000), "4.0", None)],
               theta_c0, 21, 10, 4)
    build_case("tiny_7px", MGII, [("G", np.linspace(3768.0, 3770.0, 7), "6.5", None)],
               theta_c0, 22, 8, 8, noise=0.05)
    build_case("one_px", MGII, [("G", np.array([3769.3]), "2.0", None)], theta_c0, 23, 6, 6)

    # ---- anchors quoted in SURVEY.md 8(a)-notes -------------------------------------------
    cfg = make_config(MGII)
    m = VoigtModel(cfg, FWHM="6.5").compile()
    fl = m.model_flux(theta_c0, wave_c0)
    print("anchor sum(flux) =", repr(fl.sum()), " expected 3906.1710281280307")

    # ---- DLA-like HI Lya: damping wings dominate, tau0 ~ 1e7 --------------------------------
    dla = [(0.1, "HI", [1215.67], 1)]
    build_case("dla_lya", dla, [("G", np.linspace(1290.0, 1385.0, 3000), "3.0", None)],
               np.array([20.3, 30.0, 10.0]), 31, 10, 4, spread=1e-2)

    # ---- C2-mini: 19 lines -----------------------------------------------------------------
    th2 = theta_random(12, 8)
    build_case("c2_mini", MULTI, [("G", np.linspace(2050.0, 3800.0, 4096), "6.5", None)], th2, 12, 16, 3)
    # same physics on a resolved sub-window (lines actually sampled)
    build_case("c2_window", MULTI, [("G", np.linspace(3480.0, 3800.0, 6000), "6.5", None)], th2, 12, 8, 2)

    # ---- C3-mini: two instruments, tabulated (normalising) kernel + Gaussian ------------------
    th3 = theta_random(13, 8)
    build_case("c3_mini", MULTI,
               [("A", np.linspace(2050.0, 2925.0, 2048), "6.5", cos_like_kernel()),
                ("B", np.linspace(2925.0, 3800.0, 2048), "2.5", None)], th3, 13, 16, 3)

    # ---- C4-mini: 64 lines, 4 systems -----------------------------------------------------------
    th4 = theta_random(14, 32)
    build_case("c4_mini", STRESS, [("G", np.linspace(3700.0, 3900.0, 4096), "6.5", None)], th4, 14, 8, 2)

    # ---- real data: the reference's saved COS fit (float32 flux/error, trap T4) -----------------
    import h5py
    with h5py.File("/root/reference/src/rbvfit/tests/test.h5", "r") as f:
        wave = np.asarray(f["instruments/COS/wave"][...], dtype=np.float64)
        flux = f["instruments/COS/flux"][...]
        err = f["instruments/COS/error"][...]
        best = f["best_fit"][...]
        samples = f["samples"][::500][:14]
        cfgj = json.loads(f["config_metadata"].attrs["config_data"])
    spec = []
    for s in cfgj["systems"]:
        for g in s["ion_groups"]:
            spec.append((s["redshift"], g["ion_name"], g["transitions"], g["components"]))
    fwhm = str(cfgj.get("instrumental_params", {}).get("FWHM", "2.394991274145626"))
    cfg = make_config(spec)
    mdl = VoigtModel(cfg, FWHM=fwhm)
    C = best.size // 3
    _, lb, ub = mc.set_bounds(best[:C], best[C:2 * C], best[2 * C:])
    fit = mc.vfit({"COS": dict(model=mdl, wave=wave, flux=flux, error=err)}, best, lb, ub)
    thetas = np.vstack([best[None, :], np.clip(samples, lb + 1e-10, ub - 1e-10)])
    lnp = np.array([fit.lnprob(t) for t in thetas])
    d = fit.instrument_data["COS"]
    cm = mdl.compile()
    out = dict(name=np.array("real_cos"), theta_true=best, lb=lb, ub=ub, thetas=thetas, lnprob=lnp,
               instruments=np.array(["COS"]), versions=np.array(json.dumps(VERSIONS)),
               fwhm=np.array(fwhm))
    for k_, v_ in _tables(cm).items():
        out[f"COS__{k_}"] = v_
    out.update({"COS__wave": d["wave"], "COS__flux": d["flux"], "COS__error": d["error"],
                "COS__inv_sigma2": d["inv_sigma2"], "COS__log_inv_sigma2": d["log_inv_sigma2"],
                "COS__model_flux": np.array([cm.model_flux(t, wave) for t in thetas[:4]])})
    np.savez_compressed(os.path.join(HERE, "real_cos.npz"), **out)
    print("real_cos: lnprob(best) =", repr(lnp[0]), " expected 205.56708945563835; weights dtype",
          d["inv_sigma2"].dtype)

    # ---- H(a,x) grid from scipy.special.wofz ---------------------------------------------------------
    a = np.concatenate([[0.0, 1e-12, 3e-10], np.logspace(-9, -1, 33), [0.2, 0.5, 1.0, 3.0, 8.0]])
    x = np.concatenate([np.linspace(0, 12, 481), np.logspace(np.log10(12.0), np.log10(3e4), 160),
                        [5e-4, 4.9e-4, 1e-8, 5.99, 6.0, 6.01, 7.99, 8.0, 8.01, 27.9, 28.0, 28.1]])
    x = np.concatenate([x, -x[1:40]])
    A, X = np.meshgrid(a, x, indexing="ij")
    H = wofz(X + 1j * A).real
    np.savez_compressed(os.path.join(HERE, "hgrid.npz"), a=a, x=x, H=H,
                        versions=np.array(json.dumps(VERSIONS)))
    print("hgrid:", H.shape)

    # ---- Gaussian LSF taps for the FWHM values of SURVEY A9 ---------------------------------------
    taps = {}
    for fw in ("2.0", "2.2", "2.5", "4.0", "6.5", "8.0", "13.0", "2.394991274145626"):
        k = Gaussian1DKernel(stddev=float(fw) / 2.355)
        taps[f"fwhm_{fw}"] = np.asarray(k.array, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "taps.npz"), **taps)
    print("taps:", {k: v.size for k, v in taps.items()})

    # ---- convolution semantics with an asymmetric kernel (both branches of voigt_model.py:220-230)
    from scipy import ndimage
    from astropy.convolution import convolve as astropy_convolve
    rng = np.random.default_rng(5)
    sig = rng.uniform(0.0, 1.0, 40)
    ker = np.array([0.05, 0.1, 0.5, 0.2, 0.1, 0.03, 0.02])
    np.savez_compressed(os.path.join(HERE, "conv_semantics.npz"), signal=sig, kernel=ker,
                        scipy_nearest=ndimage.convolve1d(sig, ker, mode="nearest"),
                        astropy_extend=astropy_convolve(sig, CustomKernel(ker), boundary="extend"))


def nan_cases():
    """NaN wavelength samples: a NaN in the wavelength grid makes that model pixel NaN; the reference's
    CustomKernel branch (astropy ``convolve``, default nan_treatment='interpolate', voigt_model.py:227,230) then replaces it by the
    kernel-weighted mean of its finite neighbours and renormalises every output whose window holds a NaN, while the Gaussian
    branch (scipy ``convolve1d``, :224) lets it poison K outputs -- and lnprob.  Two fixtures, one per branch, same spectrum."""
    from astropy.convolution import convolve as astropy_convolve
    theta_c0 = np.array([13.5, 13.2, 15.0, 25.0, -40.0, 20.0])
    wave = np.linspace(3760.0, 3790.0, 700)
    bad = [0, 137, 300, 301, 455, 456, 457, 699]          # isolated, pairs, a triple, both edges (boundary='extend' pads with them)
    wave_nan = wave.copy()
    wave_nan[bad] = np.nan
    ker = np.array([0.02, 0.05, 0.12, 0.45, 0.2, 0.1, 0.04, 0.015, 0.005])      # asymmetric, sum != 1
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        build_case("nan_wave_custom", MGII, [("G", wave_nan, "6.5", ker)], theta_c0, 41, 10, 4)
        build_case("nan_wave_gauss", MGII, [("G", wave_nan, "2.5", None)], theta_c0, 41, 8, 2)
        rng = np.random.default_rng(6)
        sig = rng.uniform(0.2, 1.0, 40)
        sig[[0, 7, 8, 20, 21, 22, 39]] = np.nan
        k7 = np.array([0.05, 0.1, 0.5, 0.2, 0.1, 0.03, 0.02])
        wide = sig.copy(); wide[10:19] = np.nan                                  # a gap wider than the kernel: NaN survives
        np.savez_compressed(os.path.join(HERE, "nan_semantics.npz"), signal=sig, kernel=k7, signal_wide_gap=wide,
                            astropy_extend=astropy_convolve(sig, CustomKernel(k7), boundary="extend"),
                            astropy_extend_wide_gap=astropy_convolve(wide, CustomKernel(k7), boundary="extend"))
    print("nan_semantics: written")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "nan":
        nan_cases()
    else:
        main()
        nan_cases()
