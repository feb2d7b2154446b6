"""Pin the CPU oracle to the golden vectors generated from the real reference.

CPU-only (-m "not gpu").  Everything the GPU parity tests later trust is checked here first.
"""
import numpy as np
import pytest
from scipy.special import wofz

from conftest import golden_cases, load_golden, FLUX_ATOL, LNPROB_RTOL, LNPROB_ATOL
from oracle import voigt_oracle as vo


@pytest.mark.parametrize("name", golden_cases())
def test_model_flux_matches_reference(name):
    z = load_golden(name)
    for iname in [str(s) for s in z["instruments"]]:
        data = vo.data_from_fixture(z, iname)
        ref = z[f"{iname}__model_flux"]
        for i in range(ref.shape[0]):
            got = vo.model_flux(data, z["thetas"][i], z[f"{iname}__wave"])
            # scipy 1.7.1 (fixtures) vs the scipy of this interpreter: 1.9e-14 rel on wofz
            np.testing.assert_allclose(got, ref[i], rtol=0, atol=FLUX_ATOL)


@pytest.mark.parametrize("name", golden_cases())
def test_lnprob_matches_reference(name):
    z = load_golden(name)
    insts = vo.instruments_from_fixture(z)
    got = vo.lnprob_batch(z["thetas"], z["lb"], z["ub"], insts)
    ref = z["lnprob"]
    assert np.array_equal(np.isneginf(got), np.isneginf(ref))     # -inf classes match exactly
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    fin = np.isfinite(ref)
    np.testing.assert_allclose(got[fin], ref[fin], rtol=LNPROB_RTOL, atol=LNPROB_ATOL)


def test_survey_anchors():
    """Deterministic anchors of SURVEY.md 8(a)-notes (reference under scipy 1.7.1)."""
    z = load_golden("c0_mgii")
    data = vo.data_from_fixture(z, "G")
    theta = np.array([13.5, 13.2, 15.0, 25.0, -40.0, 20.0])
    wave = np.linspace(3755.0, 3795.0, 4096)
    fl = vo.model_flux(data, theta, wave)
    assert abs(fl.sum() - 3906.1710281280307) < 1e-9
    np.testing.assert_allclose(
        fl[[0, 1400, 1480, 1500, 2048, 2470, 4095]],
        [0.999971635106006, 0.6596301594942791, 0.469756096067918, 0.2216979870179351,
         0.999967946799313, 0.6933889281882308, 0.9999717802796302], rtol=0, atol=1e-12)
    assert fl.argmin() == 1431 and abs(fl.min() - 0.005050158130151798) < 1e-12
    un = vo.model_flux(data, theta, wave, return_unconvolved=True)
    assert abs(un.sum() - 3906.2804039695293) < 1e-9
    assert abs(un[1500] - 0.21908604195808032) < 1e-12
    # noise-free data, err = 0.05: lnprob(theta) = 0.5 * 4096 * ln(400) exactly
    inst = vo.OracleInstrument.from_error(data, wave, fl, np.full(4096, 0.05))
    lb = np.array([11.5, 11.2, 2, 2, -90, -30.0])
    ub = np.array([15.5, 15.2, 55, 65, 10, 70.0])
    assert abs(vo.lnprob(theta, lb, ub, [inst]) - 12270.519392477147) < 1e-8
    th2 = theta + np.array([0.05, -0.03, 1.5, -2, 3, -4])
    assert abs(vo.lnprob(th2, lb, ub, [inst]) - 11614.3384821761) < 2e-6


def test_real_cos_tables_and_weights():
    z = load_golden("real_cos")
    assert abs(z["lnprob"][0] - 205.56708945563835) < 1e-12
    np.testing.assert_array_equal(z["COS__N_idx"], [0, 0, 1])
    np.testing.assert_array_equal(z["COS__b_idx"], [2, 2, 3])
    np.testing.assert_array_equal(z["COS__v_idx"], [4, 4, 5])
    np.testing.assert_allclose(z["COS__zfac"], [1, 1, 1.162005])
    assert z["COS__inv_sigma2"].dtype == np.float32      # trap T4: weights inherit float32
    assert z["COS__taps"].size == 9


def test_hgrid_wofz_noise_floor():
    """scipy.special.wofz of this interpreter vs the fixture's scipy 1.7.1 build."""
    z = load_golden("hgrid")
    A, X = np.meshgrid(z["a"], z["x"], indexing="ij")
    H = wofz(X + 1j * A).real
    np.testing.assert_allclose(H, z["H"], rtol=2e-13, atol=0)


def test_gaussian_taps_match_astropy():
    z = load_golden("taps")
    for key in z.files:
        fw = float(key.split("_", 1)[1])
        got = vo.gaussian_taps(fw, normalize=False)
        assert got.size == z[key].size
        np.testing.assert_allclose(got, z[key], rtol=5e-16, atol=0)
    sizes = {fw: vo.gaussian_taps(fw).size for fw in (2.0, 2.2, 2.5, 4.0, 6.5, 8.0, 13.0)}
    assert sizes == {2.0: 7, 2.2: 9, 2.5: 9, 4.0: 15, 6.5: 23, 8.0: 29, 13.0: 45}


def test_convolution_semantics_asymmetric_kernel():
    z = load_golden("conv_semantics")
    got1 = vo.lsf_convolve(z["signal"], z["kernel"], vo.LSF_SCIPY_NEAREST)
    np.testing.assert_allclose(got1, z["scipy_nearest"], rtol=0, atol=2e-16)
    got2 = vo.lsf_convolve(z["signal"], z["kernel"], vo.LSF_ASTROPY_EXTEND)
    np.testing.assert_allclose(got2, z["astropy_extend"], rtol=0, atol=4e-16)


def test_astropy_branch_interpolates_nan_samples():
    """voigt_model.py:227,230: astropy's convolve (CustomKernel / COS branch) runs with its default
    nan_treatment='interpolate' -- NaN samples are left out and every output is renormalised by the kernel weight that was
    used; a gap wider than the kernel keeps its NaN.  The scipy (Gaussian) branch poisons K outputs.  Reference-made vectors:
    tests/golden/nan_semantics.npz, nan_wave_custom.npz, nan_wave_gauss.npz (make_golden.py nan)."""
    z = load_golden("nan_semantics")
    for sig, ref in (("signal", "astropy_extend"), ("signal_wide_gap", "astropy_extend_wide_gap")):
        got = vo.lsf_convolve(z[sig], z["kernel"], vo.LSF_ASTROPY_EXTEND)
        assert np.array_equal(np.isnan(got), np.isnan(z[ref]))
        np.testing.assert_allclose(got[~np.isnan(got)], z[ref][~np.isnan(z[ref])], rtol=0, atol=4e-16)
    assert np.isnan(z["astropy_extend_wide_gap"]).sum() == 3 and not np.isnan(z["astropy_extend"]).any()
    poisoned = vo.lsf_convolve(z["signal"], z["kernel"], vo.LSF_SCIPY_NEAREST)
    assert np.isnan(poisoned).sum() > np.isnan(z["signal"]).sum()
    zc, zg = load_golden("nan_wave_custom"), load_golden("nan_wave_gauss")
    assert np.isnan(zc["G__wave"]).sum() == 8 and not np.isnan(zc["G__model_flux"]).any()
    assert np.isfinite(zc["lnprob"][:8]).all() and np.isneginf(zc["lnprob"][8:]).all()
    assert np.isnan(zg["lnprob"][:6]).all() and np.isneginf(zg["lnprob"][6:]).all()


def test_fast_method_is_bug_compatible_in_far_wings():
    """Trap T9: the 'fast' wings fall as x^-6; pixel 0 of C1 differs visibly from wofz."""
    zf = load_golden("c0_mgii_fast")
    zw = load_golden("c0_mgii")
    df = vo.data_from_fixture(zf, "G")
    assert df.voigt_method == "fast"
    theta = np.array([13.5, 13.2, 15.0, 25.0, -40.0, 20.0])
    wave = zw["G__wave"]
    fast = vo.model_flux(df, theta, wave, return_unconvolved=True)
    exact = vo.model_flux(vo.data_from_fixture(zw, "G"), theta, wave, return_unconvolved=True)
    assert abs(fast.sum() - 3906.4924624856358) < 1e-9
    assert abs(exact[0] - 0.9999996356) < 1e-9 and abs(fast[0] - 1.0) < 1e-12
