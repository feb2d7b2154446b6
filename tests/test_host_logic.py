"""Host-side mirror of the reference interface (CPU only): table construction, bounds, taps,
validation, walker sharding, sampler."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_golden
from rbvfit_amd import atomic, LSF_SCIPY_NEAREST, LSF_ASTROPY_EXTEND, LSF_NONE
from rbvfit_amd.lsf import gaussian_taps
from rbvfit_amd.model import FitConfiguration, VoigtModel, mean_fwhm_pixels, tables_from_rbvfit
from rbvfit_amd.vfit import set_bounds, vfit
from rbvfit_amd.dist import shard_bounds
from rbvfit_amd.sampler import StretchMoveSampler, initialize_walkers

MGII = [(0.348, "MgII", [2796.35, 2803.53], 2)]
MULTI = [(0.348, "MgII", [2796.35, 2803.53], 3), (0.348, "FeII", [2600.17, 2586.65, 2382.77], 3),
         (0.348, "CIV", [1548.20, 1550.77], 2)]
STRESS = [(0.348 + 0.01 * i, "MgII", [2796.35, 2803.53], 8) for i in range(4)]


def _model(spec, fwhm="6.5", **kw):
    cfg = FitConfiguration()
    for z, ion, tr, nc in spec:
        cfg.add_system(z, ion, tr, nc)
    kw.setdefault("normalize_kernel", False)          # the fixtures hold astropy-4.3.1 (raw) taps
    return VoigtModel(cfg, FWHM=fwhm, **kw)


@pytest.mark.parametrize("spec,fixture", [(MGII, "c0_mgii"), (MULTI, "c2_mini"), (STRESS, "c4_mini")])
def test_tables_equal_the_reference_compiled_model(spec, fixture):
    """Line order, theta index maps, float32 atomic data and z factors of the reference's
    CompiledModelData (voigt_model.py:386-442) captured in the golden fixtures."""
    z = load_golden(fixture)
    m = _model(spec)
    np.testing.assert_array_equal(m.atomic_lambda0, z["G__lambda0"])
    assert m.atomic_gamma.dtype == np.float32 and m.atomic_f.dtype == np.float32        # trap T1
    np.testing.assert_array_equal(m.atomic_gamma.astype(np.float64), z["G__gamma"])
    np.testing.assert_array_equal(m.atomic_f.astype(np.float64), z["G__f"])
    np.testing.assert_array_equal(m.z_factors, z["G__zfac"])
    np.testing.assert_array_equal(m.N_indices, z["G__N_idx"])
    np.testing.assert_array_equal(m.b_indices, z["G__b_idx"])
    np.testing.assert_array_equal(m.v_indices, z["G__v_idx"])
    np.testing.assert_allclose(m.taps, z["G__taps"], rtol=5e-16)
    assert m.lsf_mode == LSF_SCIPY_NEAREST == int(z["G__lsf_mode"])


def test_real_cos_tables():
    z = load_golden("real_cos")
    cfg = FitConfiguration()
    cfg.add_system(0.0, "SiII", [1190.4158, 1193.2897], 1)
    cfg.add_system(0.162005, "HI", [1025.7223], 1)
    m = VoigtModel(cfg, FWHM=str(z["fwhm"]), normalize_kernel=False)
    np.testing.assert_array_equal(m.atomic_lambda0, z["COS__lambda0"])
    np.testing.assert_array_equal(m.N_indices, [0, 0, 1])
    np.testing.assert_array_equal(m.b_indices, [2, 2, 3])
    np.testing.assert_array_equal(m.v_indices, [4, 4, 5])
    assert m.taps.size == 9


def test_kernel_branches():
    assert _model(MGII, None).lsf_mode == LSF_NONE and _model(MGII, None).taps is None
    m = _model(MGII, "6.5", kernel_taps=[0.1, 0.5, 0.2])
    assert m.lsf_mode == LSF_ASTROPY_EXTEND
    assert abs(gaussian_taps(6.5, normalize=True).sum() - 1.0) < 1e-15
    assert abs(gaussian_taps(6.5).sum() - 0.999972) < 1e-6          # trap T2: raw taps are not normalised
    with pytest.raises(ValueError):
        _model(MGII, "6.5", voigt_method="nope")


def test_default_taps_are_the_pinned_ones_and_normalised_taps_are_their_quotient():
    """VoigtModel's default kernel path (no normalize_kernel argument) is the one the golden fixtures pin: the reference's
    own Gaussian1DKernel(...).array (astropy 4.3.1, tests/golden/taps.npz).  normalize_kernel=True is pinned against the
    same fixture: taps / sum(taps)."""
    z = load_golden("taps")
    cfg = FitConfiguration()
    cfg.add_system(0.348, "MgII", [2796.35, 2803.53], 2)
    for key in z.files:
        fw = key[len("fwhm_"):]
        ref = np.asarray(z[key], dtype=np.float64)
        m = VoigtModel(cfg, FWHM=fw)                                   # the default
        assert m.lsf_mode == LSF_SCIPY_NEAREST and m.taps.size == ref.size
        np.testing.assert_allclose(m.taps, ref, rtol=5e-16, atol=0)
        mn = VoigtModel(cfg, FWHM=fw, normalize_kernel=True)
        np.testing.assert_allclose(mn.taps, ref / ref.sum(), rtol=5e-16, atol=0)
        assert abs(mn.taps.sum() - 1.0) < 1e-15


def test_mean_fwhm_pixels_matches_the_reference():
    """core/voigt_model.py:33-58 on a linear, a log-spaced and an irregular grid (values made by the reference itself,
    tests/golden/make_golden_host.py)."""
    z = load_golden("host_helpers")
    for name in ("linear", "loglam", "irregular"):
        got = [mean_fwhm_pixels(float(f), z[f"{name}__wave"]) for f in z["fwhm_kms"]]
        np.testing.assert_allclose(got, z[f"{name}__pixels"], rtol=1e-14, atol=0)
    with pytest.raises(ValueError, match="strictly positive"):
        mean_fwhm_pixels(10.0, np.array([0.0, 1.0, 2.0]))
    with pytest.raises(ValueError, match="at least two points"):
        mean_fwhm_pixels(10.0, np.array([1000.0]))


def test_cos_lsf_points_at_kernel_taps():
    cfg = FitConfiguration()
    cfg.add_system(0.0, "SiII", [1190.4158], 1)
    with pytest.raises(ImportError, match="kernel_taps"):
        VoigtModel(cfg, FWHM="COS")
    cfg2 = FitConfiguration(FWHM="COS")                                # (FWHM through the configuration, as the reference allows)
    cfg2.add_system(0.0, "SiII", [1190.4158], 1)
    with pytest.raises(ImportError, match="linetools"):
        VoigtModel(cfg2)
    assert VoigtModel(cfg, FWHM="COS", kernel_taps=[0.2, 0.6, 0.2]).lsf_mode == LSF_ASTROPY_EXTEND


def test_atomic_lookup_mirrors_rb_setline():
    info = atomic.lookup(2796.3, "closest")
    assert info["wave"] == 2796.352 and info["fval"].dtype == np.float32 and info["gamma"].dtype == np.float32
    assert info["fval"] == np.float32(0.6123) and info["name"] == "MgII 2796"
    with pytest.raises(KeyError):
        atomic.lookup(2796.3, "Exact")
    with pytest.raises(ValueError):
        atomic.lookup(2796.3, "nearest")


def test_unknown_wavelength_raises_instead_of_snapping_to_an_unrelated_line(tmp_path):
    """With only the 39 built-in transitions loaded, 'closest' must not turn FeII 1608 into AlII 1670
    or ZnII 2026 into AlIII 1862 (what an unguarded argmin over a subset does); FitConfiguration and
    VoigtModel go through the same lookup.  A caller-supplied list in the format of rbvfit's
    lines/atom_full.dat (rb_setline.py:66-98) lifts the limit, as the reference's full list does."""
    import importlib
    from rbvfit_amd.model import FitConfiguration
    at = importlib.reload(atomic)                       # private table state for this test
    try:
        n0 = at.table_size()
        for lam in (1608.45, 2026.14, 977.02):
            with pytest.raises(at.UnknownLineError):
                at.lookup(lam, "closest")
        with pytest.raises(at.UnknownLineError):
            FitConfiguration().add_system(0.1, "FeII", [1608.45], 1)
        assert at.lookup(2796.3, "closest")["name"] == "MgII 2796"       # within 0.5 A: still snaps
        tab = tmp_path / "atoms.dat"
        tab.write_text("# ion wrest fval gamma\n"
                       "FeII   1608.4511 0.057700  2.740E8\n"
                       "ZnII   2026.1360 0.489000  4.070E8\n"
                       "\n"
                       "MgII   2796.3520 0.612300  2.612E8\n")
        assert at.load_table(str(tab)) == 3
        assert at.table_size() == n0 + 2                  # the MgII row replaced the built-in one
        info = at.lookup(1608.45, "closest")
        assert info["name"] == "FeII 1608" and info["fval"] == np.float32(0.0577) and info["gamma"] == np.float32(2.74e8)
        assert at.lookup(1500.0, "closest")["name"] in ("SiII 1526", "CIV 1548")   # full list: unconditional, as the reference
        lst = tmp_path / "sub.lst"
        lst.write_text("wrest ion n f\n609.95      MgX 609   0.0842       2\n")
        assert at.load_table(str(lst), fmt="lst", full=False) == 1
        assert at.lookup(609.9, "Exact" if False else "closest")["fval"] == np.float32(0.0842)
        with pytest.raises(ValueError):
            bad = tmp_path / "bad.dat"
            bad.write_text("FeII 1608.45\n")
            at.load_table(str(bad))
    finally:
        importlib.reload(atomic)


REF_ATOMS = "/root/reference/src/rbvfit/lines/atom_full.dat"


@pytest.mark.skipif(not os.path.exists(REF_ATOMS), reason="the reference checkout is only present in the build container")
def test_builtin_rows_equal_the_reference_list_and_the_full_list_loads():
    import importlib
    at = importlib.reload(atomic)
    try:
        builtin = {(r[0], r[1]): r for r in at._LINES}
        assert at.load_table(REF_ATOMS) == 320
        full = {(r[0], r[1]): r for r in at._LINES}
        assert len(full) == 320                            # every built-in row is a row of the reference's list
        for key, row in builtin.items():
            assert np.float32(full[key][2]) == np.float32(row[2]) and np.float32(full[key][3]) == np.float32(row[3])
        assert at.lookup(1608.45, "closest")["name"] == "FeII 1608"
    finally:
        importlib.reload(atomic)


def test_set_bounds_matches_fixture():
    z = load_golden("c2_mini")
    th = z["theta_true"]
    C = th.size // 3
    _, lb, ub = set_bounds(th[:C], th[C:2 * C], th[2 * C:])
    np.testing.assert_array_equal(lb, z["lb"])
    np.testing.assert_array_equal(ub, z["ub"])
    _, lb2, _ = set_bounds([13.0], [20.0], [0.0], Nlow=[10.0])
    assert lb2[0] == 10.0


def test_vfit_validation_errors_match_the_reference():
    with pytest.raises(TypeError):
        vfit._validate_unified_instrument_data([1, 2])
    with pytest.raises(ValueError):
        vfit._validate_unified_instrument_data({})
    with pytest.raises(ValueError):
        vfit._validate_unified_instrument_data({"A": {"model": None, "wave": [1], "flux": [1]}})
    with pytest.raises(ValueError):
        vfit._validate_unified_instrument_data({"A": {"model": None, "wave": [1, 2], "flux": [1], "error": [1, 2]}})
    with pytest.raises(ValueError):
        vfit._validate_guesses([1.0, 5.0], [0.0, 0.0], [2.0, 2.0])
    with pytest.raises(ValueError):
        vfit._validate_guesses([1.0], [0.0, 0.0], [2.0, 2.0])


def _tables_from_fixture(z, inst):
    from rbvfit_amd.model import CompiledModelData
    g = lambda k: z[f"{inst}__{k}"]
    gamma, f = g("gamma"), g("f")
    if bool(g("gamma_is_f32")):
        gamma, f = gamma.astype(np.float32), f.astype(np.float32)
    C = len(z["lb"]) // 3
    return CompiledModelData(g("lambda0"), gamma, f, g("zfac"), g("N_idx"), g("b_idx"), g("v_idx"), g("taps"),
                             int(g("lsf_mode")), len(g("lambda0")), C, "fast" if int(g("voigt_method")) == 1 else "wofz")


class _OracleEngine:
    """Stands for rbvfit_amd.Engine in the CPU suite (the HIP engine needs a GPU): same calls, the oracle's arithmetic."""
    def __init__(self, device_id=0):
        self.inst = []

    def set_bounds(self, lb, ub):
        self.lb, self.ub = np.asarray(lb, float), np.asarray(ub, float)

    def add_instrument(self, wave, flux, inv_sigma2, log_inv_sigma2, lambda0, gamma, f, zfac, N_idx, b_idx, v_idx,
                       taps=None, lsf_mode=0, voigt_method=0):
        from oracle import voigt_oracle as vo
        d = vo.OracleModelData(lambda0, gamma, f, zfac, np.asarray(N_idx, np.int64), np.asarray(b_idx, np.int64),
                               np.asarray(v_idx, np.int64), taps, lsf_mode, "fast" if voigt_method == 1 else "wofz")
        self.inst.append(vo.OracleInstrument(d, wave, flux, inv_sigma2, log_inv_sigma2))
        return len(self.inst) - 1

    def lnprob(self, th):
        from oracle import voigt_oracle as vo
        return vo.lnprob_batch(th, self.lb, self.ub, self.inst)

    def model_flux(self, idx, rows):
        from oracle import voigt_oracle as vo
        return np.array([vo.model_flux(self.inst[idx].data, t, self.inst[idx].wave) for t in np.atleast_2d(rows)])

    def close(self):
        pass


def test_user_callable_instruments_are_evaluated_on_the_host(monkeypatch):
    """SURVEY A4 / vfit_mcmc.py:242-248, 304: an instrument whose 'model' is any callable (theta, wave) -> flux is taken
    verbatim; its likelihood term is added to the table-driven instruments'.  Rows outside the prior box evaluate no
    model; any exception inside a row's likelihood makes that row -inf (vfit_mcmc.py:317-319)."""
    from oracle import voigt_oracle as vo
    import rbvfit_amd.vfit as V
    monkeypatch.setattr(V, "Engine", _OracleEngine)
    z = np.load(os.path.join(GOLDEN, "c3_mini.npz"), allow_pickle=False)
    insts = vo.instruments_from_fixture(z)
    calls = []

    def model_b(theta, wave):                                    # a plain Python callable: the reference's pass-through branch
        calls.append(np.array(theta, copy=True))
        assert np.ndim(theta) == 1 and wave is not None
        return vo.model_flux(insts[1].data, theta, wave)

    data = {"A": {"model": _tables_from_fixture(z, "A"), "wave": z["A__wave"], "flux": z["A__flux"], "error": z["A__error"]},
            "B": {"model": model_b, "wave": z["B__wave"], "flux": z["B__flux"], "error": z["B__error"]}}
    fit = vfit(data, z["theta_true"], z["lb"], z["ub"])
    assert fit.instrument_data["B"]["index"] is None and fit.instrument_data["A"]["index"] == 0
    thetas = z["thetas"].copy()
    thetas[3, 0] = z["lb"][0] - 1.0                              # outside the box: no model is evaluated for it
    got = fit.lnprob(thetas)
    want = vo.lnprob_batch(thetas, z["lb"], z["ub"], insts)
    assert got[3] == -np.inf and len(calls) == int(np.sum(~np.isneginf(want))) < len(thetas)
    fin = np.isfinite(want)
    np.testing.assert_allclose(got[fin], want[fin], rtol=1e-13)
    np.testing.assert_allclose(got[fin & (np.arange(len(got)) != 3)], z["lnprob"][fin & (np.arange(len(got)) != 3)], rtol=1e-10, atol=1e-7)
    assert isinstance(fit.lnprob(thetas[0]), float) and fit.lnprob(thetas[0]) == got[0]
    # lnlike: no prior, every row evaluated, the out-of-bounds one too
    ll = fit.lnlike(thetas)
    np.testing.assert_allclose(ll, [vo.lnlike(t, insts) for t in thetas], rtol=1e-13)
    # chi2 counts both instruments' weights
    const = sum(float(np.sum(np.asarray(i.log_inv_sigma2, float))) for i in insts)
    np.testing.assert_allclose(fit.chi2(thetas[:2]), -2.0 * ll[:2] + const, rtol=1e-13)

    # an exception in the callable -> that row's likelihood is -inf, the others are untouched
    def model_raises(theta, wave):
        if theta[0] > z["theta_true"][0]:
            raise RuntimeError("model failed")
        return vo.model_flux(insts[1].data, theta, wave)
    data["B"]["model"] = model_raises
    fit2 = vfit(data, z["theta_true"], z["lb"], z["ub"])
    got2 = fit2.lnprob(z["thetas"])
    bad = z["thetas"][:, 0] > z["theta_true"][0]
    assert bad.any() and (~bad).any()
    assert np.all(np.isneginf(got2[bad]))
    np.testing.assert_allclose(got2[~bad], z["lnprob"][~bad], rtol=1e-10, atol=1e-7)
    # NaN from the callable is NOT caught: it propagates (trap T7)
    data["B"]["model"] = lambda theta, wave: np.full(wave.shape, np.nan)
    assert np.all(np.isnan(vfit(data, z["theta_true"], z["lb"], z["ub"]).lnprob(z["thetas"][:2])))
    # only callables: no table-driven instrument at all
    only = vfit({"B": {"model": model_b, "wave": z["B__wave"], "flux": z["B__flux"], "error": z["B__error"]}},
                z["theta_true"], z["lb"], z["ub"])
    np.testing.assert_allclose(only.lnprob(z["thetas"][:3]), vo.lnprob_batch(z["thetas"][:3], z["lb"], z["ub"], insts[1:]), rtol=1e-13)
    # the device-resident walker loops cannot call back into Python
    with pytest.raises(ValueError):
        fit.runmcmc(sampler="device")
    # a bound model_flux of a compiled model (what the reference's own _compile_models stores) is NOT a host callable:
    # its tables go to the engine and the method is never called on the host
    from rbvfit_amd.model import CompiledVoigtModel
    cm = CompiledVoigtModel(_tables_from_fixture(z, "B"))
    data["B"]["model"] = cm.model_flux
    fit3 = vfit(data, z["theta_true"], z["lb"], z["ub"])
    assert fit3.instrument_data["B"]["index"] == 1 and not fit3._host_instruments
    np.testing.assert_allclose(fit3.lnprob(z["thetas"][:3]), z["lnprob"][:3], rtol=1e-10, atol=1e-7)


def test_normalize_kernel_default_follows_the_installed_astropy(monkeypatch):
    """the mirror's Gaussian taps are raw (astropy 4.3.1, the pinned fixtures) unless the astropy
    installed next to it normalises its Gaussian kernels; no astropy -> raw."""
    import rbvfit_amd.model as M
    cfg = FitConfiguration(); cfg.add_system(0.348, "MgII", [2796.35, 2803.53], 1)
    z = load_golden("taps")
    raw = z["fwhm_6.5"]
    for answer, want_norm in ((None, False), (False, False), (True, True)):
        monkeypatch.setattr(M, "_ASTROPY_NORMALIZES", answer)
        m = VoigtModel(cfg, FWHM="6.5")
        assert m.normalize_kernel is want_norm
        assert (abs(m.taps.sum() - 1.0) < 1e-14) == want_norm
        if not want_norm:
            np.testing.assert_allclose(m.taps, raw, rtol=4e-16, atol=0)     # (exp of two libm builds: 1 ulp)
    monkeypatch.setattr(M, "_ASTROPY_NORMALIZES", True)
    assert abs(VoigtModel(cfg, FWHM="6.5", normalize_kernel=False).taps.sum() - 1.0) > 1e-6      # explicit choice wins
    monkeypatch.setattr(M, "_ASTROPY_NORMALIZES", "unknown")
    assert M.astropy_normalizes_gaussian() in (None, True, False)


def test_tables_from_rbvfit_duck_typing():
    class Gaussian1DKernel:           # stands for astropy's class of the same name
        array = np.array([0.25, 0.5, 0.25])

    class Data:
        atomic_lambda0 = np.array([2796.352]); atomic_gamma = np.array([2.612e8], dtype=np.float32)
        atomic_f = np.array([0.6123], dtype=np.float32); z_factors = np.array([1.348])
        N_indices = np.array([0]); b_indices = np.array([1]); v_indices = np.array([2])
        kernel = Gaussian1DKernel(); n_lines = 1; total_components = 1; voigt_method = "fast"

    class Compiled:
        data = Data()

    t = tables_from_rbvfit(Compiled())
    assert t.lsf_mode == LSF_SCIPY_NEAREST and t.voigt_method == "fast" and t.taps.size == 3
    kw = t.engine_kwargs()
    assert kw["gamma"].dtype == np.float64 and kw["gamma"][0] == float(np.float32(2.612e8))


def test_shard_bounds_cover_all_rows_once():
    for W in (0, 1, 3, 7, 50, 512, 2048):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                lo, hi = shard_bounds(W, world, r)
                assert 0 <= lo <= hi <= W
                seen += list(range(lo, hi))
            assert seen == list(range(W))


def test_stretch_move_sampler_recovers_a_gaussian():
    mu, sig = np.array([1.0, -2.0, 0.5]), np.array([0.5, 2.0, 1.0])
    calls = []

    def lp(th):
        calls.append(th.shape)
        return -0.5 * np.sum(((th - mu) / sig) ** 2, axis=1)

    s = StretchMoveSampler(40, 3, lp, seed=3)
    p0 = initialize_walkers(mu, mu - 10, mu + 10, 40, 1e-2, lp, np.random.default_rng(0))
    s.run_mcmc(p0, 1500)
    c = s.get_chain(discard=500, flat=True)
    assert np.all(np.abs(c.mean(0) - mu) < 0.15 * sig) and np.all(np.abs(c.std(0) / sig - 1) < 0.1)
    assert all(sh == (20, 3) for sh in calls[2:])           # one batched call per half-ensemble
    assert 0.3 < s.acceptance_fraction.mean() < 0.9
    with pytest.raises(ValueError):
        StretchMoveSampler(5, 3, lp)
    with pytest.raises(ValueError):
        StretchMoveSampler(40, 3, lambda th: np.full(len(th), np.nan)).run_mcmc(p0, 1)


def test_initialize_walkers_redraws_non_finite_rows():
    lb, ub = np.array([0.0, 0.0]), np.array([1.0, 1.0])
    lp = lambda th: np.where(th[:, 0] > 0.5, 0.0, -np.inf)
    pos = initialize_walkers([0.5, 0.5], lb, ub, 64, 0.1, lp, np.random.default_rng(1))
    assert np.all(pos[:, 0] > 0.5) and np.all((pos > lb) & (pos < ub))
    with pytest.raises(RuntimeError):
        initialize_walkers([0.5, 0.5], lb, ub, 8, 0.1, lambda th: np.full(len(th), -np.inf),
                           np.random.default_rng(1), max_attempts=3)


# ---- Philox4x32-10 of the device sampler (host evaluation of the same function) -----------------
def _philox_py(ctr, key):
    M0, M1, W0, W1, mask = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85, 0xFFFFFFFF
    c, k = list(ctr), list(key)
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [(p1 >> 32) ^ c[1] ^ k[0], p1 & mask, (p0 >> 32) ^ c[3] ^ k[1], p0 & mask]
        k = [(k[0] + W0) & mask, (k[1] + W1) & mask]
    return c


def _philox_lib(ctr, key):
    import ctypes as C
    from rbvfit_amd import _lib
    lib = _lib.load()
    c = (C.c_uint32 * 4)(*ctr); k = (C.c_uint32 * 2)(*key); o = (C.c_uint32 * 4)()
    lib.vp_philox4x32(c, k, o)
    return list(o)


def test_philox4x32_known_answers_and_python_restatement():
    # Random123 known-answer vectors for philox4x32-10
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, out in kat:
        assert tuple(_philox_py(ctr, key)) == out
        assert tuple(_philox_lib(ctr, key)) == out
    rng = np.random.default_rng(5)
    for _ in range(50):
        ctr = [int(v) for v in rng.integers(0, 2 ** 32, 4)]
        key = [int(v) for v in rng.integers(0, 2 ** 32, 2)]
        assert _philox_lib(ctr, key) == _philox_py(ctr, key)


# ---- host ensemble drivers on an analytic posterior (no GPU needed) ----------------------------------
def _gauss3():
    cov = np.array([[1.0, 0.6, -0.3], [0.6, 2.0, 0.4], [-0.3, 0.4, 0.5]])
    mean = np.array([1.0, -2.0, 0.5])
    icov = np.linalg.inv(cov)

    def lnprob(x):
        d = np.atleast_2d(x) - mean
        out = -0.5 * np.einsum("ij,jk,ik->i", d, icov, d)
        out[np.any(np.abs(d) > 50, axis=1)] = -np.inf            # a box prior, as vfit has
        return out
    return mean, cov, lnprob


@pytest.mark.parametrize("kind", ["stretch", "slice"])
def test_host_samplers_recover_a_correlated_gaussian(kind):
    from rbvfit_amd.sampler import EnsembleSliceSampler, StretchMoveSampler, gelman_rubin
    mean, cov, lnprob = _gauss3()
    rng = np.random.default_rng(0)
    p0 = mean + 0.1 * rng.standard_normal((24, 3))
    if kind == "stretch":
        s = StretchMoveSampler(24, 3, lnprob, seed=1)
        s.run_mcmc(p0, 3000)
        burn = 500
    else:
        s = EnsembleSliceSampler(24, 3, lnprob, seed=1)
        s.run_mcmc(p0, 800)
        burn = 200
        assert min(s.batch_sizes) < 12 and max(s.batch_sizes) == 24          # ragged batches: 12, then fewer
        assert 0.1 < s.mu < 20                                               # tuned scale stays sane
        assert 4 < s.n_lnprob_evals / (800 * 24) < 15                        # ~5-10 evaluations per walker-step
    flat = s.get_chain(discard=burn, flat=True)
    sd = np.sqrt(np.diag(cov))
    assert np.all(np.abs(flat.mean(axis=0) - mean) < 0.15 * sd)
    assert np.all(np.abs(np.cov(flat.T) - cov) < 0.2 * np.outer(sd, sd))
    assert np.all(gelman_rubin(s.get_chain(discard=burn)) < 1.1)
    np.testing.assert_allclose(s.lnprobability[-1], lnprob(s.chain[-1]), rtol=1e-12, atol=1e-12)


def test_slice_sampler_rejects_nan_and_bad_shapes():
    from rbvfit_amd.sampler import EnsembleSliceSampler
    with pytest.raises(ValueError):
        EnsembleSliceSampler(5, 3, lambda x: np.zeros(len(x)))
    s = EnsembleSliceSampler(8, 2, lambda x: np.full(len(x), np.nan), seed=0)
    with pytest.raises(ValueError, match="NaN"):
        s.run_mcmc(np.zeros((8, 2)), 1)
    s = EnsembleSliceSampler(8, 2, lambda x: np.full(len(x), -np.inf), seed=0)
    with pytest.raises(ValueError, match="finite"):
        s.run_mcmc(np.zeros((8, 2)), 1)


def test_farfield_expansion_series_and_tail_bound():
    """The arithmetic farfield_kernel relies on (csrc/voigt_kernels.h), checked in exact-enough NumPy: with
    x = x_c (1 + r t), |t| <= 1,   x^-k = x_c^-k sum_j (-1)^j C(k+j-1, j) r^j t^j,   and the first 12 terms of the k = 2
    series miss the full value by less than 13 |r|^12 / (1 - |r|)^2 (relative); higher k only as weights of x_c^-(k-2)."""
    from math import comb
    t = np.linspace(-1.0, 1.0, 401)
    for r in (0.125, -0.125, 0.06, -0.01):
        for k in (2, 4, 6, 12):
            full = (1.0 + r * t) ** (-k)
            part = sum((-1) ** j * comb(k + j - 1, j) * r ** j * t ** j for j in range(12))
            err = np.max(np.abs(part - full) / full)
            if k == 2:
                assert err <= 13.0 * abs(r) ** 12 / (1.0 - abs(r)) ** 2 + 1e-15, (r, k, err)
            # the terms of a line are K_m x_c^-(2m+2): against the k = 2 term the k-th one is down by x_c^-(k-2) <= 30^-(k-2),
            # which outweighs the faster growth of its coefficients
            assert err * 30.0 ** (-(k - 2)) <= 13.0 * 0.125 ** 12 / 0.875 ** 2 * 1.0001 + 1e-15, (r, k, err)
    # the kernel's acceptance rule keeps K_0 x_c^-2 * bound below 1e-16 in optical depth: at the widest allowed block
    # (|r| = 1/8) that admits lines whose optical depth at the block is below ~4e-7
    bound = 13.0 * 0.125 ** 12 / 0.875 ** 2
    assert 3e-7 < 1e-16 / bound < 5e-7
