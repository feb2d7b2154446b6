"""GPU tests at BASELINE.json's full sizes (C1..C4) and of the host mirror on the device.

At full size the oracle is only run on a handful of rows (it needs 1-200 ms per row); the rest of
the batch is covered by size-independent properties: lnprob recomputed on the host from the
engine's own model flux, additivity over instruments, permutation invariance, -inf/NaN classes.
"""
import numpy as np
import pytest

from conftest import load_golden, FLUX_ATOL, LNPROB_RTOL, LNPROB_ATOL

pytestmark = pytest.mark.gpu


def _oracle_instruments(wl):
    from oracle import voigt_oracle as vo
    insts = []
    for data, (wave, flux, err) in zip(wl.tables, wl.spectra):
        od = vo.OracleModelData(data.atomic_lambda0, data.atomic_gamma, data.atomic_f, data.z_factors,
                                data.N_indices, data.b_indices, data.v_indices,
                                data.taps if data.taps is not None else np.zeros(0), data.lsf_mode, data.voigt_method)
        insts.append(vo.OracleInstrument.from_error(od, wave, flux, err))
    return vo, insts


def _host_lnlike_from_flux(wl, thetas):
    """-0.5 sum[(flux-model)^2 w - log w] with the model rows taken from the engine."""
    total = np.zeros(len(thetas))
    for k, (wave, flux, err) in enumerate(wl.spectra):
        model = wl.engine.model_flux(k, thetas)
        w = 1.0 / err ** 2
        total += -0.5 * np.sum((flux - model) ** 2 * w - np.log(w), axis=1)
    return total


def _posterior_and_wide_walkers(wl, walkers, name):
    """The batch the full-size tests evaluate: the ensemble a stretch-move burn-in of the device sampler leaves (posterior
    width: theta standard deviations of 0.005-0.3 instead of SURVEY 8d's 1e-3 ball, so the multipole tiers, far-field masks
    and line-core flags differ from walker to walker), with every 8th row replaced by a draw of the fuzz tests' width around
    theta_true (0.2 dex / 3 km/s / 15 km/s, clipped into the prior box)."""
    nsteps = {"C1": 600, "C2": 200, "C3": 200, "C4": 100}[name]
    pos = wl.engine.stretch_run(wl.thetas, nsteps, seed=21, store_chain=False)[0]
    th = np.ascontiguousarray(pos[:walkers])
    C = wl.ndim // 3
    rng = np.random.default_rng(77)
    wide = np.arange(3, walkers, 8)
    scale = np.concatenate([np.full(C, 0.2), np.full(C, 3.0), np.full(C, 15.0)])
    th[wide] = np.clip(wl.theta_true + rng.standard_normal((wide.size, wl.ndim)) * scale, wl.lb + 1e-9, wl.ub - 1e-9)
    return th, wide


# BASELINE.json walker counts: C1 512 and C2 1024 on one GPU; C3 2048 and C4 4096 over 8 GPUs -- run here both as
# the per-GPU share (256 / 512 walkers: what one rank of the sharded job evaluates) and as the whole ensemble on
# one GPU (the unsharded batch a single-GPU user would submit).
@pytest.mark.parametrize("name,walkers,n_oracle,n_prop", [("C1", 512, 24, 64), ("C2", 1024, 16, 16),
                                                         ("C3", 256, 16, 16), ("C3", 2048, 16, 16),
                                                         ("C4", 32, 8, 4), ("C4", 512, 8, 4), ("C4", 4096, 8, 4)])
def test_full_size_config(name, walkers, n_oracle, n_prop):
    from rbvfit_amd.workloads import make_workload
    # (the burn-in ensemble has at least 2 D + 2 walkers whatever the batch evaluated afterwards)
    wl = make_workload(name, walkers=max(walkers, 2 * {"C1": 6, "C2": 24, "C3": 24, "C4": 96}[name] + 64))
    try:
        th, wide = _posterior_and_wide_walkers(wl, walkers, name)
        spread = th[np.setdiff1d(np.arange(walkers), wide)].std(axis=0)
        assert np.median(spread) > 2e-3                # not the 1e-3 ball any more
        th[5, 0] = wl.lb[0] - 0.25                    # out of bounds -> -inf, model not evaluated
        th[7, -1] = wl.ub[-1] + 3.0
        got = wl.engine.lnprob(th)
        kind, info = wl.engine.last_launch_kind, wl.engine.last_farfield_info
        # which launch structure the configuration exercises (the automatic choice on this batch size)
        want_kind = {("C1", 512): "walker", ("C2", 1024): "tiles+farfield", ("C3", 256): "tiles+farfield",
                     ("C3", 2048): "tiles+farfield", ("C4", 32): "tiles", ("C4", 512): "tiles+farfield",
                     ("C4", 4096): "tiles+farfield"}[(name, walkers)]
        assert kind == want_kind, kind
        if name == "C4" and walkers >= 512:            # narrow pixels: cluster members enter the expansions line by line
            assert info["variant"] == "members" and info["covered_members"] > 0, info
        elif kind == "tiles+farfield":
            assert info["variant"] == "lines+clusters" and info["covered"] > 0, info
        assert got.shape == (walkers,)
        assert np.isneginf(got[5]) and np.isneginf(got[7])
        ok = np.ones(walkers, bool); ok[[5, 7]] = False
        assert np.all(np.isfinite(got[ok]))
        # oracle on rows spread over the batch: burn-in rows and wide rows, the first, the middle and the last
        vo, insts = _oracle_instruments(wl)
        rows_o = np.unique(np.concatenate([np.arange(0, 4), wide[: n_oracle // 2],
                                           np.linspace(8, walkers - 1, max(n_oracle - 4 - n_oracle // 2, 2)).astype(int)]))
        rows_o = rows_o[(rows_o != 5) & (rows_o != 7)]
        assert rows_o.size >= n_oracle - 2
        ref = vo.lnprob_batch(th[rows_o], wl.lb, wl.ub, insts)
        np.testing.assert_allclose(got[rows_o], ref, rtol=LNPROB_RTOL, atol=LNPROB_ATOL)
        # the same batch with the far-field expansions forced the other way (on where the rule left them off, off where it
        # switched them on): against the oracle rows and against the automatic structure
        if kind != "walker":
            wl.engine.set_option("farfield", 0 if kind == "tiles+farfield" else 1)
            other = wl.engine.lnprob(th)
            assert wl.engine.last_launch_kind == ("tiles" if kind == "tiles+farfield" else "tiles+farfield")
            if name == "C4":
                assert (wl.engine.last_farfield_info["variant"] == "members") == (kind == "tiles")
            wl.engine.set_option("farfield", -1)
            np.testing.assert_allclose(other[rows_o], ref, rtol=LNPROB_RTOL, atol=LNPROB_ATOL)
            np.testing.assert_allclose(other[ok], got[ok], rtol=1e-12, atol=1e-9)
        # property: lnprob == likelihood recomputed on the host from the engine's model flux
        rows = np.arange(8, 8 + n_prop)
        np.testing.assert_allclose(got[rows], _host_lnlike_from_flux(wl, th[rows]), rtol=1e-11, atol=1e-7)
        # property: permutation of the walkers permutes the result, bit for bit
        perm = np.random.default_rng(3).permutation(walkers)
        assert np.array_equal(wl.engine.lnprob(th[perm]), got[perm])
        # property: a second pass repeats exactly (the large ensembles go through other tile geometries and the finalize launch)
        assert np.array_equal(wl.engine.lnprob(th), got)
        # flux rows vs oracle: a burn-in row and a wide row, with the blocks' expansions and without
        for k, inst in enumerate(insts):
            for ffo in (1, 0):
                wl.engine.set_option("flux_farfield", ffo)
                fl = wl.engine.model_flux(k, th[[0, 3]])
                for i, r in enumerate((0, 3)):
                    np.testing.assert_allclose(fl[i], vo.model_flux(inst.data, th[r], inst.wave), rtol=0, atol=FLUX_ATOL)
            wl.engine.set_option("flux_farfield", -1)
    finally:
        wl.engine.close()


def test_joint_fit_is_the_sum_of_its_instruments():
    """C3: lnprob of the 2-instrument context == sum of single-instrument contexts (shared theta)."""
    import rbvfit_amd
    from rbvfit_amd.workloads import make_workload
    wl = make_workload("C3", walkers=64)
    try:
        joint = wl.engine.lnprob(wl.thetas)
        parts = np.zeros(64)
        for data, (wave, flux, err) in zip(wl.tables, wl.spectra):
            with rbvfit_amd.Engine(0) as e:
                e.set_bounds(wl.lb, wl.ub)
                e.add_instrument(wave, flux, 1 / err ** 2, np.log(1 / err ** 2), **data.engine_kwargs())
                parts += e.lnprob(wl.thetas)
        np.testing.assert_allclose(joint, parts, rtol=1e-13, atol=0)
    finally:
        wl.engine.close()


def test_nan_and_zero_error_classes_follow_the_reference():
    """NaN theta is in-bounds for the reference's comparisons and propagates; error == 0 gives
    inf - inf = NaN in the reference's likelihood."""
    from oracle import voigt_oracle as vo
    import rbvfit_amd
    z = load_golden("ragged_1000")
    insts = vo.instruments_from_fixture(z)
    th = z["thetas"][:4].copy()
    th[1, 2] = np.nan
    g = lambda k: z[f"G__{k}"]
    err = g("error").copy(); err[10] = 0.0
    with np.errstate(all="ignore"):
        w, lw = 1.0 / err ** 2, np.log(1.0 / err ** 2)
        bad = vo.OracleInstrument(insts[0].data, insts[0].wave, insts[0].flux, w, lw)
        ref_nan = vo.lnprob_batch(th, z["lb"], z["ub"], insts)
        ref_zero = vo.lnprob_batch(th, z["lb"], z["ub"], [bad])
    with rbvfit_amd.Engine(0) as e:
        e.set_bounds(z["lb"], z["ub"])
        kw = dict(lambda0=g("lambda0"), gamma=g("gamma"), f=g("f"), zfac=g("zfac"), N_idx=g("N_idx"), b_idx=g("b_idx"),
                  v_idx=g("v_idx"), taps=g("taps"), lsf_mode=int(g("lsf_mode")))
        e.add_instrument(g("wave"), g("flux"), g("inv_sigma2"), g("log_inv_sigma2"), **kw)
        got_nan = e.lnprob(th)
    with rbvfit_amd.Engine(0) as e:
        e.set_bounds(z["lb"], z["ub"])
        e.add_instrument(g("wave"), g("flux"), w, lw, **kw)
        got_zero = e.lnprob(th)
    assert np.array_equal(np.isnan(got_nan), np.isnan(ref_nan)) and np.isnan(got_nan[1])
    assert np.array_equal(np.isnan(got_zero), np.isnan(ref_zero)) and np.all(np.isnan(got_zero))


@pytest.mark.parametrize("b_value", [0.05, 0.002, 1e-4])
def test_lines_outside_the_fast_domain(b_value):
    """Damping parameter a > 0.1 (tiny b): continued fraction / Gaussian-sum generic path vs the
    oracle's scipy wofz.  b = 0.05 -> a ~ 0.15; 0.002 -> a ~ 3.6; 1e-4 -> a ~ 73."""
    from oracle import voigt_oracle as vo
    import rbvfit_amd
    z = load_golden("c0_mgii")
    data = vo.data_from_fixture(z, "G")
    th = z["theta_true"].copy()
    th[2] = b_value                      # first component's Doppler parameter
    th[0] = 12.0
    lb, ub = z["lb"].copy(), z["ub"].copy()
    lb[2] = 0.0
    insts = vo.instruments_from_fixture(z)
    ref_flux = vo.model_flux(data, th, z["G__wave"])
    ref_lnp = vo.lnprob(th, lb, ub, insts)
    g = lambda k: z[f"G__{k}"]
    with rbvfit_amd.Engine(0) as e:
        e.set_bounds(lb, ub)
        e.add_instrument(g("wave"), g("flux"), g("inv_sigma2"), g("log_inv_sigma2"), g("lambda0"), g("gamma"), g("f"),
                         g("zfac"), g("N_idx"), g("b_idx"), g("v_idx"), taps=g("taps"), lsf_mode=int(g("lsf_mode")))
        fl = e.model_flux(0, th)[0]
        lnp = e.lnprob(th)[0]
        # The launch for the flagged walkers is sized by what the batch BEFORE flagged (none: a few workgroups that walk the
        # walkers; some: one workgroup per walker): a mixed batch of 200 rows right after a quiet one, then again after itself,
        # as lnprob (tile launches forced) and as model_flux -- same bits both times, and the oracle's values.
        quiet = np.tile(z["theta_true"], (200, 1))
        mixed = quiet.copy()
        mixed[::7, 2] = b_value
        mixed[::7, 0] = 12.0
        e.set_option("walker", 0)
        q0 = e.lnprob(quiet)
        m1 = e.lnprob(mixed)               # after a quiet batch
        m2 = e.lnprob(mixed)               # after a batch with flagged walkers
        q1 = e.lnprob(quiet)
        assert np.array_equal(m1, m2) and np.array_equal(q0, q1)
        f0 = e.model_flux(0, quiet[:40])
        f1 = e.model_flux(0, mixed[:40])
        f2 = e.model_flux(0, mixed[:40])
        assert np.array_equal(f1, f2) and np.array_equal(f0[1], f1[1])
    # ... and with the fixture's own prior box (every line of an in-bounds walker in the fast domain): model_flux has no box, its
    # small batches are ONE walker_kernel launch (flux form) that leaves the rows with a line outside the fast domain to the
    # generic launch behind it -- mixed rows, after a quiet batch and after a flagged one, and a batch of flagged rows only
    with rbvfit_amd.Engine(0) as e2:
        e2.set_bounds(z["lb"], z["ub"])
        e2.add_instrument(g("wave"), g("flux"), g("inv_sigma2"), g("log_inv_sigma2"), g("lambda0"), g("gamma"), g("f"),
                          g("zfac"), g("N_idx"), g("b_idx"), g("v_idx"), taps=g("taps"), lsf_mode=int(g("lsf_mode")))
        w0 = e2.model_flux(0, quiet[:40])
        w1 = e2.model_flux(0, mixed[:40])
        w2 = e2.model_flux(0, mixed[:40])
        w3 = e2.model_flux(0, np.tile(th, (5, 1)))
        e2.set_option("flux_walker", 0)
        t1 = e2.model_flux(0, mixed[:40])
    assert np.array_equal(w1, w2) and np.array_equal(w1, t1) and np.array_equal(w0[1], w1[1]) and np.array_equal(w3[0], w1[0])
    np.testing.assert_allclose(w1[0], ref_flux, rtol=0, atol=FLUX_ATOL)
    np.testing.assert_allclose(fl, ref_flux, rtol=0, atol=FLUX_ATOL)
    np.testing.assert_allclose(lnp, ref_lnp, rtol=LNPROB_RTOL, atol=LNPROB_ATOL)
    np.testing.assert_allclose(m1[0], ref_lnp, rtol=LNPROB_RTOL, atol=LNPROB_ATOL)
    np.testing.assert_allclose(m1[1], vo.lnprob(z["theta_true"], lb, ub, insts), rtol=LNPROB_RTOL, atol=LNPROB_ATOL)
    np.testing.assert_allclose(f1[0], ref_flux, rtol=0, atol=FLUX_ATOL)


def test_vfit_mirror_and_compiled_model_on_device():
    from oracle import voigt_oracle as vo
    from rbvfit_amd.model import FitConfiguration, VoigtModel
    from rbvfit_amd.vfit import vfit
    z = load_golden("real_cos")                       # float32 flux/error from the reference's saved fit
    cfg = FitConfiguration()
    cfg.add_system(0.0, "SiII", [1190.4158, 1193.2897], 1)
    cfg.add_system(0.162005, "HI", [1025.7223], 1)
    model = VoigtModel(cfg, FWHM=str(z["fwhm"]), normalize_kernel=False)      # fixture taps are astropy-4.3.1's (raw)
    inst = {"COS": {"model": model, "wave": z["COS__wave"], "flux": z["COS__flux"], "error": z["COS__error"]}}
    fit = vfit(inst, z["theta_true"], z["lb"], z["ub"], no_of_Chain=20, no_of_steps=30)
    try:
        assert fit.instrument_data["COS"]["inv_sigma2"].dtype == np.float32         # trap T4
        one = fit.lnprob(z["thetas"][0])
        assert isinstance(one, float) and abs(one - 205.56708945563835) < 1e-7
        batch = fit.lnprob(z["thetas"])
        np.testing.assert_allclose(batch, z["lnprob"], rtol=LNPROB_RTOL, atol=LNPROB_ATOL)
        oob = z["thetas"][0].copy(); oob[0] = z["lb"][0] - 1
        assert fit.lnprob(oob) == -np.inf and fit.lnprior(oob) == -np.inf and np.isfinite(fit.lnlike(oob))
        insts = vo.instruments_from_fixture(z)
        assert abs(fit.lnlike(oob) - vo.lnlike(oob, insts)) < 1e-7
        # compiled model: single theta -> (P,), batch -> (W, P); evaluate() ignores voigt_method (T10)
        cm = model.compile()
        f1 = cm.model_flux(z["thetas"][0], z["COS__wave"])
        assert f1.shape == (184,)
        np.testing.assert_allclose(f1, z["COS__model_flux"][0], rtol=0, atol=FLUX_ATOL)
        fb = cm(z["thetas"][:4], z["COS__wave"])
        np.testing.assert_allclose(fb, z["COS__model_flux"], rtol=0, atol=FLUX_ATOL)
        un = model.evaluate(z["thetas"][0], z["COS__wave"], return_unconvolved=True)
        ref_un = vo.model_flux(insts[0].data, z["thetas"][0], z["COS__wave"], return_unconvolved=True)
        np.testing.assert_allclose(un, ref_un, rtol=0, atol=FLUX_ATOL)
        with pytest.raises(TypeError):
            import pickle
            pickle.dumps(cm)
        with pytest.raises(ValueError):
            fit.runmcmc(use_pool=True)
        s = fit.runmcmc(seed=5)                        # 20 walkers x 30 steps, batched lnprob
        assert fit.samples.shape[1] == 6 and np.all(np.isfinite(s.lnprobability))
        assert fit.best_theta.shape == (6,)
        lp0 = fit.lnprob(fit.theta)
        fit.runmcmc(optimize=True, seed=6)             # vfit_mcmc.py:507-512: theta <- optimize_guess(theta)
        assert fit.lnprob(fit.theta) >= lp0 and np.all(fit.theta >= fit.lb) and np.all(fit.theta <= fit.ub)
    finally:
        fit.close()


def test_walker_kernel_split_form_small_batches():
    """Batches of at most one walker per compute unit -- the reference's default is 50 walkers (vfit_mcmc.py:127-135) -- run every
    walker as 2, 4 or 8 workgroups of one-pass tiles (WalkerArgs::split).  Checked on C1: the number of groups by batch size,
    every row against the oracle, and bit for bit against (i) any other number of groups, (ii) the one-pass tile launches
    (option "geom" = 1), (iii) the same row in a batch of another size, (iv) the same batch through the host entry with and
    without the pre-armed launch; -inf rows and a NaN row among them."""
    from rbvfit_amd.workloads import make_workload
    wl = make_workload("C1", walkers=256)
    vo, insts = _oracle_instruments(wl)
    e = wl.engine
    try:
        th = wl.thetas.copy()
        th[5, 0] = wl.lb[0] - 0.5                          # outside the box
        th[9, 2] = np.nan                                  # NaN parameter: NaN lnprob (trap T7)
        th[250, 4] = wl.ub[4] + 1.0
        e.set_option("walker", 0); e.set_option("geom", 1); e.set_option("finalize", 0)
        tiles = e.lnprob(th)
        assert e.last_launch_kind == "tiles"
        e.set_option("walker", -1); e.set_option("geom", -1); e.set_option("finalize", -1)
        for W in (256, 32, 1):                              # off by default: a row's bits do not depend on the size of its batch
            e.lnprob(th[:W])
            assert e.last_launch_kind == "walker" and e.last_walker_split == 0
        e.set_option("walker_split", -1)
        ncu = 256                                           # MI355X
        for W in (256, 128, 64, 33, 32, 17, 7, 1):          # "by batch size": eight groups while W x 8 workgroups leave a CU at most one
            got = e.lnprob(th[:W])
            G = 8 if W * 8 <= ncu else 0
            assert e.last_launch_kind == "walker" and e.last_walker_split == G, (W, e.last_launch_kind, e.last_walker_split)
            if G:
                assert np.array_equal(got, tiles[:W], equal_nan=True), W
        assert np.isneginf(tiles[5]) and np.isneginf(tiles[250]) and np.isnan(tiles[9])
        rows = [0, 1, 5, 9, 17, 100, 255]
        ref = vo.lnprob_batch(th[rows], wl.lb, wl.ub, insts)
        fin = np.isfinite(ref)
        np.testing.assert_allclose(tiles[rows][fin], ref[fin], rtol=LNPROB_RTOL, atol=LNPROB_ATOL)
        for W, groups in ((256, (2, 4, 0)), (128, (2, 4, 8, 0)), (12, (2, 4, 8, 0))):      # forced: any number of groups, same bits
            for G in groups:
                e.set_option("walker_split", G)
                got = e.lnprob(th[:W])
                assert e.last_walker_split == G
                if G:
                    assert np.array_equal(got, tiles[:W], equal_nan=True)
                else:                                      # the ordinary form sums two-pass tiles: the last bit may differ
                    ok = np.isfinite(tiles[:W])
                    np.testing.assert_allclose(got[ok], tiles[:W][ok], rtol=4e-16, atol=0)
        e.set_option("walker_split", 8)
        e.set_option("prearm", 1)                           # the split form behind a pushed-to launch
        for _ in range(4):
            assert np.array_equal(e.lnprob(th[:64]), tiles[:64], equal_nan=True)
        assert e.prearm_counts["used"] >= 2 and e.last_walker_split == 8
        e.set_option("prearm", -1)
        e.set_option("walker_split", -1)
        # the device-resident stretch move on the split form: chain identical to the host loop's replay of the same draws is
        # covered in test_gpu_sampler (half-ensembles of <= 256 walkers take it by themselves); here: it runs and moves
        for W in (64, 48):                                  # half-ensembles of 32 / 24 walkers: the split form, with and without overlap
            pos, lp, *_ = e.stretch_run(wl.thetas[:W], 20, seed=5, store_chain=False)
            half = e.lnprob(pos[:W // 2])
            assert e.last_walker_split == 8
            np.testing.assert_array_equal(lp[:W // 2], half)
        e.set_option("walker_split", 0)
    finally:
        wl.engine.close()


def test_vfit_user_callable_instrument_mixed_with_gpu_instruments():
    """SURVEY A4 (vfit_mcmc.py:242-248, 304): one instrument of a joint fit on the GPU, the other a plain Python callable
    (here the oracle's model_flux: the reference's own arithmetic) evaluated on the host per row; -inf rows evaluate no
    model, an exception inside a row makes that row -inf (vfit_mcmc.py:317-319)."""
    from oracle import voigt_oracle as vo
    from rbvfit_amd.model import CompiledModelData
    from rbvfit_amd.vfit import vfit
    z = load_golden("c3_mini")
    insts = vo.instruments_from_fixture(z)

    def tables(inst):
        g = lambda k: z[f"{inst}__{k}"]
        return CompiledModelData(g("lambda0"), g("gamma").astype(np.float32), g("f").astype(np.float32), g("zfac"), g("N_idx"),
                                 g("b_idx"), g("v_idx"), g("taps"), int(g("lsf_mode")), len(g("lambda0")), len(z["lb"]) // 3, "wofz")

    calls = []

    def model_a(theta, wave):
        calls.append(1)
        return vo.model_flux(insts[0].data, theta, wave)

    data = {"A": {"model": model_a, "wave": z["A__wave"], "flux": z["A__flux"], "error": z["A__error"]},
            "B": {"model": tables("B"), "wave": z["B__wave"], "flux": z["B__flux"], "error": z["B__error"]}}
    fit = vfit(data, z["theta_true"], z["lb"], z["ub"], no_of_Chain=48, no_of_steps=2)
    try:
        got = fit.lnprob(z["thetas"])
        ref = z["lnprob"]
        assert np.array_equal(np.isneginf(got), np.isneginf(ref)) and len(calls) == int(np.sum(~np.isneginf(ref)))
        fin = np.isfinite(ref)
        np.testing.assert_allclose(got[fin], ref[fin], rtol=LNPROB_RTOL, atol=LNPROB_ATOL)
        np.testing.assert_allclose(fit.lnlike(z["thetas"][:3]), [vo.lnlike(t, insts) for t in z["thetas"][:3]], rtol=LNPROB_RTOL)
        s = fit.runmcmc(seed=3, sampler="host")           # the host walker loop calls back into Python per row
        assert np.all(np.isfinite(s.lnprobability))
        with pytest.raises(ValueError):
            fit.runmcmc(sampler="device")
        data["A"]["model"] = lambda theta, wave: (_ for _ in ()).throw(RuntimeError("boom"))
        fit2 = vfit(data, z["theta_true"], z["lb"], z["ub"])
        try:
            assert np.all(np.isneginf(fit2.lnprob(z["thetas"])))
        finally:
            fit2.close()
    finally:
        fit.close()


def test_engine_argument_errors():
    import rbvfit_amd
    z = load_golden("one_px")
    g = lambda k: z[f"G__{k}"]
    with rbvfit_amd.Engine(0) as e:
        with pytest.raises(rbvfit_amd.RbvfitAmdError):        # bounds first
            e.add_instrument(g("wave"), g("flux"), g("inv_sigma2"), g("log_inv_sigma2"), g("lambda0"), g("gamma"),
                             g("f"), g("zfac"), g("N_idx"), g("b_idx"), g("v_idx"))
        e.set_bounds(z["lb"], z["ub"])
        with pytest.raises(rbvfit_amd.RbvfitAmdError):        # lnprob before any instrument
            e.lnprob(z["thetas"])
        with pytest.raises(rbvfit_amd.RbvfitAmdError):        # theta index outside [0, D)
            e.add_instrument(g("wave"), g("flux"), g("inv_sigma2"), g("log_inv_sigma2"), g("lambda0"), g("gamma"),
                             g("f"), g("zfac"), g("N_idx") + 100, g("b_idx"), g("v_idx"))
        with pytest.raises(rbvfit_amd.RbvfitAmdError):        # even number of taps
            e.add_instrument(g("wave"), g("flux"), g("inv_sigma2"), g("log_inv_sigma2"), g("lambda0"), g("gamma"),
                             g("f"), g("zfac"), g("N_idx"), g("b_idx"), g("v_idx"), taps=[0.5, 0.5], lsf_mode=1)
        e.add_instrument(g("wave"), g("flux"), g("inv_sigma2"), g("log_inv_sigma2"), g("lambda0"), g("gamma"),
                         g("f"), g("zfac"), g("N_idx"), g("b_idx"), g("v_idx"), taps=g("taps"), lsf_mode=int(g("lsf_mode")))
        with pytest.raises(ValueError):
            e.lnprob(np.zeros((3, 5)))
        assert e.lnprob(np.zeros((0, 6))).shape == (0,)      # empty batch
        np.testing.assert_allclose(e.lnprob(z["thetas"])[:4], z["lnprob"][:4], rtol=LNPROB_RTOL, atol=LNPROB_ATOL)


def test_components_and_batched_gradient():
    """'Next' rows N3 (per-line unconvolved profiles, voigt_model.py:232-259) and N2 (batched
    finite-difference stencil for optimize_guess, vfit_mcmc.py:355-360)."""
    from oracle import voigt_oracle as vo
    from rbvfit_amd.model import FitConfiguration, VoigtModel
    from rbvfit_amd.vfit import vfit
    z = load_golden("c0_mgii")
    cfg = FitConfiguration(); cfg.add_system(0.348, "MgII", [2796.35, 2803.53], 2)
    model = VoigtModel(cfg, FWHM="6.5", normalize_kernel=False)        # fixture taps are astropy-4.3.1's (raw)
    cm = model.compile()
    th = z["thetas"][:3]
    comp = cm.components(th, z["G__wave"])
    assert comp.shape == (3, 4, 4096)
    data = vo.data_from_fixture(z, "G")
    for i in range(3):
        t = th[i]
        N = 10 ** t[data.N_indices]; b = t[data.b_indices]; v = t[data.v_indices]
        zt = data.z_factors * (1 + v / 299792.458) - 1
        wr = z["G__wave"][None, :] / (1 + zt[:, None])
        tau = vo.voigt_tau(data.atomic_lambda0, data.atomic_gamma, data.atomic_f, N, b, wr)
        np.testing.assert_allclose(comp[i], np.exp(-tau), rtol=0, atol=FLUX_ATOL)
        np.testing.assert_allclose(np.prod(comp[i], axis=0), cm.model_flux(t, z["G__wave"], convolved=False), rtol=0, atol=1e-12)
    # VoigtModel.evaluate(return_components=True): the reference's dictionary (voigt_model.py:232-259, 509-558)
    ev = model.evaluate(th[0], z["G__wave"], return_components=True)
    assert set(ev) == {"flux", "components", "component_info"} and len(ev["components"]) == 4
    np.testing.assert_array_equal(ev["flux"], cm.model_flux(th[0], z["G__wave"]))
    np.testing.assert_array_equal(np.array(ev["components"]), comp[0])
    t = th[0]
    for i, info in enumerate(ev["component_info"]):
        assert info["line_index"] == i and info["lambda0"] == float(data.atomic_lambda0[i])
        assert info["gamma"] == float(data.atomic_gamma[i]) and info["f_value"] == float(data.atomic_f[i])
        assert info["N_value"] == float(10 ** t[data.N_indices[i]]) and info["b_value"] == t[data.b_indices[i]]
        assert info["v_value"] == t[data.v_indices[i]]
        assert info["z_total"] == float(data.z_factors[i] * (1 + t[data.v_indices[i]] / 299792.458) - 1)
    np.testing.assert_array_equal(model.evaluate(th[0], z["G__wave"], return_unconvolved=True),
                                  cm.model_flux(th[0], z["G__wave"], convolved=False))
    inst = {"G": {"model": model, "wave": z["G__wave"], "flux": z["G__flux"], "error": z["G__error"]}}
    fit = vfit(inst, z["theta_true"], z["lb"], z["ub"])
    try:
        t0 = z["thetas"][1]
        f, g = fit.lnprob_and_grad(t0)
        singles = np.array([(fit.lnprob(t0 + 1e-8 * np.eye(6)[k]) - fit.lnprob(t0)) / 1e-8 for k in range(6)])
        assert f == fit.lnprob(t0)
        np.testing.assert_allclose(g, singles, rtol=0, atol=0)                  # same stencil, one batch
        best = fit.optimize_guess(t0)
        assert np.all(best >= z["lb"]) and np.all(best <= z["ub"])
        assert fit.lnprob(best) >= fit.lnprob(t0)
        # quick fit (vfit_mcmc.py:362-406, quick_fit_interface.py): chi2 objective, curvature errors
        insts = vo.instruments_from_fixture(z)
        rows = np.vstack([z["thetas"][:5], z["lb"][None, :] - 0.1])          # incl. a row outside the box
        ref_chi2 = np.array([sum(np.sum(((i.flux - vo.model_flux(i.data, r, i.wave)) / z["G__error"]) ** 2) for i in insts)
                             for r in rows])
        np.testing.assert_allclose(fit.chi2(rows), ref_chi2, rtol=1e-9)
        assert isinstance(fit.chi2(rows[0]), float)
        tb, terr = fit.fit_quick()
        assert tb.shape == (6,) and terr.shape == (6,) and np.all(terr > 0) and np.all(np.isfinite(terr))
        assert fit.chi2(tb) <= fit.chi2(np.asarray(fit.theta)) + 1e-6
        # errors = 1/sqrt(curvature): redo one axis serially with the oracle
        k, d = 2, max(abs(tb[2]) * 0.01, abs(fit.theta[2]) * 0.01, 1e-6)
        c = [sum(np.sum(((i.flux - vo.model_flux(i.data, tb + s * d * np.eye(6)[k], i.wave)) / z["G__error"]) ** 2)
                 for i in insts) for s in (0, 1, -1)]
        np.testing.assert_allclose(terr[k], np.sqrt(d * d / (c[1] - 2 * c[0] + c[2])), rtol=1e-5)
    finally:
        fit.close()


def test_curve_of_growth_grid_is_one_batch():
    """'Next' row N4 (compute_cog.py:23-183): EW(N, b) for HI Lya vs the oracle, incl. the damped regime."""
    from oracle import voigt_oracle as vo
    from rbvfit_amd.cog import compute_cog
    Nlist, blist = np.linspace(12.0, 21.0, 19), [5.0, 20.0, 60.0]
    cog = compute_cog(1215.67, Nlist, blist)
    assert cog.Wlist.shape == (19, 3) and cog.st["name"] == "HI 1215"
    d = vo.OracleModelData(np.array([1215.6701]), np.array([6.265e8], dtype=np.float32), np.array([0.4164], dtype=np.float32),
                           np.array([1.0]), np.array([0]), np.array([1]), np.array([2]), np.zeros(0), 0)
    trap = getattr(np, "trapezoid", None) or np.trapz
    for i in (0, 7, 13, 18):
        for j in range(3):
            fl = vo.model_flux(d, np.array([Nlist[i], blist[j], 0.0]), cog.wave)
            assert abs(cog.Wlist[i, j] - trap(1 - fl, x=cog.wave)) < 1e-11
    assert np.all(np.diff(cog.Wlist, axis=0) > 0)            # EW grows with N
    # the integral is a reduction on the device behind the model launch (vp_model_flux_rowsum): against the host's trapezoid
    # over the engine's own rows
    NN, BB = np.meshgrid(np.asarray(Nlist), np.asarray(blist), indexing="ij")
    theta = np.stack([NN.ravel(), BB.ravel(), np.zeros(NN.size)], axis=1)
    rows = cog.model_compiled.model_flux(theta, cog.wave)
    np.testing.assert_allclose(cog.Wlist.ravel(), trap(1.0 - rows, x=cog.wave, axis=1), rtol=0, atol=2e-13)
    assert isinstance(cog.model_compiled.equivalent_width(theta[5], cog.wave), float)
    cog.model_compiled.close()


@pytest.mark.parametrize("K", [65, 301, 1025, 2049])
def test_long_tabulated_kernels(K):
    """Long LSFs take wider tiles (2/4-wave workgroups, up to 8192 evaluated pixels per tile)."""
    import rbvfit_amd
    from oracle import voigt_oracle as vo
    from rbvfit_amd.model import FitConfiguration, VoigtModel
    rng = np.random.default_rng(K)
    cfg = FitConfiguration(); cfg.add_system(0.348, "MgII", [2796.35, 2803.53], 2)
    j = np.arange(K) - K // 2
    taps = np.exp(-0.5 * (j / (K / 9.0)) ** 2) * (1 + 0.3 * np.sin(j / 7.0) ** 2) + 1e-4
    model = VoigtModel(cfg, FWHM="6.5", kernel_taps=taps)
    data = model.compile().data
    wave = np.linspace(3755.0, 3795.0, 5000)
    th = np.array([[13.5, 13.2, 15.0, 25.0, -40.0, 20.0], [14.1, 12.9, 8.0, 31.0, -55.0, 12.0]])
    od = vo.OracleModelData(data.atomic_lambda0, data.atomic_gamma, data.atomic_f, data.z_factors, data.N_indices,
                            data.b_indices, data.v_indices, data.taps, data.lsf_mode, data.voigt_method)
    err = np.full(wave.size, 0.05)
    flux = vo.model_flux(od, th[0], wave) + rng.normal(0, 0.05, wave.size)
    inst = vo.OracleInstrument.from_error(od, wave, flux, err)
    lb, ub = th.min(0) - 5, th.max(0) + 5
    with rbvfit_amd.Engine(0) as e:
        e.set_bounds(lb, ub)
        e.add_instrument(wave, flux, inst.inv_sigma2, inst.log_inv_sigma2, **data.engine_kwargs())
        got = e.lnprob(th)
        fl = e.model_flux(0, th)
    np.testing.assert_allclose(got, vo.lnprob_batch(th, lb, ub, [inst]), rtol=LNPROB_RTOL, atol=LNPROB_ATOL)
    for i in range(2):
        np.testing.assert_allclose(fl[i], vo.model_flux(od, th[i], wave), rtol=0, atol=FLUX_ATOL)


def test_integration_md_binding_stub_runs_as_written():
    """INTEGRATION.md section 2 shows the ctypes stub a maintainer would add to rbvfit; execute that
    very text against the in-tree library with a duck-typed fitter (the attributes the stub reads:
    vfit.lb/ub/instrument_data, the bound model_flux's __self__.data = CompiledModelData fields,
    kernel.array) and compare with the golden lnprob."""
    import os
    import re
    import types
    from rbvfit_amd import _lib
    z = load_golden("c0_mgii")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    md = open(os.path.join(root, "INTEGRATION.md")).read()
    block = re.search(r"```python\n(# src/rbvfit/_amd_backend\.py.*?)```", md, flags=re.S).group(1)
    block = block.replace('C.CDLL("librbvfit_amd.so")', f'C.CDLL({_lib.LIB_PATH!r})')
    ns = {}
    exec(compile(block, "INTEGRATION.md", "exec"), ns)

    class Gaussian1DKernel:                                   # only the class name and .array are read
        def __init__(self, array):
            self.array = array

    g = lambda k: z[f"G__{k}"]
    data = types.SimpleNamespace(atomic_lambda0=g("lambda0"), atomic_gamma=g("gamma").astype(np.float32),
                                 atomic_f=g("f").astype(np.float32), z_factors=g("zfac"), N_indices=g("N_idx"),
                                 b_indices=g("b_idx"), v_indices=g("v_idx"), kernel=Gaussian1DKernel(g("taps")),
                                 n_lines=len(g("lambda0")), voigt_method="wofz")

    class Compiled:
        def __init__(self, d):
            self.data = d

        def model_flux(self, theta, wave):                   # never called: the engine replaces it
            raise AssertionError

    fitter = types.SimpleNamespace(lb=z["lb"], ub=z["ub"], instrument_data={
        "G": {"model": Compiled(data).model_flux, "wave": g("wave"), "flux": g("flux"),
              "inv_sigma2": g("inv_sigma2"), "log_inv_sigma2": g("log_inv_sigma2")}})
    post = ns["AmdPosterior"](fitter)
    quiet = ns["AmdPosterior"](fitter, prearm=0)             # the constructor argument a shared-GPU deployment passes
    np.testing.assert_array_equal(quiet(z["thetas"]), post(z["thetas"]))
    got = post(z["thetas"])
    fin = np.isfinite(z["lnprob"])
    np.testing.assert_allclose(got[fin], z["lnprob"][fin], rtol=LNPROB_RTOL, atol=LNPROB_ATOL)
    assert np.array_equal(np.isneginf(got), np.isneginf(z["lnprob"]))
    assert isinstance(post(z["thetas"][0]), float)


def test_every_launch_gives_the_same_bits_under_permutations():
    """1200 launches of C3 at its walker count (two instruments, one with 4-wave tile workgroups, far-field expansions) over
    random permutations of the same walkers: every row must come out with the same bits every time.  (A wave of a 4-wave
    tile once read the exp table before the first wave had staged it -- one wrong walker in ~700 launches.)"""
    from rbvfit_amd.workloads import make_workload
    wl = make_workload("C3", walkers=2048)
    try:
        th = wl.thetas.copy()
        th[5, 0] = wl.lb[0] - 0.25
        got = wl.engine.lnprob(th)
        assert wl.engine.last_launch_kind == "tiles+farfield"
        rng = np.random.default_rng(5)
        for _ in range(1200):
            perm = rng.permutation(th.shape[0])
            assert np.array_equal(wl.engine.lnprob(th[perm]), got[perm])
    finally:
        wl.engine.close()
