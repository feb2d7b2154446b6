"""Device-resident stretch-move sampler (vp_stretch_run; SURVEY 8f N1: the walker loop the reference
delegates to emcee, vfit_mcmc.py:408-423, 536-540).  Chains are never compared with the reference
(trap T17: its walker init is unseeded); what is checked is that every stored lnprob IS the
posterior of the stored position, the move's bookkeeping, reproducibility, and that the sampled
distribution agrees with the host sampler's."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _workload(W=48, pixels=512):
    from rbvfit_amd.workloads import make_workload
    return make_workload("C1", walkers=W, pixels=pixels)


def test_chain_lnprob_is_the_posterior_of_the_chain_positions():
    wl = _workload()
    eng, p0 = wl.engine, wl.thetas
    pos, lp, chain, clp, nacc = eng.stretch_run(p0, 25, seed=3)
    assert chain.shape == (25, 48, 6) and clp.shape == (25, 48)
    # the half-steps ran in the one-launch walker kernel, whose tiles are the full-size ones: the same tiles give the
    # same bits whatever the batch; the small tiles a 1200-row batch of this 512-pixel spectrum would get by default
    # group the chi^2 sum differently (last-bit differences)
    ref_default = eng.lnprob(chain.reshape(-1, 6)).reshape(25, 48)
    np.testing.assert_allclose(clp, ref_default, rtol=1e-14, atol=0)
    eng.set_option("geom", 0)
    ref = eng.lnprob(chain.reshape(-1, 6)).reshape(25, 48)
    eng.set_option("geom", -1)
    np.testing.assert_array_equal(clp, ref)                      # same kernels, any batch composition
    np.testing.assert_array_equal(pos, chain[-1]); np.testing.assert_array_equal(lp, clp[-1])
    assert np.all(chain >= wl.lb) and np.all(chain <= wl.ub)    # out-of-box proposals are never accepted
    # bookkeeping: a walker's position changes exactly when a proposal was accepted
    full = np.concatenate([p0[None], chain])
    moved = np.any(full[1:] != full[:-1], axis=2).sum(axis=0)
    np.testing.assert_array_equal(moved, nacc)
    assert 0.05 < nacc.mean() / 25 < 0.95


def test_runs_are_reproducible_and_splittable():
    wl = _workload()
    eng, p0 = wl.engine, wl.thetas
    a = eng.stretch_run(p0, 20, seed=11)
    b = eng.stretch_run(p0, 20, seed=11)
    np.testing.assert_array_equal(a[2], b[2])
    h1 = eng.stretch_run(p0, 8, seed=11)
    h2 = eng.stretch_run(h1[0], 12, lnprob=h1[1], seed=11, step0=8, naccepted=h1[4])
    np.testing.assert_array_equal(np.concatenate([h1[2], h2[2]]), a[2])
    np.testing.assert_array_equal(h2[4], a[4])
    c = eng.stretch_run(p0, 20, seed=12)
    assert not np.array_equal(c[2], a[2])
    eng.set_option("no_fused_accept", 1)                     # separate accept / propose launches: same draws
    u = eng.stretch_run(p0, 20, seed=11)
    eng.set_option("no_fused_accept", 0)
    np.testing.assert_array_equal(u[2], a[2]); np.testing.assert_array_equal(u[4], a[4])
    nochain = eng.stretch_run(p0, 20, seed=11, store_chain=False)
    assert nochain[2] is None
    np.testing.assert_array_equal(nochain[0], a[0])


@pytest.mark.parametrize("W,pixels,nsteps", [(48, 512, 31), (512, 4096, 12), (6, 300, 40)])
def test_overlapped_half_steps_do_not_show_in_the_chain(W, pixels, nsteps):
    """vp_stretch_run puts consecutive half-steps on two streams and lets every walker's workgroup wait for ITS partner's
    version word instead of the whole launch before (StretchArgs::ovl; rows double-buffered by update count).  Forced on,
    forced off and the automatic choice give the same chain, stored lnprob, final state and acceptance counts, bit for
    bit -- with odd and even step counts (the final rows live in either buffer), split runs, chain chunks, and when a
    proposal's lnprob is NaN the error is the same."""
    wl = _workload(W, pixels)
    eng, p0 = wl.engine, wl.thetas
    runs = {}
    for mode in (0, 1, -1):
        eng.set_option("stretch_overlap", mode)
        a = eng.stretch_run(p0, nsteps, seed=5)
        h1 = eng.stretch_run(p0, nsteps // 2, seed=5)
        h2 = eng.stretch_run(h1[0], nsteps - nsteps // 2, lnprob=h1[1], seed=5, step0=nsteps // 2, naccepted=h1[4])
        np.testing.assert_array_equal(np.concatenate([h1[2], h2[2]]), a[2])
        np.testing.assert_array_equal(h2[0], a[0]); np.testing.assert_array_equal(h2[4], a[4])
        nochain = eng.stretch_run(p0, nsteps, seed=5, store_chain=False)
        np.testing.assert_array_equal(nochain[0], a[0]); np.testing.assert_array_equal(nochain[1], a[1])
        runs[mode] = a
    for mode in (1, -1):
        for k in (0, 1, 2, 3, 4):
            np.testing.assert_array_equal(runs[mode][k], runs[0][k])
    np.testing.assert_array_equal(runs[0][0], runs[0][2][-1])
    eng.set_option("stretch_overlap", -1)
    wl.engine.close()


def test_sampled_distribution_agrees_with_the_host_sampler():
    from rbvfit_amd.sampler import DeviceStretchSampler, StretchMoveSampler
    wl = _workload()
    eng, p0 = wl.engine, wl.thetas
    dev = DeviceStretchSampler(48, 6, eng, seed=1)
    dev.run_mcmc(p0, 1000)
    dev.run_mcmc(dev.chain[-1], 3000, lnprob0=dev.lnprobability[-1])
    host = StretchMoveSampler(48, 6, eng.lnprob, seed=2)
    host.run_mcmc(p0, 4000)
    d, h = dev.get_chain(discard=1000, flat=True), host.get_chain(discard=1000, flat=True)
    sd = h.std(axis=0)
    assert np.all(np.abs(d.mean(axis=0) - h.mean(axis=0)) < 0.5 * sd)
    assert np.all(d.std(axis=0) / sd > 0.6) and np.all(d.std(axis=0) / sd < 1.6)
    assert np.all(np.abs(d.mean(axis=0) - wl.theta_true) < 6 * sd)
    assert abs(dev.acceptance_fraction.mean() - host.acceptance_fraction.mean()) < 0.15
    assert dev.chain.shape == (4000, 48, 6) and dev.nsteps == 4000


def test_argument_errors_and_nan_posterior():
    import rbvfit_amd
    wl = _workload(W=16, pixels=128)
    eng = wl.engine
    with pytest.raises(rbvfit_amd.RbvfitAmdError):
        eng.stretch_run(wl.thetas[:15], 2)                       # odd number of walkers
    with pytest.raises(rbvfit_amd.RbvfitAmdError):
        eng.stretch_run(wl.thetas, 2, a=1.0)
    with pytest.raises(ValueError):
        eng.stretch_run(wl.thetas[0], 2)
    wave, flux, err = wl.spectra[0]
    bad = flux.copy(); bad[5] = np.nan                           # NaN flux -> NaN lnprob (trap T7)
    eng.update_spectrum(0, bad, 1.0 / err ** 2, np.log(1.0 / err ** 2))
    with pytest.raises(ValueError, match="NaN"):
        eng.stretch_run(wl.thetas, 2)
    eng.update_spectrum(0, flux, 1.0 / err ** 2, np.log(1.0 / err ** 2))
    assert np.all(np.isfinite(eng.stretch_run(wl.thetas, 2)[1]))


def test_vfit_runmcmc_on_the_device():
    from rbvfit_amd.model import FitConfiguration, VoigtModel
    from rbvfit_amd.vfit import vfit
    wl = _workload(W=16, pixels=512)
    wave, flux, err = wl.spectra[0]
    cfg = FitConfiguration(); cfg.add_system(0.348, "MgII", [2796.35, 2803.53], 2)
    fit = vfit({"G": {"model": VoigtModel(cfg, FWHM="6.5"), "wave": wave, "flux": flux, "error": err}},
               wl.theta_true, wl.lb, wl.ub, no_of_Chain=24, no_of_steps=40)
    try:
        s = fit.runmcmc(seed=4, sampler="device")
        assert s.chain.shape == (40, 24, 6) and fit.samples.shape == (32 * 24, 6)
        np.testing.assert_array_equal(s.lnprobability[-1], fit.lnprob(s.chain[-1]))
        with pytest.raises(ValueError):
            fit.runmcmc(sampler="nope")
    finally:
        fit.close()


def test_slice_sampler_ragged_batches_on_the_engine():
    """sampler='zeus' in the reference (vfit_mcmc.py:425-440): ensemble slice sampling calls lnprob
    with shrinking, ragged batches; every stored lnprob must be the engine's value for the stored
    position and the distribution must agree with the stretch move's."""
    from rbvfit_amd.model import FitConfiguration, VoigtModel
    from rbvfit_amd.sampler import EnsembleSliceSampler
    from rbvfit_amd.vfit import vfit
    wl = _workload()
    eng, p0 = wl.engine, wl.thetas
    sl = EnsembleSliceSampler(48, 6, eng.lnprob, seed=3)
    sl.run_mcmc(p0, 600)
    assert min(sl.batch_sizes) == 1 and max(sl.batch_sizes[1:]) == 24 and len(set(sl.batch_sizes)) > 10   # [0]: the initial 48
    np.testing.assert_array_equal(sl.lnprobability[-1], eng.lnprob(sl.chain[-1]))
    dev = eng.stretch_run(p0, 4000, seed=1)[2][1000:].reshape(-1, 6)
    s = sl.get_chain(discard=200, flat=True)
    sd = dev.std(axis=0)
    assert np.all(np.abs(s.mean(axis=0) - dev.mean(axis=0)) < 0.5 * sd)
    assert np.all(s.std(axis=0) / sd > 0.6) and np.all(s.std(axis=0) / sd < 1.6)
    assert np.all(s >= wl.lb) and np.all(s <= wl.ub)
    # the reference's constructor switch: sampler='zeus' -> slice sampling (zeus itself when installed)
    wave, flux, err = wl.spectra[0]
    cfg = FitConfiguration(); cfg.add_system(0.348, "MgII", [2796.35, 2803.53], 2)
    fit = vfit({"G": {"model": VoigtModel(cfg, FWHM="6.5"), "wave": wave, "flux": flux, "error": err}},
               wl.theta_true, wl.lb, wl.ub, no_of_Chain=24, no_of_steps=30, sampler="zeus")
    try:
        smp = fit.runmcmc(seed=2)
        assert type(smp).__name__ in ("EnsembleSliceSampler", "EnsembleSampler")
        assert fit.samples.shape == (24 * 24, 6)
    finally:
        fit.close()


def test_island_ensemble_with_the_device_sampler():
    """One island per rank, each running vp_stretch_run on its own GPU (here: a single rank)."""
    from rbvfit_amd.dist import IslandEnsemble
    wl = _workload(W=24, pixels=256)
    isl = IslandEnsemble(None, 24, 6, seed=5, engine=wl.engine)
    isl.run_mcmc(wl.thetas, 30)
    chain, lnp = isl.gather_chain(discard=10)
    assert chain.shape == (20, 24, 6) and lnp.shape == (20, 24)
    np.testing.assert_array_equal(lnp[-1], wl.engine.lnprob(chain[-1]))
    assert isl.island_means().shape == (1, 6)


def test_device_sampler_multi_instrument_and_large_ensembles():
    """Two instruments (C3: tabulated + Gaussian LSF, 24 parameters) and an ensemble above the
    1024-walker limit of the fused accept/propose launch."""
    from rbvfit_amd.workloads import make_workload
    wl = make_workload("C3", walkers=64, pixels=512)
    pos, lp, chain, clp, nacc = wl.engine.stretch_run(wl.thetas, 6, seed=2)
    # (the 101-tap instrument: 32-row half-steps run as launches with 4-wave tiles, a 384-row batch in the walker kernel
    # with single-wave tiles -- same values to rounding; the same launch structure gives the same bits)
    np.testing.assert_allclose(clp, wl.engine.lnprob(chain.reshape(-1, 24)).reshape(6, 64), rtol=1e-14, atol=0)
    wl.engine.set_option("walker", 0)
    np.testing.assert_array_equal(clp, wl.engine.lnprob(chain.reshape(-1, 24)).reshape(6, 64))
    wl.engine.set_option("walker", -1)
    assert np.all(chain >= wl.lb) and np.all(chain <= wl.ub) and nacc.sum() > 0
    wl = make_workload("C1", walkers=1100, pixels=256)
    pos, lp, chain, clp, nacc = wl.engine.stretch_run(wl.thetas, 4, seed=3)
    np.testing.assert_array_equal(clp[-1], wl.engine.lnprob(chain[-1]))
    full = np.concatenate([wl.thetas[None], chain])
    np.testing.assert_array_equal(np.any(full[1:] != full[:-1], axis=2).sum(axis=0), nacc)


# ---- independent host replay of vp_stretch_run (propose / accept kernels, csrc/sampler_kernels.h) ----------
def _philox_np(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10 over arrays of counters (numpy uint64 arithmetic; same function as tests/test_host_logic.py's
    scalar restatement, which is pinned by the Random123 known-answer vectors)."""
    M0, M1, W0, W1, mask = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), 0x9E3779B9, 0xBB67AE85, np.uint64(0xFFFFFFFF)
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) for c in np.broadcast_arrays(c0, c1, c2, c3))
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        c0, c1, c2, c3 = (p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0), p1 & mask, (p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1), p0 & mask
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def _u01(hi, lo):
    return (((hi << np.uint64(32)) | lo) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def _draw(seed, step, half, walkers, purpose):
    """counter layout of sampler_kernels.h::draw: (walker, step low, step high | half << 31, purpose), key = seed."""
    return _philox_np(walkers, step & 0xFFFFFFFF, (step >> 32) | (half << 31), purpose, seed & 0xFFFFFFFF, seed >> 32)


def _replay_stretch(lnprob, p0, nsteps, a, seed, step0=0):
    """The reference's emcee stretch move (vfit_mcmc.py:408-423, 536-540) as vp_stretch_run performs it, in NumPy:
    red-blue halves, z = ((a-1)u+1)^2/a, partner j = floor(u2 nC), Y = X_j - (X_j - X_k) z, accept when
    ln u < (D-1) ln z + lnprob(Y) - lnprob(X).  Only lnprob comes from the engine."""
    pos = np.array(p0, dtype=np.float64)
    W, D = pos.shape
    half = W // 2
    lp = lnprob(pos)
    nacc = np.zeros(W, dtype=np.int64)
    chain, clp = np.empty((nsteps, W, D)), np.empty((nsteps, W))
    for it in range(nsteps):
        step = step0 + it
        for h in (0, 1):
            s0, c0 = (half, 0) if h else (0, half)
            wk = np.arange(s0, s0 + half)
            r = _draw(seed, step, h, wk, 0)
            u1, u2 = _u01(r[0], r[1]), _u01(r[2], r[3])
            t = (a - 1.0) * u1 + 1.0
            z = t * t / a
            j = np.minimum((u2 * float(half)).astype(np.int64), half - 1)
            x, c = pos[wk], pos[c0 + j]
            prop = c - (c - x) * z[:, None]
            lpn = lnprob(prop)
            assert not np.any(np.isnan(lpn))
            r = _draw(seed, step, h, wk, 1)
            u = _u01(r[0], r[1])
            with np.errstate(divide="ignore"):
                acc = np.log(u) < (D - 1.0) * np.log(z) + lpn - lp[wk]
            pos[wk[acc]], lp[wk[acc]] = prop[acc], lpn[acc]
            nacc[wk[acc]] += 1
        chain[it], clp[it] = pos, lp
    return pos, lp, chain, clp, nacc


# launch structures of a half-step: accept+propose kernel + lnprob launches (W <= 1024), separate accept / propose
# launches (W > 1024), and the WHOLE half-step in one walker_kernel launch (proposal, lnprob and accept inside each
# walker's workgroup; automatic for half-ensembles of ~100-256 and ~410-512 walkers, forced here)
@pytest.mark.parametrize("W,nsteps,one_launch", [(48, 40, False), (1100, 6, False), (48, 40, True), (512, 4, None)])
def test_device_sampler_equals_an_independent_host_replay(W, nsteps, one_launch):
    """N1 pinned independently of the sampler kernels: same Philox draws, proposals and accept/reject restated in
    NumPy, lnprob of each half-ensemble proposal block from Engine.lnprob -- chain, stored lnprob, acceptance
    counts and final state must be bit-identical, also across a split run (step0)."""
    wl = _workload(W=W)
    eng, p0 = wl.engine, wl.thetas
    if one_launch is not None:
        eng.set_option("walker", 1 if one_launch else 0)
    seed = 0x1234_5678_9ABC_DEF1
    dev = eng.stretch_run(p0, nsteps, seed=seed, a=2.0)
    ref = _replay_stretch(eng.lnprob, p0, nsteps, 2.0, seed)
    for d, r in zip(dev, ref):
        np.testing.assert_array_equal(d, r)
    assert ref[4].sum() > 0
    # second leg continuing the stream from step0 with another stretch scale
    dev2 = eng.stretch_run(dev[0], 3, lnprob=dev[1], seed=seed, a=1.7, step0=nsteps)
    pos, lp, chain, clp, nacc = _replay_stretch(lambda th: eng.lnprob(th), dev[0], 3, 1.7, seed, step0=nsteps)
    np.testing.assert_array_equal(dev2[2], chain); np.testing.assert_array_equal(dev2[3], clp)
    np.testing.assert_array_equal(dev2[4], nacc)


def test_initial_nan_lnprob_is_an_error():
    """emcee and the host sampler raise when the probability function returns NaN; a walker that STARTS at NaN
    would otherwise stay frozen for the whole run (ln u < NaN is never true)."""
    wl = _workload()
    lp = wl.engine.lnprob(wl.thetas)
    lp[3] = np.nan
    with pytest.raises(ValueError, match="NaN"):
        wl.engine.stretch_run(wl.thetas, 2, lnprob=lp, seed=1)


# ---- independent host replay of vp_slice_run (csrc/slice_kernels.h) ----------------------------------------
def _replay_slice(lnprob, p0, nsteps, seed, mu=1.0, tune=True, tolerance=0.05, patience=5, maxsteps=10000, step0=0):
    """zeus' ensemble slice sampling (the reference's sampler='zeus', vfit_mcmc.py:425-440) restated in NumPy with
    the Philox draws of vp_slice_run: random split by ranking 64-bit keys, differential move, per-walker state
    machine OUT_L -> OUT_R -> SHRINK -> DONE, ONE trial point per walker and round.  The device evaluates several
    candidate points per walker and round (both edges at once, shrink draws ahead of their turn) and must end up with
    exactly what this one-at-a-time procedure produces, its count of evaluations included."""
    pos = np.array(p0, dtype=np.float64)
    W, D = pos.shape
    half = W // 2
    lp = lnprob(pos)
    gamma0 = 2.38 / np.sqrt(2.0 * D)
    good, n_evals = 0, 0
    chain, clp, hist = np.empty((nsteps, W, D)), np.empty((nsteps, W)), np.empty(nsteps)
    allw = np.arange(W)
    for it in range(nsteps):
        step = step0 + it
        r = _draw(seed, step, 0, allw, 16)
        keys = (r[0] << np.uint64(32)) | r[1]
        perm = np.lexsort((allw, keys))                       # rank by key, ties by index
        nexp = ncon = 0
        for h in (0, 1):
            S, Cc = perm[h * half:(h + 1) * half], perm[(1 - h) * half:(2 - h) * half]
            ra, rb, rc = _draw(seed, step, h, S, 17), _draw(seed, step, h, S, 18), _draw(seed, step, h, S, 19)
            l = np.minimum((_u01(ra[0], ra[1]) * float(half)).astype(np.int64), half - 1)
            mo = np.minimum((_u01(ra[2], ra[3]) * float(half - 1)).astype(np.int64), half - 2)
            m = (l + 1 + mo) % half
            X0 = pos[S].copy()
            eta = (mu * gamma0) * (pos[Cc[l]] - pos[Cc[m]])
            Z0 = lp[S] + np.log(_u01(rb[0], rb[1]))
            L = -_u01(rb[2], rb[3]); R = L + 1.0
            J = np.minimum((float(maxsteps) * _u01(rc[0], rc[1])).astype(np.int64), maxsteps - 1)
            K = (maxsteps - 1) - J
            phase = np.zeros(half, dtype=int); nshr = np.zeros(half, dtype=np.int64); Wd = np.zeros(half)
            while True:
                phase[(phase == 0) & (J <= 0)] = 1
                phase[(phase == 1) & (K <= 0)] = 2
                act = np.flatnonzero(phase != 3)
                if act.size == 0:
                    break
                t = np.where(phase[act] == 0, L[act], R[act])
                sh = act[phase[act] == 2]
                for k in sh:                                   # the c-th shrink draw of walker S[k]: purpose 32 + c
                    rr = _draw(seed, step, h, S[k:k + 1], 32 + int(nshr[k]))
                    Wd[k] = L[k] + _u01(rr[0], rr[1])[0] * (R[k] - L[k])
                t = np.where(phase[act] == 2, Wd[act], t)
                batch = np.full((half, D), np.inf)
                batch[:act.size] = X0[act] + t[:, None] * eta[act]
                v = lnprob(batch)[:act.size]
                n_evals += act.size
                assert not np.any(np.isnan(v))
                up = v > Z0[act]
                for k, ok, val, tk in zip(act, up, v, t):
                    if phase[k] == 0:
                        if ok: L[k] -= 1.0; J[k] -= 1; nexp += 1
                        else: phase[k] = 1
                    elif phase[k] == 1:
                        if ok: R[k] += 1.0; K[k] -= 1; nexp += 1
                        else: phase[k] = 2
                    else:
                        if ok:
                            pos[S[k]] = X0[k] + tk * eta[k]; lp[S[k]] = val; phase[k] = 3
                        else:
                            if tk < 0: L[k] = tk
                            else: R[k] = tk
                            nshr[k] += 1; ncon += 1
        if tune:
            ratio = 2.0 * max(1, nexp) / (max(1, nexp) + ncon)
            mu *= ratio
            good = good + 1 if abs(ratio - 1.0) < tolerance else 0
            if good >= patience:
                tune = False
        hist[it] = mu
        chain[it], clp[it] = pos, lp
    return dict(pos=pos, lnprob=lp, chain=chain, chain_lnprob=clp, mu=mu, tune=tune, mu_history=hist, n_evals=n_evals)


@pytest.mark.parametrize("W,nsteps", [(16, 12), (48, 25), (2600, 6)])     # (2600: 1300 walkers per half, two per thread of the control kernels)
def test_device_slice_sampler_equals_an_independent_host_replay(W, nsteps):
    """N1, second move: vp_slice_run against a NumPy restatement that shares only Engine.lnprob and the Philox
    function -- chain, stored lnprob, mu trajectory and the number of lnprob evaluations must be identical, also
    across a split run."""
    wl = _workload(W=W)
    eng, p0 = wl.engine, wl.thetas
    seed = 0xC0FFEE_1234_5678
    dev = eng.slice_run(p0, nsteps, seed=seed)
    ref = _replay_slice(eng.lnprob, p0, nsteps, seed)
    for k in ("chain", "chain_lnprob", "pos", "lnprob", "mu_history"):
        np.testing.assert_array_equal(dev[k], ref[k], err_msg=k)
    assert dev["n_evals"] == ref["n_evals"] and dev["mu"] == ref["mu"] and dev["tune"] == ref["tune"]
    assert np.all(dev["chain"] >= wl.lb) and np.all(dev["chain"] <= wl.ub)
    np.testing.assert_array_equal(eng.lnprob(dev["chain"][-1]), dev["chain_lnprob"][-1])
    # every walker moves every iteration (slice sampling has no rejections)
    full = np.concatenate([p0[None], dev["chain"]])
    assert np.all(np.any(full[1:] != full[:-1], axis=2))
    # continuation: run(n1) then run(n2, step0=n1, mu, tune) == run(n1+n2)
    a = eng.slice_run(p0, 5, seed=seed)
    b = eng.slice_run(a["pos"], nsteps - 5, lnprob=a["lnprob"], seed=seed, step0=5, mu=a["mu"], tune=a["tune"])
    np.testing.assert_array_equal(np.concatenate([a["chain"], b["chain"]])[:5], dev["chain"][:5])
    np.testing.assert_array_equal(b["chain"][0], dev["chain"][5])      # (the patience counter restarts: compare the first step)


def test_device_slice_sampler_split_runs_carry_the_tuning_state():
    """run(6) then run(6) is run(12) bit for bit WHILE mu is being tuned: the count of consecutive in-tolerance iterations
    travels from call to call (vp_slice_run's `tune` in/out), so tuning switches off at the same iteration either way."""
    from rbvfit_amd.sampler import DeviceSliceSampler
    wl = _workload()
    eng, p0 = wl.engine, wl.thetas
    kw = dict(seed=11, tolerance=0.6, patience=8)            # loose tolerance: the streak builds up across the split
    one = DeviceSliceSampler(48, 6, eng, **kw)
    one.run_mcmc(p0, 12)
    two = DeviceSliceSampler(48, 6, eng, **kw)
    two.run_mcmc(p0, 6)
    assert two.tune and two._tune_state > 1                  # (the split falls inside a streak)
    two.run_mcmc(two.chain[-1], 6, lnprob0=two.lnprobability[-1])
    assert not one.tune and not two.tune                     # both switched tuning off ...
    np.testing.assert_array_equal(one.mu_history, two.mu_history)      # ... at the same iteration
    np.testing.assert_array_equal(one.chain, two.chain)
    np.testing.assert_array_equal(one.lnprobability, two.lnprobability)
    assert one.n_lnprob_evals == two.n_lnprob_evals


def test_device_slice_sampler_segments_do_not_show_in_the_chain():
    """The device works through a SEGMENT of iterations on its own (slice_round_kernel decides what every next batch is; the
    host only enqueues launches and collects the chain behind the segment).  Cutting a run into segments of 5 or of 1
    iteration ("slice_seg") must not change a bit of it -- chain, lnprob, mu history, evaluation count."""
    wl = _workload()
    eng, p0 = wl.engine, wl.thetas
    ref = eng.slice_run(p0, 12, seed=21)
    try:
        for seg in (5, 1):
            eng.set_option("slice_seg", seg)
            got = eng.slice_run(p0, 12, seed=21)
            for key in ("chain", "chain_lnprob", "pos", "lnprob", "mu_history"):
                np.testing.assert_array_equal(got[key], ref[key], err_msg=f"{key} with segments of {seg}")
            assert got["n_evals"] == ref["n_evals"] and got["mu"] == ref["mu"] and got["tune_state"] == ref["tune_state"]
        # no chain kept: the same final state
        eng.set_option("slice_seg", 4)
        got = eng.slice_run(p0, 12, seed=21, store_chain=False)
        np.testing.assert_array_equal(got["pos"], ref["pos"])
        np.testing.assert_array_equal(got["lnprob"], ref["lnprob"])
    finally:
        eng.set_option("slice_seg", 0)


def test_device_slice_sampler_distribution_and_vfit_switch():
    from rbvfit_amd.model import FitConfiguration, VoigtModel
    from rbvfit_amd.sampler import DeviceSliceSampler
    from rbvfit_amd.vfit import vfit
    wl = _workload()
    eng, p0 = wl.engine, wl.thetas
    sl = DeviceSliceSampler(48, 6, eng, seed=3)
    sl.run_mcmc(p0, 200)
    sl.run_mcmc(sl.chain[-1], 400, lnprob0=sl.lnprobability[-1])
    assert sl.chain.shape == (600, 48, 6) and len(sl.mu_history) == 600 and sl.n_lnprob_evals > 600 * 48 * 3
    dev = eng.stretch_run(p0, 4000, seed=1)[2][1000:].reshape(-1, 6)
    s = sl.get_chain(discard=200, flat=True)
    sd = dev.std(axis=0)
    assert np.all(np.abs(s.mean(axis=0) - dev.mean(axis=0)) < 0.5 * sd)
    assert np.all(s.std(axis=0) / sd > 0.6) and np.all(s.std(axis=0) / sd < 1.6)
    wave, flux, err = wl.spectra[0]
    cfg = FitConfiguration(); cfg.add_system(0.348, "MgII", [2796.35, 2803.53], 2)
    fit = vfit({"G": {"model": VoigtModel(cfg, FWHM="6.5"), "wave": wave, "flux": flux, "error": err}},
               wl.theta_true, wl.lb, wl.ub, no_of_Chain=24, no_of_steps=30, sampler="zeus")
    try:
        smp = fit.runmcmc(seed=2, sampler="device-slice")
        assert type(smp).__name__ == "DeviceSliceSampler" and smp.chain.shape == (30, 24, 6)
        np.testing.assert_array_equal(smp.lnprobability[-1], fit.lnprob(smp.chain[-1]))
    finally:
        fit.close()
    lp = eng.lnprob(p0); lp[2] = -np.inf
    with pytest.raises(ValueError):
        eng.slice_run(p0, 2, lnprob=lp, seed=1)


def _multi_for(wl, ids):
    """A MultiEngine holding the workload's instrument(s) on the listed devices."""
    import rbvfit_amd
    m = rbvfit_amd.MultiEngine(ids)
    m.set_bounds(wl.lb, wl.ub)
    for data, (wave, flux, err) in zip(wl.tables, wl.spectra):
        w = 1.0 / err ** 2
        m.add_instrument(wave, flux, w, np.log(w), **data.engine_kwargs())
    return m


@pytest.mark.parametrize("walker,sync", [(1, 1), (0, 1), (1, 0), (0, 0)])
def test_sharded_ensemble_reproduces_the_single_context_chain(walker, sync):
    """vp_multi_stretch_run (BASELINE config 4's shape: ONE ensemble over several device contexts): device 0 listed two
    and three times -- blocks of 13 + 12 and 9 + 9 + 7 rows per half-step, W/2 not divisible by G -- gives the chain of
    vp_stretch_run on one context bit for bit, positions, lnprob, acceptance counts and all, through the one-launch
    walker kernel (walker = 1) and through propose / lnprob / accept launches (walker = 0); with the half-steps ordered by
    flags polled inside the kernels (sync = 1, the default when the contexts share a device) and by events between the
    contexts' streams (sync = 0); runs can be split into calls."""
    wl = _workload(W=50, pixels=700)
    eng, p0 = wl.engine, wl.thetas
    eng.set_option("walker", walker)
    ref = eng.stretch_run(p0, 24, seed=5)
    assert 0.1 < ref[4].sum() / (24 * 50) < 0.9          # the chain moves
    for ids in ([0, 0], [0, 0, 0]):
        with _multi_for(wl, ids) as m:
            m.set_option("walker", walker)
            m.set_option("multi_sync", sync)
            got = m.stretch_run(p0, 24, seed=5)
            for a, b in zip(ref, got):
                np.testing.assert_array_equal(a, b)
            # split into two calls, the second one with the state and lnprob of the first
            first = m.stretch_run(p0, 10, seed=5, store_chain=False)
            second = m.stretch_run(first[0], 14, lnprob=first[1], seed=5, step0=10, naccepted=first[4])
            np.testing.assert_array_equal(second[0], ref[0])
            np.testing.assert_array_equal(second[1], ref[1])
            np.testing.assert_array_equal(second[2], ref[2][10:])
            np.testing.assert_array_equal(second[4], ref[4])
            bad = np.array(p0); bad[3, 1] = np.nan
            with pytest.raises(ValueError, match="NaN"):
                m.stretch_run(bad, 2, seed=1)
            with pytest.raises(Exception, match="even"):
                m.stretch_run(p0[:49], 2, seed=1)


@pytest.mark.parametrize("walker,sync", [(1, 0), (0, 0), (1, 1)])
def test_ensemble_sharded_over_eight_contexts(walker, sync):
    """The same at the size of the node the scaling bench runs on: device 0 listed EIGHT times -- 25 rows per
    half-step in blocks of 4, 4, 4, 4, 4, 4, 1 and 0 (a context without rows still takes part in every barrier), eight replicas
    written through eight sets of pointers, eight flags / events per half-step -- and the slice sampler's rounds cut eight ways.
    Chains, lnprob, acceptance counts, mu history and evaluation counts of the single-context runs, bit for bit."""
    wl = _workload(W=50, pixels=700)
    eng, p0 = wl.engine, wl.thetas
    eng.set_option("walker", walker)
    ref = eng.stretch_run(p0, 16, seed=5)
    with _multi_for(wl, [0] * 8) as m:
        assert m.n_devices == 8
        m.set_option("walker", walker)
        m.set_option("multi_sync", sync)
        got = m.stretch_run(p0, 16, seed=5)
        for a, b in zip(ref, got):
            np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(m.lnprob(p0), eng.lnprob(p0))          # 50 rows in blocks of 7, 7, ..., 1
        np.testing.assert_array_equal(m.lnprob(p0[:5]), eng.lnprob(p0[:5]))  # fewer rows than contexts
        if sync == 0:
            sref = eng.slice_run(p0, 6, seed=9)
            sgot = m.slice_run(p0, 6, seed=9)
            for k in ("pos", "lnprob", "chain", "chain_lnprob", "mu_history"):
                np.testing.assert_array_equal(sref[k], sgot[k], err_msg=k)
            assert (sref["mu"], sref["tune_state"], sref["n_evals"]) == (sgot["mu"], sgot["tune_state"], sgot["n_evals"])
    wl.engine.close()


@pytest.mark.parametrize("walker", [1, 0])
def test_sharded_slice_sampler_reproduces_the_single_context_chain(walker):
    """vp_multi_slice_run: replicated sampler state, every round's lnprob batch cut into one block of trial rows per
    context (device 0 listed two and three times: 25 + 25 and 17 + 17 + 16 rows) -- chain, lnprob, mu history and the
    count of evaluations of vp_slice_run on one context, bit for bit."""
    wl = _workload(W=50, pixels=700)
    eng, p0 = wl.engine, wl.thetas
    eng.set_option("walker", walker)
    ref = eng.slice_run(p0, 12, seed=9)
    for ids in ([0, 0], [0, 0, 0]):
        with _multi_for(wl, ids) as m:
            m.set_option("walker", walker)
            got = m.slice_run(p0, 12, seed=9)
            for k in ("pos", "lnprob", "chain", "chain_lnprob", "mu_history"):
                np.testing.assert_array_equal(ref[k], got[k], err_msg=k)
            assert (ref["mu"], ref["tune_state"], ref["n_evals"]) == (got["mu"], got["tune_state"], got["n_evals"])
            lp = eng.lnprob(p0); lp[2] = -np.inf
            with pytest.raises(ValueError, match="not finite"):
                m.slice_run(p0, 2, lnprob=lp, seed=1)
