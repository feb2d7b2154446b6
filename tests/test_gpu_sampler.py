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
    ref = eng.lnprob(chain.reshape(-1, 6)).reshape(25, 48)
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
    import os
    os.environ["RBVFIT_AMD_NO_FUSED_ACCEPT"] = "1"            # separate accept / propose launches: same draws
    try:
        u = eng.stretch_run(p0, 20, seed=11)
    finally:
        del os.environ["RBVFIT_AMD_NO_FUSED_ACCEPT"]
    np.testing.assert_array_equal(u[2], a[2]); np.testing.assert_array_equal(u[4], a[4])
    nochain = eng.stretch_run(p0, 20, seed=11, store_chain=False)
    assert nochain[2] is None
    np.testing.assert_array_equal(nochain[0], a[0])


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
    np.testing.assert_array_equal(clp, wl.engine.lnprob(chain.reshape(-1, 24)).reshape(6, 64))
    assert np.all(chain >= wl.lb) and np.all(chain <= wl.ub) and nacc.sum() > 0
    wl = make_workload("C1", walkers=1100, pixels=256)
    pos, lp, chain, clp, nacc = wl.engine.stretch_run(wl.thetas, 4, seed=3)
    np.testing.assert_array_equal(clp[-1], wl.engine.lnprob(chain[-1]))
    full = np.concatenate([wl.thetas[None], chain])
    np.testing.assert_array_equal(np.any(full[1:] != full[:-1], axis=2).sum(axis=0), nacc)
