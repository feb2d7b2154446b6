"""Seeded random configurations (lines, components, redshifts, grids, LSFs, theta) on the GPU vs the
oracle: exercises tier boundaries, mixed chunks, clusters of any size, descending / non-uniform
wavelength grids, very narrow and very broad lines, saturated and damped profiles."""
import os

import numpy as np
import pytest

from conftest import FLUX_ATOL, LNPROB_RTOL, LNPROB_ATOL

pytestmark = pytest.mark.gpu

IONS = {"HI": [1215.6701, 1025.7223, 972.5368], "CIV": [1548.195, 1550.770], "MgII": [2796.352, 2803.531],
        "FeII": [2600.1729, 2586.650, 2382.765, 2344.214], "SiII": [1260.4221, 1193.2897, 1190.4158, 1526.7066],
        "OVI": [1031.927, 1037.616], "AlIII": [1854.7164, 1862.7895]}


def _random_case(seed):
    from rbvfit_amd.model import FitConfiguration, VoigtModel
    from rbvfit_amd.lsf import gaussian_taps
    rng = np.random.default_rng(seed)
    cfg = FitConfiguration()
    n_sys = int(rng.integers(1, 4))
    centres = []
    for s in range(n_sys):
        z = round(float(rng.uniform(0.0, 1.2)), 4) + 0.0001 * s
        for ion in rng.choice(list(IONS), size=int(rng.integers(1, 3)), replace=False):
            trans = list(rng.choice(IONS[ion], size=int(rng.integers(1, len(IONS[ion]) + 1)), replace=False))
            cfg.add_system(z, str(ion), trans, int(rng.integers(1, 6)))
            centres += [t * (1 + z) for t in trans]
    C = cfg.total_components
    kind = seed % 4
    fwhm = [None, "2.2", "6.5", "13.0"][kind]
    taps = None
    if seed % 7 == 3:                                   # tabulated asymmetric kernel (astropy 'extend' branch)
        j = np.arange(-15, 16)
        taps = np.exp(-0.5 * (j / 2.7) ** 2) * (1 + 0.01 * j) + 0.002
    method = "fast" if seed % 9 == 4 else "wofz"           # the reference's Tepper-Garcia option (bug-compatible wings)
    model = VoigtModel(cfg, FWHM=fwhm, kernel_taps=taps, voigt_method=method)
    # wavelength grid around a random subset of the lines; sometimes descending or non-uniform
    c0 = float(rng.choice(centres))
    span = float(rng.choice([6.0, 25.0, 120.0]))
    P = int(rng.integers(60, 2800))
    wave = np.linspace(c0 - span * rng.uniform(0.2, 0.8), c0 + span * rng.uniform(0.2, 0.8), P)
    if seed % 5 == 1:
        wave = wave[::-1].copy()
    if seed % 5 == 2:
        wave = np.sort(wave + rng.uniform(-0.3, 0.3, P) * (wave[1] - wave[0]))
    N = rng.uniform(11.5, 15.5, C)
    if seed % 6 == 0:
        N[0] = rng.uniform(17.0, 20.5)                  # saturated / damped component
    b = rng.uniform(2.5, 90.0, C)
    v = rng.uniform(-250.0, 250.0, C)
    theta = np.concatenate([N, b, v])
    lb = np.concatenate([np.full(C, 10.0), np.full(C, 1.0), np.full(C, -400.0)])
    ub = np.concatenate([np.full(C, 21.0), np.full(C, 150.0), np.full(C, 400.0)])
    W = 6
    thetas = np.clip(theta + rng.normal(0, 1, (W, 3 * C)) * np.concatenate([np.full(C, 0.2), np.full(C, 3.0), np.full(C, 15.0)]),
                     lb + 1e-9, ub - 1e-9)
    thetas[0] = theta
    err = rng.uniform(0.02, 0.1, P)
    return model, wave, err, thetas, lb, ub, rng


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("FUZZ_SEEDS", "36"))))
def test_random_configuration(seed):
    import rbvfit_amd
    from oracle import voigt_oracle as vo
    model, wave, err, thetas, lb, ub, rng = _random_case(seed)
    data = model.compile().data
    od = vo.OracleModelData(data.atomic_lambda0, data.atomic_gamma, data.atomic_f, data.z_factors, data.N_indices,
                            data.b_indices, data.v_indices, data.taps if data.taps is not None else np.zeros(0),
                            data.lsf_mode, data.voigt_method)
    clean = vo.model_flux(od, thetas[0], wave)
    flux = clean + rng.normal(0, 1, wave.size) * err
    inst = vo.OracleInstrument.from_error(od, wave, flux, err)
    ref = vo.lnprob_batch(thetas, lb, ub, [inst])
    with rbvfit_amd.Engine(0) as e:
        e.set_bounds(lb, ub)
        e.add_instrument(wave, flux, inst.inv_sigma2, inst.log_inv_sigma2, **data.engine_kwargs())
        got = e.lnprob(thetas)
        for k, v in (("walker", 0), ("geom", 0), ("finalize", 0)):   # the two-pass tile geometry and the separate
            e.set_option(k, v)                                         # final reduction (large batches) as well
        got_big = e.lnprob(thetas)
        e.set_option("walker", 1)                                      # and the one-launch walker kernel where it applies:
        e.set_option("walker_split", 0)                                # a walker as ONE workgroup of two-pass tiles ...
        got_walker = e.lnprob(thetas)
        was_walker = e.last_launch_kind == "walker"
        got_split = {}
        for G in (-1, 2, 4, 8):                                        # ... and as several workgroups of one-pass tiles (small batches)
            e.set_option("walker_split", G)
            r = e.lnprob(thetas)
            if e.last_launch_kind == "walker" and e.last_walker_split > 0:
                got_split[G] = (r, e.last_walker_split)
        e.set_option("walker_split", 0)
        e.set_option("walker", 0); e.set_option("geom", 1); e.set_option("finalize", 0)
        got_small = e.lnprob(thetas)                                   # the one-pass tile launches: the split form's geometry
        for k in ("walker", "geom", "finalize"):
            e.set_option(k, -1)
        fl = e.model_flux(0, thetas[:3])
        un = e.model_flux(0, thetas[:2], convolved=False)
    np.testing.assert_allclose(got, ref, rtol=LNPROB_RTOL, atol=LNPROB_ATOL)
    np.testing.assert_allclose(got_big, ref, rtol=LNPROB_RTOL, atol=LNPROB_ATOL)
    clustered = any(g.components >= 3 for s in model.config.systems for g in s.ion_groups)
    same_tiles = data.taps is None or data.taps.size <= 33     # (longer LSFs: 2-/4-wave tile workgroups in the launches,
                                                               # single-wave tiles in the walker kernel)
    np.testing.assert_allclose(got_small, ref, rtol=LNPROB_RTOL, atol=LNPROB_ATOL)
    for G, (r, used) in got_split.items():
        assert was_walker and (G < 0 or used <= G)
        np.testing.assert_allclose(r, ref, rtol=LNPROB_RTOL, atol=LNPROB_ATOL)
        # the sum over a walker's tiles is taken in tile order by the group that finishes last: independent of the number of groups
        np.testing.assert_array_equal(r, got_split[min(got_split)][0])
        if data.n_lines < 8 and not clustered:
            np.testing.assert_array_equal(r, got_small)        # ... and that of the one-pass tile launches, bit for bit
    if data.n_lines < 8 and not clustered and same_tiles:
        np.testing.assert_array_equal(got_walker, got_big)     # same tiles, same summation order: bit-identical
    else:       # the tile launches use multipole expansions of clusters (>= 3 components of a transition) and, from 8 lines
                # on, per-block far-field expansions; the walker kernel walks every line
        np.testing.assert_allclose(got_walker, got_big, rtol=1e-12, atol=1e-9)
    for i in range(3):
        np.testing.assert_allclose(fl[i], vo.model_flux(od, thetas[i], wave), rtol=0, atol=FLUX_ATOL)
    for i in range(2):
        np.testing.assert_allclose(un[i], vo.model_flux(od, thetas[i], wave, return_unconvolved=True), rtol=0, atol=FLUX_ATOL)


def test_more_than_64_and_128_lines():
    """Line-core flags are kept in 64-line mask words: exercise 2 and 3 words, with clusters that
    straddle the word boundaries."""
    import rbvfit_amd
    from oracle import voigt_oracle as vo
    from rbvfit_amd.model import FitConfiguration, VoigtModel
    rng = np.random.default_rng(99)
    for n_sys, comps in ((2, 9), (3, 11)):            # 2*4*9 = 72 lines, 3*4*11 = 132 lines
        cfg = FitConfiguration()
        for s in range(n_sys):
            cfg.add_system(0.30 + 0.004 * s, "FeII", [2600.1729, 2586.650, 2382.765, 2344.214], comps)
        model = VoigtModel(cfg, FWHM="4.0")
        data = model.compile().data
        assert data.n_lines == n_sys * 4 * comps
        C = cfg.total_components
        wave = np.linspace(3030.0, 3420.0, 7000)
        theta = np.concatenate([rng.uniform(12.5, 14.5, C), rng.uniform(5, 40, C), rng.uniform(-200, 200, C)])
        lb = np.concatenate([np.full(C, 10.0), np.full(C, 1.0), np.full(C, -400.0)])
        ub = np.concatenate([np.full(C, 18.0), np.full(C, 150.0), np.full(C, 400.0)])
        thetas = np.clip(theta + 0.05 * rng.standard_normal((4, 3 * C)), lb + 1e-9, ub - 1e-9)
        od = vo.OracleModelData(data.atomic_lambda0, data.atomic_gamma, data.atomic_f, data.z_factors, data.N_indices,
                                data.b_indices, data.v_indices, data.taps, data.lsf_mode, data.voigt_method)
        err = np.full(wave.size, 0.05)
        flux = vo.model_flux(od, thetas[0], wave) + rng.normal(0, 0.05, wave.size)
        inst = vo.OracleInstrument.from_error(od, wave, flux, err)
        with rbvfit_amd.Engine(0) as e:
            e.set_bounds(lb, ub)
            e.add_instrument(wave, flux, inst.inv_sigma2, inst.log_inv_sigma2, **data.engine_kwargs())
            got = e.lnprob(thetas)
            fl = e.model_flux(0, thetas[:2])
        np.testing.assert_allclose(got, vo.lnprob_batch(thetas, lb, ub, [inst]), rtol=LNPROB_RTOL, atol=LNPROB_ATOL)
        for i in range(2):
            np.testing.assert_allclose(fl[i], vo.model_flux(od, thetas[i], wave), rtol=0, atol=FLUX_ATOL)


def test_instruments_sharing_or_not_sharing_their_line_tables():
    """Three instruments on one parameter vector: B has A's line tables (its record-preparation launch is skipped, A's
    records are reused), C differs from B in the Voigt method only (records must be made again).  Every order of
    adding them agrees with the oracle."""
    import itertools
    import rbvfit_amd
    from oracle import voigt_oracle as vo
    from rbvfit_amd.model import FitConfiguration, VoigtModel
    rng = np.random.default_rng(5)
    cfg = FitConfiguration()
    cfg.add_system(0.348, "MgII", [2796.352, 2803.531], 3)
    cfg.add_system(0.349, "FeII", [2600.1729, 2586.650], 2)
    C = cfg.total_components
    theta = np.concatenate([rng.uniform(12.5, 14.0, C), rng.uniform(6, 30, C), rng.uniform(-100, 100, C)])
    lb = np.concatenate([np.full(C, 10.0), np.full(C, 1.0), np.full(C, -400.0)])
    ub = np.concatenate([np.full(C, 18.0), np.full(C, 150.0), np.full(C, 400.0)])
    thetas = np.clip(theta + 0.05 * rng.standard_normal((5, 3 * C)), lb + 1e-9, ub - 1e-9)
    specs = [("wofz", "6.5", np.linspace(3755.0, 3795.0, 1500)),
             ("wofz", "3.0", np.linspace(3480.0, 3515.0, 1100)),
             ("fast", "3.0", np.linspace(3765.0, 3790.0, 900))]
    built = []
    for method, fwhm, wave in specs:
        data = VoigtModel(cfg, FWHM=fwhm, voigt_method=method).compile().data
        od = vo.OracleModelData(data.atomic_lambda0, data.atomic_gamma, data.atomic_f, data.z_factors, data.N_indices,
                                data.b_indices, data.v_indices, data.taps, data.lsf_mode, data.voigt_method)
        err = np.full(wave.size, 0.04)
        flux = vo.model_flux(od, thetas[0], wave) + rng.normal(0, 0.04, wave.size)
        built.append((data, vo.OracleInstrument.from_error(od, wave, flux, err), wave, flux))
    for order in itertools.permutations(range(3)):
        insts = [built[k][1] for k in order]
        ref = vo.lnprob_batch(thetas, lb, ub, insts)
        with rbvfit_amd.Engine(0) as e:
            e.set_bounds(lb, ub)
            for k in order:
                data, oi, wave, flux = built[k]
                e.add_instrument(wave, flux, oi.inv_sigma2, oi.log_inv_sigma2, **data.engine_kwargs())
            got = e.lnprob(thetas)
            e.set_option("geom", 0); e.set_option("finalize", 0)
            got_big = e.lnprob(thetas)
        np.testing.assert_allclose(got, ref, rtol=LNPROB_RTOL, atol=LNPROB_ATOL, err_msg=str(order))
        np.testing.assert_allclose(got_big, ref, rtol=LNPROB_RTOL, atol=LNPROB_ATOL, err_msg=str(order))


@pytest.mark.parametrize("seed", [3, 11, 29])
def test_farfield_expansions_agree_with_direct_evaluation(seed):
    """From 8 lines on, the tile launches take the lines that are far from a block of 192 pixels from ONE polynomial per
    (walker, block) instead of walking them (farfield_kernel).  Same spectrum with the expansions on and off, and the
    oracle: many lines, a damped component (must stay out of the expansions until it is far enough), wide and narrow
    grids, ascending and descending."""
    import rbvfit_amd
    from oracle import voigt_oracle as vo
    from rbvfit_amd.model import FitConfiguration, VoigtModel
    rng = np.random.default_rng(seed)
    cfg = FitConfiguration()
    cfg.add_system(0.348, "MgII", [2796.352, 2803.531], 3)
    cfg.add_system(0.3485, "FeII", [2600.1729, 2586.650, 2382.765], 3)
    cfg.add_system(0.90, "CIV", [1548.195, 1550.770], 2)
    cfg.add_system(2.05, "HI", [1215.6701], 1)
    C = cfg.total_components
    model = VoigtModel(cfg, FWHM="6.5")
    data = model.compile().data
    assert data.n_lines >= 8
    lo, hi, P = [(2900.0, 3800.0, 9000), (3650.0, 3800.0, 6000), (3000.0, 3790.0, 3000)][seed % 3]
    wave = np.linspace(lo, hi, P)
    if seed % 2:
        wave = wave[::-1].copy()
    N = rng.uniform(12.5, 14.5, C); N[-1] = rng.uniform(19.0, 20.3)        # the HI component is damped
    theta = np.concatenate([N, rng.uniform(6, 40, C), rng.uniform(-150, 150, C)])
    lb = np.concatenate([np.full(C, 10.0), np.full(C, 1.0), np.full(C, -400.0)])
    ub = np.concatenate([np.full(C, 21.0), np.full(C, 150.0), np.full(C, 400.0)])
    thetas = np.clip(theta + rng.standard_normal((5, 3 * C)) * np.concatenate([np.full(C, 0.1), np.full(C, 2.0), np.full(C, 10.0)]),
                     lb + 1e-9, ub - 1e-9)
    od = vo.OracleModelData(data.atomic_lambda0, data.atomic_gamma, data.atomic_f, data.z_factors, data.N_indices,
                            data.b_indices, data.v_indices, data.taps, data.lsf_mode, data.voigt_method)
    err = np.full(P, 0.03)
    flux = vo.model_flux(od, thetas[0], wave) + rng.normal(0, 0.03, P)
    inst = vo.OracleInstrument.from_error(od, wave, flux, err)
    ref = vo.lnprob_batch(thetas, lb, ub, [inst])
    got = {}
    for ff in (0, 1):
        with rbvfit_amd.Engine(0) as e:
            e.set_option("farfield", ff)                      # (read when the instrument is added)
            e.set_bounds(lb, ub)
            e.add_instrument(wave, flux, inst.inv_sigma2, inst.log_inv_sigma2, **data.engine_kwargs())
            e.set_option("walker", 0)
            a = e.lnprob(thetas)
            assert e.last_launch_kind == ("tiles+farfield" if ff else "tiles")
            e.set_option("geom", 0); e.set_option("finalize", 0)
            b = e.lnprob(thetas)
            got[ff] = (a, b)
    for k in (0, 1):
        np.testing.assert_allclose(got[1][k], got[0][k], rtol=1e-12, atol=1e-9)
        np.testing.assert_allclose(got[1][k], ref, rtol=LNPROB_RTOL, atol=LNPROB_ATOL)


def _narrow_pixel_case(seed):
    """C4-like geometry at fuzz size: pixels of 0.1-0.3 km/s (a 192-pixel block is a few Doppler widths wide, so the
    instrument gets farfield_kernel<9, true>: members of near clusters line by line, 9 terms from |x| >= 14), >= 8 lines in
    clusters of 3-6 components of the MgII doublet at two or three nearby redshifts, sometimes a damped component and
    lines outside clusters, theta spreads of the size test_random_configuration uses."""
    from rbvfit_amd.model import FitConfiguration, VoigtModel
    rng = np.random.default_rng(4000 + seed)
    cfg = FitConfiguration()
    n_sys = 2 + seed % 2
    for s in range(n_sys):
        cfg.add_system(round(0.348 + 0.0012 * s * float(rng.uniform(0.6, 1.4)), 5), "MgII", [2796.352, 2803.531],
                       int(rng.integers(3, 7)))
    if seed % 4 == 1:                                    # two FeII lines inside the window: lines outside clusters
        cfg.add_system(0.4515, "FeII", [2600.1729, 2586.650], 2)
    C = cfg.total_components
    fwhm = [None, "2.2", "6.5"][seed % 3]
    model = VoigtModel(cfg, FWHM=fwhm)
    pix_kms = float(rng.choice([0.1, 0.2, 0.3]))
    P = int(rng.integers(5000, 12000))
    centre = float(rng.uniform(3768.0, 3781.0))
    dlam = centre * pix_kms / 299792.458
    wave = centre + dlam * (np.arange(P) - P * rng.uniform(0.3, 0.7))
    if seed % 5 == 2:
        wave = wave[::-1].copy()
    N = rng.uniform(12.3, 14.6, C)
    if seed % 3 == 0:
        N[int(rng.integers(0, C))] = rng.uniform(18.3, 20.2)          # one damped component
    b = rng.uniform(3.0, 45.0, C)
    v = rng.uniform(-180.0, 180.0, C)
    theta = np.concatenate([N, b, v])
    lb = np.concatenate([np.full(C, 10.0), np.full(C, 1.0), np.full(C, -400.0)])
    ub = np.concatenate([np.full(C, 21.0), np.full(C, 150.0), np.full(C, 400.0)])
    W = 8
    thetas = np.clip(theta + rng.normal(0, 1, (W, 3 * C)) * np.concatenate([np.full(C, 0.2), np.full(C, 3.0), np.full(C, 15.0)]),
                     lb + 1e-9, ub - 1e-9)
    thetas[0] = theta
    err = rng.uniform(0.02, 0.1, P)
    return model, wave, err, thetas, lb, ub, rng


@pytest.mark.parametrize("seed", range(10))
def test_narrow_pixel_farfield_members_wide_theta(seed):
    """farfield_kernel<9, true>: forced on, wide theta, asserted to be the instance that ran and to have
    taken cluster members one by one; lnprob AND the convolved model flux through the expansions against the oracle, and
    against the launches without expansions / with the plain <6, false> instance."""
    import rbvfit_amd
    from oracle import voigt_oracle as vo
    model, wave, err, thetas, lb, ub, rng = _narrow_pixel_case(seed)
    data = model.compile().data
    assert data.n_lines >= 8
    od = vo.OracleModelData(data.atomic_lambda0, data.atomic_gamma, data.atomic_f, data.z_factors, data.N_indices,
                            data.b_indices, data.v_indices, data.taps if data.taps is not None else np.zeros(0),
                            data.lsf_mode, data.voigt_method)
    flux = vo.model_flux(od, thetas[0], wave) + rng.normal(0, 1, wave.size) * err
    inst = vo.OracleInstrument.from_error(od, wave, flux, err)
    ref = vo.lnprob_batch(thetas, lb, ub, [inst])
    ref_flux = [vo.model_flux(od, thetas[i], wave) for i in range(4)]
    res = {}
    for name, opts in (("members", {"farfield": 1}), ("plain", {"farfield": 1, "no_ff_members": 1}), ("off", {"farfield": 0})):
        with rbvfit_amd.Engine(0) as e:
            for k, val in opts.items():
                e.set_option(k, val)                          # (read when the instrument is added)
            e.set_bounds(lb, ub)
            e.add_instrument(wave, flux, inst.inv_sigma2, inst.log_inv_sigma2, **data.engine_kwargs())
            e.set_option("walker", 0)
            lp = e.lnprob(thetas)
            kind, info = e.last_launch_kind, e.last_farfield_info
            e.set_option("geom", 0); e.set_option("finalize", 0)
            lp_big = e.lnprob(thetas)
            info_big = e.last_farfield_info
            e.set_option("flux_farfield", 1 if opts["farfield"] else 0)
            fl = e.model_flux(0, thetas[:4])
            info_flux = e.last_farfield_info
            res[name] = (lp, lp_big, fl, kind, info, info_big, info_flux)
    lp, lp_big, fl, kind, info, info_big, info_flux = res["members"]
    assert kind == "tiles+farfield"
    for i_ in (info, info_big, info_flux):
        assert i_["variant"] == "members" and i_["covered"] > 0
    assert res["plain"][3] == "tiles+farfield" and res["plain"][4]["variant"] == "lines+clusters"
    assert res["off"][3] == "tiles" and res["off"][4]["variant"] == "none" and res["off"][6]["variant"] == "none"
    # the members instance took lines the plain one had to leave to the tile kernel: cluster members, one by one
    assert info_big["covered_members"] > res["plain"][5]["covered_members"]
    for name in ("members", "plain", "off"):
        a, b_, f = res[name][:3]
        np.testing.assert_allclose(a, ref, rtol=LNPROB_RTOL, atol=LNPROB_ATOL, err_msg=name)
        np.testing.assert_allclose(b_, ref, rtol=LNPROB_RTOL, atol=LNPROB_ATOL, err_msg=name)
        for i in range(4):
            np.testing.assert_allclose(f[i], ref_flux[i], rtol=0, atol=FLUX_ATOL, err_msg=name)
    for k in (0, 1):
        np.testing.assert_allclose(res["members"][k], res["off"][k], rtol=1e-12, atol=1e-9)


@pytest.mark.parametrize("n_inst", [2, 3, 4])
def test_several_instruments_in_one_walker_launch(n_inst):
    """Up to four instruments with the same line tables and at most 16 tiles together run as ONE walker_kernel launch
    (the waves of each further instrument behind those of the one before, one set of records): against the oracle and, tile
    for tile the same arithmetic, bit-identical to the preparation + tile + finalize launches."""
    import rbvfit_amd
    from oracle import voigt_oracle as vo
    from rbvfit_amd.model import FitConfiguration, VoigtModel
    rng = np.random.default_rng(17)
    cfg = FitConfiguration()
    cfg.add_system(0.348, "MgII", [2796.352, 2803.531], 2)
    C = cfg.total_components
    theta = np.array([13.5, 13.2, 15.0, 25.0, -40.0, 20.0])
    lb = np.concatenate([np.full(C, 10.0), np.full(C, 2.0), np.full(C, -300.0)])
    ub = np.concatenate([np.full(C, 17.0), np.full(C, 100.0), np.full(C, 300.0)])
    thetas = np.clip(theta + 1e-2 * rng.standard_normal((40, 3 * C)), lb + 1e-9, ub - 1e-9)
    thetas[5, 0] = 25.0                                        # one walker outside the box
    insts, engine_args = [], []
    specs = [("6.5", np.linspace(3755.0, 3795.0, 2500)), ("3.0", np.linspace(3760.0, 3790.0, 1400)),
             ("4.0", np.linspace(3764.0, 3774.0, 700)[::-1].copy()), (None, np.linspace(3775.0, 3785.0, 500))][:n_inst]
    for fwhm, wave in specs:
        data = VoigtModel(cfg, FWHM=fwhm).compile().data
        od = vo.OracleModelData(data.atomic_lambda0, data.atomic_gamma, data.atomic_f, data.z_factors, data.N_indices,
                                data.b_indices, data.v_indices, data.taps, data.lsf_mode, data.voigt_method)
        err = np.full(wave.size, 0.04)
        flux = vo.model_flux(od, theta, wave) + rng.normal(0, 0.04, wave.size)
        oi = vo.OracleInstrument.from_error(od, wave, flux, err)
        insts.append(oi); engine_args.append((wave, flux, oi, data))
    ref = vo.lnprob_batch(thetas, lb, ub, insts)
    with rbvfit_amd.Engine(0) as e:
        e.set_bounds(lb, ub)
        for wave, flux, oi, data in engine_args:
            e.add_instrument(wave, flux, oi.inv_sigma2, oi.log_inv_sigma2, **data.engine_kwargs())
        got = e.lnprob(thetas)
        assert e.last_launch_kind == "walker"
        # the same launch left waiting on the GPU for the next call's theta (pre-armed: walker_kernel2 / walker_kernel4 too)
        e.set_option("prearm", 1)
        for _ in range(6):
            np.testing.assert_array_equal(e.lnprob(thetas), got)
        assert e.prearm_counts["used"] >= 5 and e.last_launch_kind == "walker"
        e.set_option("prearm", -1)
        e.set_option("walker", 0); e.set_option("geom", 0); e.set_option("finalize", 0); e.set_option("tile_multi", 0)
        launches = e.lnprob(thetas)
        assert e.last_launch_kind == "tiles"
        # ... and with the tiles of all the instruments in one launch (tile_kernel_multi), final reduction by launch and by ticket
        e.set_option("tile_multi", 1)
        one = e.lnprob(thetas)
        assert e.last_launch_kind == "tiles-multi"
        np.testing.assert_array_equal(one, launches)
        e.set_option("finalize", 1)
        np.testing.assert_array_equal(e.lnprob(thetas), launches)
        assert e.last_launch_kind == "tiles-multi"
        e.set_option("finalize", 0); e.set_option("tile_multi", -1)
        e.set_option("walker", -1)
        pos, lp, chain, clp, nacc = e.stretch_run(thetas[:32], 12, seed=4)        # half-steps as one launch each
        e.set_option("geom", 0)
        np.testing.assert_array_equal(clp[-1], e.lnprob(chain[-1]))
    assert got[5] == -np.inf and launches[5] == -np.inf
    np.testing.assert_array_equal(got, launches)
    np.testing.assert_allclose(got, ref, rtol=LNPROB_RTOL, atol=LNPROB_ATOL)


@pytest.mark.parametrize("kind", ["gauss45", "tab101"])
def test_long_lsf_in_the_walker_kernel(kind):
    """LSFs longer than 33 taps make the tile LAUNCHES use 2- or 4-wave workgroups; the walker kernel keeps single-wave
    tiles of 384 evaluated pixels (more halo, one launch).  Different tiles, so agreement with the launches is to rounding;
    both against the oracle."""
    import rbvfit_amd
    from oracle import voigt_oracle as vo
    from rbvfit_amd.model import FitConfiguration, VoigtModel
    from rbvfit_amd.workloads import cos_like_kernel
    rng = np.random.default_rng(23)
    cfg = FitConfiguration()
    cfg.add_system(0.348, "MgII", [2796.352, 2803.531], 2)
    model = VoigtModel(cfg, FWHM="13.0") if kind == "gauss45" else VoigtModel(cfg, FWHM="6.5", kernel_taps=cos_like_kernel())
    data = model.compile().data
    assert (data.taps.size > 33)
    theta = np.array([13.5, 13.2, 15.0, 25.0, -40.0, 20.0])
    lb = np.concatenate([np.full(2, 10.0), np.full(2, 2.0), np.full(2, -300.0)])
    ub = np.concatenate([np.full(2, 17.0), np.full(2, 100.0), np.full(2, 300.0)])
    thetas = np.clip(theta + 1e-2 * rng.standard_normal((24, 6)), lb + 1e-9, ub - 1e-9)
    wave = np.linspace(3760.0, 3790.0, 2100)
    od = vo.OracleModelData(data.atomic_lambda0, data.atomic_gamma, data.atomic_f, data.z_factors, data.N_indices,
                            data.b_indices, data.v_indices, data.taps, data.lsf_mode, data.voigt_method)
    err = np.full(wave.size, 0.04)
    flux = vo.model_flux(od, theta, wave) + rng.normal(0, 0.04, wave.size)
    oi = vo.OracleInstrument.from_error(od, wave, flux, err)
    ref = vo.lnprob_batch(thetas, lb, ub, [oi])
    with rbvfit_amd.Engine(0) as e:
        e.set_bounds(lb, ub)
        e.add_instrument(wave, flux, oi.inv_sigma2, oi.log_inv_sigma2, **data.engine_kwargs())
        e.set_option("walker", 1)                     # (by default only from 192 walkers on for these LSFs)
        got = e.lnprob(thetas)
        assert e.last_launch_kind == "walker"
        e.set_option("walker", 0)
        launches = e.lnprob(thetas)
        assert e.last_launch_kind == "tiles"
        e.set_option("walker", -1)
        big = np.tile(thetas, (10, 1))
        got_big = e.lnprob(big)
        assert e.last_launch_kind == "walker"
    np.testing.assert_allclose(got, ref, rtol=LNPROB_RTOL, atol=LNPROB_ATOL)
    np.testing.assert_allclose(launches, ref, rtol=LNPROB_RTOL, atol=LNPROB_ATOL)
    np.testing.assert_allclose(got, launches, rtol=1e-13, atol=0)
    np.testing.assert_array_equal(got_big, np.tile(got, 10))
