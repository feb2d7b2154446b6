"""GPU parity tests proper: the HIP path, called through the C ABI, against the committed golden
vectors (generated from the real reference) and against the CPU oracle on the same inputs."""
import numpy as np
import pytest

from conftest import golden_cases, load_golden, FLUX_ATOL, LNPROB_RTOL, LNPROB_ATOL, H_RTOL
from helpers import engine_from_fixture, fixture_instruments

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def oracle():
    from oracle import voigt_oracle as vo
    return vo


# launch structure: "auto" (by batch size), "walker" (one launch, workgroup = walker, wherever it applies), or
# prep + tile (+ finalize) launches with tile geometry two-pass "0" / one-pass "1" x final reduction own launch
# "f0" / ticket "f1"
@pytest.mark.parametrize("geom", ["auto", "walker", "0-f0", "0-f1", "1-f0", "1-f1"])
@pytest.mark.parametrize("name", golden_cases())
def test_lnprob_matches_golden(name, geom):
    z = load_golden(name)
    with engine_from_fixture(z) as eng:
        if geom == "walker":
            eng.set_option("walker", 1)
        elif geom != "auto":
            eng.set_option("walker", 0)
            eng.set_option("geom", int(geom[0]))
            eng.set_option("finalize", int(geom[-1]))
        got = eng.lnprob(z["thetas"])
    ref = z["lnprob"]
    assert np.array_equal(np.isneginf(got), np.isneginf(ref))
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    fin = np.isfinite(ref)
    np.testing.assert_allclose(got[fin], ref[fin], rtol=LNPROB_RTOL, atol=LNPROB_ATOL)


@pytest.mark.parametrize("name", golden_cases())
def test_model_flux_matches_golden(name):
    z = load_golden(name)
    with engine_from_fixture(z) as eng:
        for k, inst in enumerate(fixture_instruments(z)):
            ref = z[f"{inst}__model_flux"]
            got = eng.model_flux(k, z["thetas"][:ref.shape[0]])
            np.testing.assert_allclose(got, ref, rtol=0, atol=FLUX_ATOL)


def test_voigt_h_matches_scipy_grid():
    """The tiered device Faddeeva vs scipy.special.wofz (golden grid, scipy 1.7.1)."""
    import rbvfit_amd
    z = load_golden("hgrid")
    a, x, H = z["a"], z["x"], z["H"]
    order = np.argsort(x)          # tiers are chosen per 64 consecutive x: exercise sorted ...
    with rbvfit_amd.Engine(0) as eng:
        got_sorted = eng.voigt_h(a, x[order])
        got_raw = eng.voigt_h(a, x)        # ... and unsorted (mixed tiers inside a wave)
    for got, ref in ((got_sorted, H[:, order]), (got_raw, H)):
        ok = a > 0                          # a == 0: wings are exp(-x^2) only; checked absolutely below
        np.testing.assert_allclose(got[ok], ref[ok], rtol=H_RTOL, atol=1e-300)
        np.testing.assert_allclose(got[~ok], ref[~ok], rtol=H_RTOL, atol=1e-17)


def test_batch_equals_singles_and_permutation():
    z = load_golden("c0_mgii")
    th = z["thetas"]
    with engine_from_fixture(z) as eng:
        full = eng.lnprob(th)
        singles = np.array([eng.lnprob(t)[0] for t in th])
        perm = np.random.default_rng(0).permutation(len(th))
        permuted = eng.lnprob(th[perm])
    assert np.array_equal(full, singles, equal_nan=True)          # deterministic reductions
    assert np.array_equal(full[perm], permuted, equal_nan=True)


def test_unconvolved_flux_matches_oracle(oracle):
    z = load_golden("c0_mgii")
    data = oracle.data_from_fixture(z, "G")
    with engine_from_fixture(z) as eng:
        got = eng.model_flux(0, z["thetas"][:3], convolved=False)
    for i in range(3):
        ref = oracle.model_flux(data, z["thetas"][i], z["G__wave"], return_unconvolved=True)
        np.testing.assert_allclose(got[i], ref, rtol=0, atol=FLUX_ATOL)


def test_repeated_calls_with_changing_theta_are_fresh():
    """The fused final reduction hands partial sums between workgroups (possibly on different
    XCDs) through agent-scope atomics; a stale read would surface as a value of the previous call."""
    z = load_golden("c3_mini")                 # two instruments -> tickets span two launches
    rng = np.random.default_rng(7)
    with engine_from_fixture(z) as eng:
        base = z["thetas"][:12]
        ref_first = eng.lnprob(base)
        for it in range(25):
            th = np.clip(z["theta_true"] + 0.05 * rng.standard_normal((64, z["theta_true"].size)) *
                         np.concatenate([np.full(8, 1.0), np.full(8, 5.0), np.full(8, 10.0)]),
                         z["lb"] + 1e-9, z["ub"] - 1e-9)
            got = eng.lnprob(th)
            k = rng.integers(0, 64, 4)
            singles = np.array([eng.lnprob(th[i])[0] for i in k])
            assert np.array_equal(got[k], singles)
        assert np.array_equal(eng.lnprob(base), ref_first)


def test_context_usable_from_other_threads():
    """SURVEY 8(b) threading: the GUI calls runmcmc from a QThread (gui/fitting_tab.py:51-77); a
    context created on one thread must serve calls from others, one at a time (internal mutex)."""
    import threading
    z = load_golden("c0_mgii")
    eng = engine_from_fixture(z)
    try:
        ref = eng.lnprob(z["thetas"])
        results, errors = {}, []

        def work(k):
            try:
                for _ in range(20):
                    results[k] = eng.lnprob(z["thetas"][k::4])
            except Exception as e:                              # pragma: no cover
                errors.append(e)

        ts = [threading.Thread(target=work, args=(k,)) for k in range(4)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert not errors
        for k in range(4):
            assert np.array_equal(results[k], ref[k::4], equal_nan=True)
    finally:
        eng.close()


def test_four_threads_mixing_every_entry_point():
    """One context hammered by four threads that mix the host entries: lnprob (zero-copy and copy paths),
    model_flux (stages theta in the context's d_theta and the flux in its scratch buffer), per-line
    components, the Faddeeva test hook (re-allocates the scratch) and the device sampler (re-allocates it
    again).  Every entry holds the context mutex from argument check to copy-back, so each result must be
    exactly what the same call returns on a quiet context."""
    import threading
    z = load_golden("c0_mgii")
    eng = engine_from_fixture(z)
    try:
        th = z["thetas"]
        big = np.tile(th, (1 + (1 << 20) // (th.size * 8), 1))          # > 1 MiB of theta: the H2D-copy path of lnprob
        ref = dict(lnp=eng.lnprob(th), lnp_big=eng.lnprob(big), flux=eng.model_flux(0, th[:6]),
                   raw=eng.model_flux(0, th[6:9], convolved=False), comp=eng.model_flux_components(0, th[:2], 4),
                   h=eng.voigt_h(np.array([1e-4, 3e-3]), np.linspace(-40, 40, 4001)),
                   mc=eng.stretch_run(th[:16], 5, seed=3))
        errors = []

        def work(k):
            try:
                for it in range(6):
                    order = [(k + it + j) % 7 for j in range(7)]
                    for what in order:
                        if what == 0:
                            assert np.array_equal(eng.lnprob(th), ref["lnp"], equal_nan=True)
                        elif what == 1:
                            assert np.array_equal(eng.model_flux(0, th[:6]), ref["flux"])
                        elif what == 2:
                            assert np.array_equal(eng.model_flux(0, th[6:9], convolved=False), ref["raw"])
                        elif what == 3:
                            assert np.array_equal(eng.model_flux_components(0, th[:2], 4), ref["comp"])
                        elif what == 4:
                            assert np.array_equal(eng.voigt_h(np.array([1e-4, 3e-3]), np.linspace(-40, 40, 4001)), ref["h"])
                        elif what == 5:
                            got = eng.stretch_run(th[:16], 5, seed=3)
                            assert all(np.array_equal(g, r) for g, r in zip(got, ref["mc"]))
                        else:
                            assert np.array_equal(eng.lnprob(big), ref["lnp_big"], equal_nan=True)
            except BaseException as e:                          # pragma: no cover
                errors.append(repr(e))

        ts = [threading.Thread(target=work, args=(k,)) for k in range(4)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert not errors, errors[:3]
    finally:
        eng.close()


def test_fork_guard_raises_in_child():
    """Not fork-safe: a child created by fork() must get an error, not a GPU fault."""
    import os
    z = load_golden("one_px")
    eng = engine_from_fixture(z)
    try:
        eng._pid = os.getpid() + 1                              # what a forked child would see
        with pytest.raises(RuntimeError, match="fork"):
            eng.lnprob(z["thetas"])
    finally:
        eng._pid = os.getpid()
        eng.close()


def test_lnprob_torch_orders_with_default_and_side_streams():
    """Engine.lnprob_torch: theta produced by torch ops, result consumed by torch ops, on the default
    stream (handle 0 = "context stream" to the C ABI: fenced through vp_ctx_stream) and on a side stream."""
    import torch
    z = load_golden("c0_mgii")
    with engine_from_fixture(z) as eng:
        ref = eng.lnprob(z["thetas"])
        base = torch.from_numpy(z["thetas"]).cuda()
        for use_side in (False, True):
            ctx = torch.cuda.stream(torch.cuda.Stream()) if use_side else torch.cuda.stream(torch.cuda.default_stream())
            with ctx:
                for rep in range(20):
                    junk = torch.randn(2048, 2048, device="cuda") @ torch.randn(2048, 2048, device="cuda")   # keep the stream busy
                    theta = (base + 0.0 * junk[0, 0]).contiguous()          # depends on the work before it
                    out = eng.lnprob_torch(theta)
                    got = (out * 1.0).cpu().numpy()                         # consumed on the same stream
                    assert np.array_equal(got, ref, equal_nan=True)
        with pytest.raises(ValueError):
            eng.lnprob_torch(base.float())
        assert eng.stream_handle != 0


def test_multi_context_sharding_from_one_process():
    """vp_multi_* (one process, several device contexts, no torch/RCCL): the batch is cut into
    contiguous blocks of ceil(W/G) rows, every block is enqueued before any is waited for, and the results land
    in the caller's vector in walker order.  The box has one GPU, so the same device is listed two and three
    times -- the sharding, the ragged cases (W < G, W not divisible) and the error path are what is exercised."""
    import rbvfit_amd
    from rbvfit_amd.dist import shard_bounds
    from helpers import fixture_instruments
    z = load_golden("c0_mgii")
    with engine_from_fixture(z) as one:
        ref = one.lnprob(z["thetas"])
    for ids in ([0, 0], [0, 0, 0]):
        with rbvfit_amd.MultiEngine(ids) as m:
            assert m.n_devices == len(ids)
            m.set_bounds(z["lb"], z["ub"])
            for inst in fixture_instruments(z):
                g = lambda k: z[f"{inst}__{k}"]
                m.add_instrument(g("wave"), g("flux"), g("inv_sigma2"), g("log_inv_sigma2"), g("lambda0"), g("gamma"), g("f"),
                                 g("zfac"), g("N_idx"), g("b_idx"), g("v_idx"), taps=g("taps"), lsf_mode=int(g("lsf_mode")),
                                 voigt_method=int(g("voigt_method")))
            for W in (len(z["thetas"]), 7, 2, 1):
                got = m.lnprob(z["thetas"][:W])
                assert np.array_equal(np.isneginf(got), np.isneginf(ref[:W]))
                fin = np.isfinite(ref[:W])
                np.testing.assert_allclose(got[fin], ref[:W][fin], rtol=1e-13, atol=0)
                # each block is exactly what a single context returns for those rows
                with engine_from_fixture(z) as one:
                    for r in range(len(ids)):
                        lo, hi = shard_bounds(W, len(ids), r)
                        if hi > lo:
                            assert np.array_equal(got[lo:hi], one.lnprob(z["thetas"][lo:hi]), equal_nan=True)
            import ctypes as C
            dp = C.POINTER(C.c_double)
            bad, buf = np.zeros((3, 5)), np.empty(3)
            with pytest.raises(rbvfit_amd.RbvfitAmdError, match="device slot 0.*D=5"):       # wrong D: the failing slot is named
                m._check(m._lib.vp_multi_lnprob_batch(m._m, 3, 5, bad.ctypes.data_as(dp), buf.ctypes.data_as(dp)))
    with pytest.raises(rbvfit_amd.RbvfitAmdError):
        rbvfit_amd.MultiEngine([0, 99])


def test_multi_context_errors_on_a_fresh_object_leave_the_output_alone():
    """Error paths of vp_multi_lnprob_batch before any batch has run (no pinned staging buffer exists yet): lnprob with
    bounds but no instrument, and a wrong D, return the error of the failing slot -- no crash -- and do not touch `out`;
    an instrument that one context refuses is taken back from the contexts that had accepted it."""
    import ctypes as C
    import rbvfit_amd
    from helpers import fixture_instruments
    z = load_golden("c0_mgii")
    dp = C.POINTER(C.c_double)
    with rbvfit_amd.MultiEngine([0, 0]) as m:
        m.set_bounds(z["lb"], z["ub"])
        th = np.ascontiguousarray(z["thetas"][:8])
        out = np.full(8, 123.0)
        with pytest.raises(rbvfit_amd.RbvfitAmdError, match="device slot 0.*no instrument"):
            m._check(m._lib.vp_multi_lnprob_batch(m._m, 8, th.shape[1], th.ctypes.data_as(dp), out.ctypes.data_as(dp)))
        assert np.all(out == 123.0)
        inst = fixture_instruments(z)[0]
        g = lambda k: z[f"{inst}__{k}"]
        args = (g("wave"), g("flux"), g("inv_sigma2"), g("log_inv_sigma2"), g("lambda0"), g("gamma"), g("f"), g("zfac"), g("N_idx"),
                g("b_idx"), g("v_idx"))
        kw = dict(taps=g("taps"), lsf_mode=int(g("lsf_mode")), voigt_method=int(g("voigt_method")))
        with pytest.raises(rbvfit_amd.RbvfitAmdError):                  # an even number of taps: refused by the first context
            m.add_instrument(*args, **dict(kw, taps=g("taps")[:-1]))
        m.add_instrument(*args, **kw)
        bad = np.zeros((3, 5))
        with pytest.raises(rbvfit_amd.RbvfitAmdError, match="device slot 0.*D=5"):
            m._check(m._lib.vp_multi_lnprob_batch(m._m, 3, 5, bad.ctypes.data_as(dp), out.ctypes.data_as(dp)))
        assert np.all(out == 123.0)
        with engine_from_fixture(z) as one:
            np.testing.assert_array_equal(m.lnprob(th), one.lnprob(th))
        # a failure on the SECOND context is undone on the first: the two stay in step
        c1 = C.c_void_p(m._lib.vp_multi_ctx(m._m, 1))
        one_ctx_only = m._lib.vp_add_instrument
        n0 = m._lib.vp_num_instruments(C.c_void_p(m._lib.vp_multi_ctx(m._m, 0)))
        m._lib.vp_set_option(c1, b"lds_pad", 10 ** 9)            # context 1 will refuse the next instrument (LDS budget)
        with pytest.raises(rbvfit_amd.RbvfitAmdError, match="device slot 1"):
            m.add_instrument(*args, **kw)
        m._lib.vp_set_option(c1, b"lds_pad", 0)
        assert m._lib.vp_num_instruments(C.c_void_p(m._lib.vp_multi_ctx(m._m, 0))) == n0
        assert m._lib.vp_num_instruments(c1) == n0
        with engine_from_fixture(z) as one:
            np.testing.assert_array_equal(m.lnprob(th), one.lnprob(th))


@pytest.mark.parametrize("structure", ["walker", "tiles-ticket", "tiles-finalize"])
def test_host_entry_completion_modes_agree(structure):
    """vp_lnprob_batch learns that its batch is done in one of three ways (host_spin 0: hipStreamSynchronize, 1: a completion
    word the stream writes, 2: polling the output rows, which are pre-set to a NaN pattern no arithmetic produces and are each
    written once).  All three return the same bits -- finite rows, -inf rows (out of bounds: written by the launch that
    applies the prior), NaN rows (a NaN parameter) --, in every launch structure, also when the batch is re-submitted many
    times, and an input that CARRIES the pattern (as theta, or as a flux pixel) falls back instead of hanging or lying."""
    import struct
    import rbvfit_amd
    z = load_golden("c0_mgii")
    th = np.tile(z["thetas"], (4, 1))[:100].copy()
    th[3, 0] = z["lb"][0] - 1.0                       # -inf
    th[17, 4] = z["ub"][4] + 5.0
    th[40, 2] = np.nan                                # NaN propagates
    sentinel = struct.unpack("<d", struct.pack("<Q", 0x7FF8A5C3965A3C69))[0]
    res = {}
    for spin in (0, 1, 2):
        with engine_from_fixture(z) as eng:
            eng.set_option("host_spin", spin)
            if structure == "walker":
                eng.set_option("walker", 1)
            else:
                eng.set_option("walker", 0)
                eng.set_option("finalize", 1 if structure == "tiles-ticket" else 0)
            a = eng.lnprob(th)
            for _ in range(50):
                assert np.array_equal(eng.lnprob(th), a, equal_nan=True)
            one = np.array([eng.lnprob(th[i])[0] for i in (0, 3, 40, 99)])
            assert np.array_equal(one, a[[0, 3, 40, 99]], equal_nan=True)
            th2 = th.copy(); th2[40, 2] = sentinel        # a NaN with the pattern's payload: same class of result, by the fallback
            b = eng.lnprob(th2)
            assert np.array_equal(np.isnan(b), np.isnan(a)) and np.array_equal(b[~np.isnan(b)], a[~np.isnan(a)])
            res[spin] = a
    assert np.isneginf(res[2][3]) and np.isneginf(res[2][17]) and np.isnan(res[2][40])
    assert np.array_equal(res[0], res[1], equal_nan=True) and np.array_equal(res[0], res[2], equal_nan=True)
    fin = np.isfinite(res[2])
    ref = np.tile(z["lnprob"], 4)[:100]
    np.testing.assert_allclose(res[2][fin], ref[fin], rtol=LNPROB_RTOL, atol=LNPROB_ATOL)
    # the pattern inside the spectrum (every row NaN): the context notices when the instrument is added and never polls rows
    g = lambda k: z[f"{fixture_instruments(z)[0]}__{k}"]
    flux = g("flux").copy(); flux[7] = sentinel
    with rbvfit_amd.Engine(0) as eng:
        eng.set_bounds(z["lb"], z["ub"])
        eng.add_instrument(g("wave"), flux, g("inv_sigma2"), g("log_inv_sigma2"), g("lambda0"), g("gamma"), g("f"), g("zfac"),
                           g("N_idx"), g("b_idx"), g("v_idx"), taps=g("taps"), lsf_mode=int(g("lsf_mode")), voigt_method=int(g("voigt_method")))
        c = eng.lnprob(th)
    assert np.array_equal(np.isneginf(c), np.isneginf(res[2])) and np.all(np.isnan(c[~np.isneginf(c)]))


def test_prearmed_launches_change_nothing():
    """vp_lnprob_batch may leave the launch for the NEXT batch waiting on the GPU (option "prearm").  Whatever happens to that
    launch -- used by the next call, expired because the caller took too long, sent away by another entry point or by a batch
    of another shape, told to go for a batch with rows outside the box or a NaN -- the call returns the bits the ordinary launch
    returns, and the counters say which of these happened."""
    import time
    z = load_golden("c0_mgii")
    rng = np.random.default_rng(5)
    base = np.tile(z["thetas"], (4, 1))[:100].copy()
    batches = []
    for k in range(6):
        th = base + 1e-3 * rng.standard_normal(base.shape) * (z["ub"] - z["lb"])
        th = np.clip(th, z["lb"], z["ub"])
        batches.append(th)
    batches[2][3, 0] = z["lb"][0] - 1.0               # -inf row
    batches[3][40, 2] = np.nan                        # NaN row
    with engine_from_fixture(z) as ref:
        ref.set_option("prearm", 0)
        ref.set_option("walker", 1)
        want = [ref.lnprob(th) for th in batches]
        want_small = ref.lnprob(batches[0][:32])
        flux = ref.model_flux(0, batches[0][:2])
        assert ref.prearm_counts == dict(used=0, expired=0, cancelled=0)
    assert np.isneginf(want[2][3]) and np.isnan(want[3][40])
    with engine_from_fixture(z) as eng:
        eng.set_option("prearm", 1)
        eng.set_option("walker", 1)
        # back to back: every call after the first starts through the launch the call before left behind
        for rep in range(20):
            for th, w in zip(batches, want):
                assert np.array_equal(eng.lnprob(th), w, equal_nan=True)
        c = eng.prearm_counts
        assert c["used"] >= 100 and c["cancelled"] == 0, c
        # another entry point in between: the waiting launch is sent away, both results are right
        np.testing.assert_array_equal(eng.model_flux(0, batches[0][:2]), flux)
        assert np.array_equal(eng.lnprob(batches[1]), want[1], equal_nan=True)
        assert eng.prearm_counts["cancelled"] == c["cancelled"] + 1
        # a batch of another shape
        assert np.array_equal(eng.lnprob(batches[0][:32]), want_small)
        assert np.array_equal(eng.lnprob(batches[4]), want[4])
        assert eng.prearm_counts["cancelled"] >= c["cancelled"] + 3
        # a caller that takes longer than the launch waits
        eng.set_option("prearm_us", 50)
        assert np.array_equal(eng.lnprob(batches[5]), want[5])
        e0 = eng.prearm_counts["expired"]
        for k in range(5):
            time.sleep(0.005)
            assert np.array_equal(eng.lnprob(batches[k]), want[k], equal_nan=True)
        assert eng.prearm_counts["expired"] >= e0 + 4, eng.prearm_counts
        # ... and one whose go word races the expiry (gaps around the waiting time): right either way
        for k in range(300):
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < (20 + (k % 61)) * 1e-6:
                pass
            assert np.array_equal(eng.lnprob(batches[k % 6]), want[k % 6], equal_nan=True)
    # more workgroups than the GPU holds at once (1024 walkers of 12 waves: the second round starts as the first one's leave)
    big = np.tile(batches[0], (11, 1))[:1024].copy()
    with engine_from_fixture(z) as ref:
        ref.set_option("prearm", 0)
        want_big = ref.lnprob(big)
        kind = ref.last_launch_kind
    with engine_from_fixture(z) as eng:
        eng.set_option("prearm", 1)
        for _ in range(10):
            assert np.array_equal(eng.lnprob(big), want_big)
        assert eng.last_launch_kind == kind
        if kind == "walker":
            assert eng.prearm_counts["used"] >= 8, eng.prearm_counts
    # two contexts on one GPU, called in turn: a launch that waits holds its compute units, so each call sends the OTHER context's
    # waiting launch away first -- nobody sits out the other's waiting time (1 ms by default)
    with engine_from_fixture(z) as e1, engine_from_fixture(z) as e2:
        for e in (e1, e2):
            e.set_option("prearm", 1)
            e.set_option("walker", 1)
        for e in (e1, e2, e1, e2):
            assert np.array_equal(e.lnprob(batches[0]), want[0])
        t0 = time.perf_counter()
        for k in range(100):
            assert np.array_equal(e1.lnprob(batches[k % 6]), want[k % 6], equal_nan=True)
            assert np.array_equal(e2.lnprob(batches[(k + 1) % 6]), want[(k + 1) % 6], equal_nan=True)
        per_call = (time.perf_counter() - t0) / 200
        assert per_call < 300e-6, per_call
        assert e1.prearm_counts["cancelled"] >= 90 and e2.prearm_counts["cancelled"] >= 90
    # by default (prearm = -1) a loop of calls arms, a lone call does not
    with engine_from_fixture(z) as eng:
        eng.set_option("walker", 1)
        eng.lnprob(batches[0])
        assert eng.prearm_counts == dict(used=0, expired=0, cancelled=0)
        time.sleep(0.01)
        eng.lnprob(batches[0])
        assert eng.prearm_counts["used"] == 0
        for _ in range(20):
            assert np.array_equal(eng.lnprob(batches[0]), want[0])
        assert eng.prearm_counts["used"] >= 15
