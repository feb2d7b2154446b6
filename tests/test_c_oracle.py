"""Pin the plain-C restatement (oracle/voigt_oracle.c) to the golden vectors (CPU only)."""
import numpy as np
import pytest

from conftest import golden_cases, load_golden, FLUX_ATOL, LNPROB_RTOL, LNPROB_ATOL
from oracle import voigt_oracle as vo
from oracle import c_oracle


def test_rew_matches_scipy_golden_grid():
    z = load_golden("hgrid")
    A, X = np.meshgrid(z["a"], z["x"], indexing="ij")
    got = c_oracle.rew(X.ravel(), A.ravel()).reshape(A.shape)
    ok = z["H"] > 0
    np.testing.assert_allclose(got[ok], z["H"][ok], rtol=2e-13, atol=0)
    assert np.all(got[~ok] == 0) or np.allclose(got[~ok], z["H"][~ok], atol=1e-300)


@pytest.mark.parametrize("name", golden_cases())
def test_c_oracle_lnprob_and_flux(name):
    z = load_golden(name)
    insts = vo.instruments_from_fixture(z)
    co = c_oracle.COracle(insts, z["lb"], z["ub"])
    got = co.lnprob_batch(z["thetas"], nthreads=2)
    ref = z["lnprob"]
    assert np.array_equal(np.isneginf(got), np.isneginf(ref))
    fin = np.isfinite(ref)
    np.testing.assert_allclose(got[fin], ref[fin], rtol=LNPROB_RTOL, atol=LNPROB_ATOL)
    for k, iname in enumerate([str(s) for s in z["instruments"]]):
        ref_fl = z[f"{iname}__model_flux"]
        for i in range(min(2, len(ref_fl))):
            np.testing.assert_allclose(co.model_flux(k, z["thetas"][i]), ref_fl[i], rtol=0, atol=FLUX_ATOL)
