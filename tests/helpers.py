"""Shared test plumbing: build a rbvfit_amd.Engine from a golden fixture."""
import numpy as np


def fixture_instruments(z):
    return [str(s) for s in z["instruments"]]


def engine_from_fixture(z, device_id=0):
    import rbvfit_amd
    eng = rbvfit_amd.Engine(device_id)
    eng.set_bounds(z["lb"], z["ub"])
    for inst in fixture_instruments(z):
        g = lambda k: z[f"{inst}__{k}"]
        eng.add_instrument(g("wave"), g("flux"), g("inv_sigma2"), g("log_inv_sigma2"),
                           g("lambda0"), g("gamma"), g("f"), g("zfac"), g("N_idx"), g("b_idx"), g("v_idx"),
                           taps=g("taps"), lsf_mode=int(g("lsf_mode")), voigt_method=int(g("voigt_method")))
    return eng
