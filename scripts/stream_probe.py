"""Diagnostic for Engine.lnprob_torch stream ordering (default stream vs side stream)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from conftest import load_golden
from helpers import engine_from_fixture
z = load_golden("c0_mgii")
eng = engine_from_fixture(z)
ref = eng.lnprob(z["thetas"])
base = torch.from_numpy(z["thetas"]).cuda()
for use_side in (False, True, False, True):
    st = torch.cuda.Stream() if use_side else torch.cuda.default_stream()
    bad = 0
    with torch.cuda.stream(st):
        for rep in range(40):
            junk = torch.randn(2048, 2048, device="cuda") @ torch.randn(2048, 2048, device="cuda")
            theta = (base + 0.0 * junk[0, 0]).contiguous()
            out = eng.lnprob_torch(theta)
            got = (out * 1.0).cpu().numpy()
            if not np.array_equal(got, ref, equal_nan=True):
                bad += 1
                if bad <= 3:
                    d = np.flatnonzero(~((got == ref) | (np.isnan(got) & np.isnan(ref))))
                    print("  mismatch rep", rep, "rows", d[:8], "got", got[d[:4]], "ref", ref[d[:4]], "theta finite", bool(torch.isfinite(theta).all()))
    print("side" if use_side else "default", "stream: mismatches", bad, "of 40", flush=True)
