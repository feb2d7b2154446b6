import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import engine_from_fixture
name = sys.argv[1] if len(sys.argv) > 1 else "c0_mgii"
z = np.load(f"tests/golden/{name}.npz")
eng = engine_from_fixture(z)
for conv in (False, True):
    got = eng.model_flux(0, z["thetas"][:2], convolved=conv)
    if conv:
        ref = z["G__model_flux"][:2]
    else:
        from oracle import voigt_oracle as vo
        d = vo.data_from_fixture(z, "G")
        ref = np.array([vo.model_flux(d, t, z["G__wave"], return_unconvolved=True) for t in z["thetas"][:2]])
    diff = np.abs(got - ref)
    print("conv", conv, "max diff", diff.max(), "at", np.unravel_index(diff.argmax(), diff.shape))
    bad = np.where(diff[0] > 1e-10)[0]
    print(" n bad", bad.size, "first/last", bad[:5], bad[-5:])
    if bad.size:
        # group into runs
        runs = np.split(bad, np.where(np.diff(bad) > 1)[0] + 1)
        print(" runs:", [(r[0], r[-1]) for r in runs][:20])
        i = bad[0]
        print(" sample", i, got[0, i], ref[0, i])
