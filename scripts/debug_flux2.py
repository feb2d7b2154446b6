import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import engine_from_fixture
from oracle import voigt_oracle as vo
z = np.load("tests/golden/c4_mini.npz")
eng = engine_from_fixture(z)
th = z["thetas"][1]
got = eng.model_flux(0, th[None, :], convolved=False)[0]
d = vo.data_from_fixture(z, "G"); wave = z["G__wave"]
N = 10 ** th[d.N_indices]; b = th[d.b_indices]; v = th[d.v_indices]
zt = d.z_factors * (1 + v / 299792.458) - 1
wr = wave[None, :] / (1 + zt[:, None])
tau = vo.voigt_tau(d.atomic_lambda0, d.atomic_gamma, d.atomic_f, N, b, wr)
ref = np.exp(-tau.sum(0))
dt = -np.log(got / ref)
px = np.arange(2121, 2249)
A = tau[:, px].T                      # (128, 64)
coef, res, rk, sv = np.linalg.lstsq(A, dt[px], rcond=None)
print("dtau at", [2121, 2150, 2184, 2185, 2220, 2248], dt[[2121, 2150, 2184, 2185, 2220, 2248]])
for l in np.argsort(-np.abs(coef))[:8]:
    print("line", l, "coef", coef[l])
print("residual", np.abs(A @ coef - dt[px]).max())
