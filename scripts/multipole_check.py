"""Worst relative truncation error of the distance-tiered multipole expansion of a cluster of lines (voigt_kernels.h:
prep_cluster / multipole_rb) against scipy.special.wofz, over random clusters of 2..8 members.  Tiers: (member |x|
floor, |y| / max|delta| floor, terms)."""
import numpy as np
from scipy.special import wofz
from math import comb
rng = np.random.default_rng(0)
NW = 14
# wing coefficient table c[m][i]
def dfo2(n):
    v = 1.0
    for k in range(1, n + 1): v *= (2 * k - 1) * 0.5
    return v
WC = np.zeros((NW, NW))
for m in range(NW):
    for i in range(m + 1):
        WC[m][i] = comb(2 * m + 1, 2 * i + 1) * (-1) ** i * dfo2(m - i)
def Km(T, a, M):
    a2 = a * a; pref = T * a / np.sqrt(np.pi)
    out = []
    for m in range(M):
        cm = 0.0
        for i in range(m, -1, -1): cm = cm * a2 + WC[m][i]
        out.append(pref * cm)
    return np.array(out)
def multipole(alpha, delta, K, NQ, M):
    Q = np.zeros(NQ)
    for l in range(len(alpha)):
        for jq in range(NQ):
            jj = jq + 2
            for m in range(M):
                nn = 2 * m + 2
                if nn <= jj:
                    Q[jq] += K[l][m] * alpha[l] ** (-nn) * comb(jj - 1, nn - 1) * (-delta[l]) ** (jj - nn)
    return Q
TIERS = [(3000.0, 1000.0, 5), (500.0, 100.0, 7), (100.0, 30.0, 10), (30.0, 10.0, 15), (30.0, 4.0, 26)]
worst = {t[2]: 0.0 for t in TIERS}
for trial in range(300):
    n = rng.integers(2, 9)
    b = rng.uniform(5, 60, n); v = rng.uniform(-150, 150, n)
    a = rng.uniform(1e-5, 0.02, n) * 20 / b
    T = 10 ** rng.uniform(-1, 3, n)
    bmax = b.max()
    alpha = bmax / b                       # x_l = alpha_l (y + delta_l), y in units of bmax
    centre = v.mean() / bmax               # in y units (approx; device centres in 1/wave units)
    delta = centre - v / bmax              # x_l = (V - v_l)/b_l with V velocity coordinate: y = V/bmax - centre ...
    dmax = np.abs(delta).max()
    K = [Km(T[l], a[l], 6) for l in range(n)]
    Q = multipole(alpha, delta, K, 26, 6)
    for X, invrho, J in TIERS:
        Y = max(X + dmax, invrho * dmax)
        for y in np.concatenate([Y * np.array([1.0, 1.01, 1.5, 3.0]), -Y * np.array([1.0, 1.2])]):
            xl = alpha * (y + delta)
            exact = np.sum(T * wofz(xl + 1j * a).real)
            q = 1.0 / y
            acc = 0.0
            for jq in range(J - 1, -1, -1): acc = acc * q + Q[jq]
            approx = acc * q * q
            worst[J] = max(worst[J], abs(approx / exact - 1))
print(worst)
