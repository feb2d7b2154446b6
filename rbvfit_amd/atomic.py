"""Small built-in atomic line table for the host-side model builder.

Role in the reference: ``rb_setline(lambda_rest, 'closest')`` (rb_setline.py:25-64) looks the
requested wavelength up in ``lines/atom_full.dat`` and returns ``wave`` (float64) and
``fval``/``gamma`` as **float32** (rb_setline.py:42,44 -- parity trap T1).  The engine itself takes
the per-line arrays at the C ABI; this table only exists so that tests, the benchmark and users
without rbvfit installed can describe common UV/optical absorbers.  Values are the published
oscillator strengths / damping constants (Morton 2003 and updates) for a subset of common lines;
``register_line`` adds more.  When rbvfit is installed, ``rbvfit_amd.model.tables_from_rbvfit``
takes the arrays from its compiled model instead and this table is not consulted.
"""
from __future__ import annotations

import numpy as np

# ion, rest wavelength [A], f, gamma [s^-1]
_LINES = [
    ("HI", 1215.6701, 0.416400, 6.265e8), ("HI", 1025.7223, 0.079120, 1.897e8),
    ("HI", 972.5368, 0.029000, 8.126e7), ("HI", 949.7431, 0.013940, 4.203e7),
    ("HI", 937.8035, 0.007799, 1.973e7),
    ("CII", 1334.5323, 0.1278, 2.870e8), ("CII", 1036.3367, 0.1231, 2.290e9),
    ("CIV", 1548.195, 0.190800, 2.654e8), ("CIV", 1550.770, 0.095220, 2.641e8),
    ("NV", 1238.821, 0.157000, 3.411e8), ("NV", 1242.804, 0.078230, 3.378e8),
    ("OI", 1302.1685, 0.048870, 5.750e8), ("OI", 1039.2304, 0.009197, 1.888e8),
    ("OVI", 1031.927, 0.132900, 4.163e8), ("OVI", 1037.616, 0.066090, 4.095e8),
    ("NaI", 5891.5833, 0.6311, 6.064e7), ("NaI", 5897.5581, 0.3180, 6.098e7),
    ("MgI", 2852.9642, 1.830000, 5.000e8),
    ("MgII", 2796.352, 0.6123, 2.612e8), ("MgII", 2803.531, 0.3054, 2.592e8),
    ("AlII", 1670.7874, 1.8330, 1.460e9),
    ("AlIII", 1862.7895, 0.2789, 5.361e8), ("AlIII", 1854.7164, 0.5602, 5.432e8),
    ("SiII", 1808.0126, 0.00218, 6.749e6), ("SiII", 1526.7066, 0.11600, 1.960e9),
    ("SiII", 1304.3702, 0.09400, 1.720e9), ("SiII", 1260.4221, 1.007000, 2.533e9),
    ("SiII", 1193.2897, 0.499100, 3.495e9), ("SiII", 1190.4158, 0.250200, 3.503e9),
    ("SiIII", 1206.500, 1.669000, 2.550e9),
    ("SiIV", 1393.755, 0.5140, 8.825e8), ("SiIV", 1402.770, 0.2553, 8.656e8),
    ("CaII", 3934.777, 0.6346, 1.456e8), ("CaII", 3969.591, 0.3145, 1.414e8),
    ("FeII", 2600.1729, 0.2130, 2.700e8), ("FeII", 2586.650, 0.06840, 2.720e8),
    ("FeII", 2382.765, 0.3006, 3.100e8), ("FeII", 2374.4612, 0.03260, 2.990e8),
    ("FeII", 2344.214, 0.109700, 2.680e8),
]


def register_line(ion: str, wrest: float, fval: float, gamma: float) -> None:
    """Add a transition to the in-process table."""
    _LINES.append((str(ion), float(wrest), float(fval), float(gamma)))


def lookup(lambda_rest: float, method: str = "closest"):
    """Mirror of ``rb_setline(lambda_rest, method)``: returns dict(wave float64, fval float32,
    gamma float32, name).  'closest' = nearest wavelength in the table (rb_setline.py:55-56),
    'Exact' = within 1e-3 A (rb_setline.py:53-54)."""
    waves = np.array([r[1] for r in _LINES], dtype=np.float64)
    if method == "closest":
        i = int(np.abs(lambda_rest - waves).argmin())
    elif method == "Exact":
        hits = np.where(np.abs(lambda_rest - waves) < 1e-3)[0]
        if hits.size == 0:
            raise KeyError(f"no line within 1e-3 A of {lambda_rest}")
        i = int(hits[0])
    else:
        raise ValueError("Specify a valid matching method: 'closest' or 'Exact'")
    ion, w, fv, gm = _LINES[i]
    return {"wave": np.float64(w), "fval": np.float32(fv), "gamma": np.float32(gm),
            "name": f"{ion} {int(w)}"}
