"""Small built-in atomic line table for the host-side model builder.

Role in the reference: ``rb_setline(lambda_rest, 'closest')`` (rb_setline.py:25-64) looks the
requested wavelength up in ``lines/atom_full.dat`` and returns ``wave`` (float64) and
``fval``/``gamma`` as **float32** (rb_setline.py:42,44 -- parity trap T1).  The engine itself takes
the per-line arrays at the C ABI; this table only exists so that tests, the benchmark and users
without rbvfit installed can describe common UV/optical absorbers.  Values are the published
oscillator strengths / damping constants (Morton 2003 and updates) for a subset of 39 common lines;
``load_table(path)`` reads a full list in the format of rbvfit's ``lines/atom_full.dat`` (320
rows there) from a caller-supplied path, ``register_line(s)`` add single rows, and a wavelength that
is not in the built-in subset raises instead of snapping to an unrelated line.  When rbvfit is installed, ``rbvfit_amd.model.tables_from_rbvfit``
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


# 'closest' snaps to the nearest table row whatever the distance (rb_setline.py:55-56).  Over the
# reference's full 320-row list the nearest row IS the intended line; over the built-in subset a
# wavelength that is simply missing would silently become an unrelated transition (ZnII 2026 ->
# AlIII 1862).  So while only the built-in subset is loaded, a 'closest' match farther than this many
# Angstrom raises; after load_table() of a full list the reference's unconditional behaviour applies.
CLOSEST_TOLERANCE_A = 0.5
_full_table_loaded = False


class UnknownLineError(KeyError):
    pass


def register_line(ion: str, wrest: float, fval: float, gamma: float) -> None:
    """Add a transition to the in-process table."""
    _LINES.append((str(ion), float(wrest), float(fval), float(gamma)))
    _cache.clear()


def register_lines(rows) -> int:
    """Bulk form: an iterable of (ion, wrest, fval, gamma).  Rows already present (same ion and
    wavelength to 1e-4 A) replace the built-in values.  Returns the number of rows taken."""
    n = 0
    for ion, w, fv, gm in rows:
        ion, w = str(ion), float(w)
        for k, r in enumerate(_LINES):
            if r[0] == ion and abs(r[1] - w) < 1e-4:
                _LINES[k] = (ion, w, float(fv), float(gm))
                break
        else:
            _LINES.append((ion, w, float(fv), float(gm)))
        n += 1
    _cache.clear()
    return n


def load_table(path: str, fmt: str = "atom", full: bool = True) -> int:
    """Read a line list from a caller-supplied file and merge it into the in-process table.

    ``fmt='atom'``: the four whitespace-separated columns ``ion  wrest  fval  gamma`` of rbvfit's
    ``lines/atom_full.dat`` (what ``read_line_list('atom')`` parses, rb_setline.py:66-98); blank
    lines and lines starting with ``#`` or ``;`` are skipped.  ``fmt='lst'``: the
    ``wrest  ion  number  fval`` files with one header line (``lls.lst``, ``dla.lst``; gamma = 0).
    ``full=True`` declares the list complete for the caller's purpose: 'closest' then snaps without a
    distance limit, as the reference does.  The file is NOT shipped with this package: pass the
    path of an rbvfit checkout/installation (see ``load_rbvfit_table``)."""
    global _full_table_loaded
    rows = []
    with open(path, "r") as f:
        lines = f.read().splitlines()
    if fmt == "lst":
        lines = lines[1:]
    for ln in lines:
        t = ln.split()
        if not t or t[0][0] in "#;":
            continue
        if fmt == "atom":
            if len(t) < 4:
                raise ValueError(f"{path}: expected 'ion wrest fval gamma', got {ln!r}")
            rows.append((t[0], float(t[1]), float(t[2]), float(t[3])))
        elif fmt == "lst":
            if len(t) < 4:
                raise ValueError(f"{path}: expected 'wrest ion number fval', got {ln!r}")
            rows.append((t[1], float(t[0]), float(t[3]), 0.0))
        else:
            raise ValueError("fmt must be 'atom' or 'lst'")
    n = register_lines(rows)
    if full and n:
        _full_table_loaded = True
    return n


def load_rbvfit_table() -> int:
    """Load ``lines/atom_full.dat`` from an installed rbvfit (or from the directory named by
    ``RBVFIT_AMD_LINELIST``).  Returns the number of rows, 0 when neither is available."""
    import os
    path = os.environ.get("RBVFIT_AMD_LINELIST")
    if not path:
        try:
            from importlib.resources import files
            cand = files("rbvfit").joinpath("lines/atom_full.dat")
            path = str(cand) if cand.is_file() else None
        except Exception:
            path = None
    return load_table(path) if path else 0


def table_size() -> int:
    return len(_LINES)


_cache = {}


def _waves():
    w = _cache.get("w")
    if w is None:
        w = _cache["w"] = np.array([r[1] for r in _LINES], dtype=np.float64)
    return w


def lookup(lambda_rest: float, method: str = "closest"):
    """Mirror of ``rb_setline(lambda_rest, method)``: returns dict(wave float64, fval float32,
    gamma float32, name).  'closest' = nearest wavelength in the table (rb_setline.py:55-56),
    'Exact' = within 1e-3 A (rb_setline.py:53-54).  With only the built-in subset loaded, a
    'closest' match farther than ``CLOSEST_TOLERANCE_A`` raises ``UnknownLineError`` instead of
    returning an unrelated transition."""
    waves = _waves()
    if method == "closest":
        i = int(np.abs(lambda_rest - waves).argmin())
        if not _full_table_loaded and abs(float(lambda_rest) - waves[i]) > CLOSEST_TOLERANCE_A:
            ion, w = _LINES[i][0], _LINES[i][1]
            raise UnknownLineError(
                f"no line within {CLOSEST_TOLERANCE_A} A of {lambda_rest} in the built-in table of {len(_LINES)} "
                f"transitions (nearest: {ion} {w}); call rbvfit_amd.atomic.load_table(<path to rbvfit's "
                "lines/atom_full.dat>) or register_line(ion, wrest, fval, gamma)")
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
