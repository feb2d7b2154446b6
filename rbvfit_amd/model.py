"""Host-side mirror of the reference's model interface for the accelerated path.

Mirrors (same names, argument meaning and error behaviour, reduced to what the path needs):
  FitConfiguration.add_system            core/fit_configuration.py:318-351
  VoigtModel(config, FWHM, voigt_method) core/voigt_model.py:334-384  (+ _cache_atomic_parameters
                                         :386-412, _setup_fast_mapping :414-442, _setup_kernel :444-464)
  VoigtModel.compile() -> CompiledVoigtModel.model_flux(theta, wavelength)   :466-507, :295-315
  CompiledModelData                      :265-280 (astropy kernel object -> taps + lsf_mode)

Everything here is setup; evaluation goes through ``rbvfit_amd.Engine`` (HIP).
"""
from __future__ import annotations

import hashlib
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Union

import numpy as np

from . import _lib as L
from . import atomic
from .engine import Engine
from .lsf import gaussian_taps


# ------------------------------------------------------------------------------------------------
# physics description (minimal FitConfiguration)
# ------------------------------------------------------------------------------------------------
@dataclass
class IonGroup:
    ion_name: str
    transitions: List[float]
    components: int


@dataclass
class AbsorptionSystem:
    redshift: float
    ion_groups: List[IonGroup] = field(default_factory=list)


class FitConfiguration:
    """systems (z) -> ion groups (ion, transitions, components); theta layout is global
    [N_0..N_{C-1} | b_0.. | v_0..] (core/voigt_model.py:440-442, parameter_manager.py:123-126)."""

    def __init__(self, FWHM=None):
        self.systems: List[AbsorptionSystem] = []
        self.instrumental_params = {}
        if FWHM is not None:
            self.instrumental_params["FWHM"] = FWHM

    def add_system(self, z: float, ion: str = "auto", transitions: Sequence[float] = None,
                   components: int = 1) -> None:
        if transitions is None:
            raise ValueError("transitions list cannot be None")
        if components < 1:
            raise ValueError("components must be >= 1")
        system = next((s for s in self.systems if s.redshift == z), None)
        if system is None:
            system = AbsorptionSystem(float(z))
            self.systems.append(system)
        if any(g.ion_name == ion for g in system.ion_groups):
            raise ValueError(f"Ion {ion} already exists in system at z={z}")
        # the reference snaps user wavelengths to the database values (fit_configuration.py:104-126, T13)
        snapped = [float(atomic.lookup(w, "closest")["wave"]) for w in transitions]
        system.ion_groups.append(IonGroup(str(ion), snapped, int(components)))

    def validate(self) -> None:
        if not self.systems:
            raise ValueError("No absorption systems defined")

    @property
    def total_components(self) -> int:
        return sum(g.components for s in self.systems for g in s.ion_groups)


# ------------------------------------------------------------------------------------------------
# compiled tables
# ------------------------------------------------------------------------------------------------
@dataclass
class CompiledModelData:
    atomic_lambda0: np.ndarray
    atomic_gamma: np.ndarray      # float32 like the reference (rb_setline.py:44)
    atomic_f: np.ndarray          # float32 (rb_setline.py:42)
    z_factors: np.ndarray
    N_indices: np.ndarray
    b_indices: np.ndarray
    v_indices: np.ndarray
    taps: Optional[np.ndarray]    # kernel.array of the reference's astropy kernel, or None
    lsf_mode: int
    n_lines: int
    total_components: int
    voigt_method: str = "wofz"

    def engine_kwargs(self):
        """Arguments of Engine.add_instrument after the table part (T1: float32 -> widened)."""
        return dict(lambda0=np.asarray(self.atomic_lambda0, dtype=np.float64),
                    gamma=np.asarray(self.atomic_gamma).astype(np.float64),
                    f=np.asarray(self.atomic_f).astype(np.float64),
                    zfac=np.asarray(self.z_factors, dtype=np.float64),
                    N_idx=np.asarray(self.N_indices, dtype=np.int32),
                    b_idx=np.asarray(self.b_indices, dtype=np.int32),
                    v_idx=np.asarray(self.v_indices, dtype=np.int32),
                    taps=self.taps, lsf_mode=self.lsf_mode,
                    voigt_method=L.VOIGT_FAST if self.voigt_method == "fast" else L.VOIGT_WOFZ)


def tables_from_rbvfit(compiled) -> CompiledModelData:
    """Adapter for a *reference* ``CompiledVoigtModel`` (or its ``.data``): duck-typed, rbvfit and
    astropy are not imported here.  Gaussian1DKernel -> scipy 'nearest' branch, any other kernel
    object -> the astropy 'extend' (normalising) branch (core/voigt_model.py:220-230)."""
    d = getattr(compiled, "data", compiled)
    k = getattr(d, "kernel", None)
    if k is None:
        taps, mode = None, L.LSF_NONE
    else:
        taps = np.asarray(k.array, dtype=np.float64)
        mode = L.LSF_SCIPY_NEAREST if type(k).__name__ == "Gaussian1DKernel" else L.LSF_ASTROPY_EXTEND
    return CompiledModelData(np.asarray(d.atomic_lambda0), np.asarray(d.atomic_gamma), np.asarray(d.atomic_f),
                             np.asarray(d.z_factors), np.asarray(d.N_indices), np.asarray(d.b_indices),
                             np.asarray(d.v_indices), taps, mode, int(d.n_lines), int(d.total_components),
                             getattr(d, "voigt_method", "wofz"))


def mean_fwhm_pixels(FWHM_vel_kms: float, wave_obs_grid) -> float:
    """FWHM in km/s -> mean FWHM in pixels of a wavelength grid (core/voigt_model.py:33-58): the mean over the grid
    of lambda FWHM / c divided by ``np.gradient`` of the grid.  Same checks and messages as the reference."""
    wave_obs_grid = np.asarray(wave_obs_grid)
    if np.any(wave_obs_grid <= 0):
        raise ValueError("Wavelength grid must be strictly positive.")
    if len(wave_obs_grid) < 2:
        raise ValueError("Wavelength grid must have at least two points.")
    c_kms = 299792.458
    delta_lambda = np.gradient(wave_obs_grid)
    fwhm_lambda = wave_obs_grid * FWHM_vel_kms / c_kms
    return float(np.mean(fwhm_lambda / delta_lambda))


_ASTROPY_NORMALIZES = "unknown"


def astropy_normalizes_gaussian() -> Optional[bool]:
    """Does the installed astropy hand out sum-normalised ``Gaussian1DKernel`` arrays?  True / False from the package
    itself (the kernel the reference would build, core/voigt_model.py:464: an 8-sigma truncation loses 6e-5 of the sum
    unless the constructor renormalises), None where astropy cannot be imported.  Asked once per process."""
    global _ASTROPY_NORMALIZES
    if _ASTROPY_NORMALIZES == "unknown":
        try:
            from astropy.convolution import Gaussian1DKernel
            _ASTROPY_NORMALIZES = bool(abs(float(np.sum(Gaussian1DKernel(2.0).array)) - 1.0) < 1e-9)
        except Exception:
            _ASTROPY_NORMALIZES = None
    return _ASTROPY_NORMALIZES


class VoigtModel:
    """Single-instrument model description.  ``FWHM`` is in pixels, as a string or float, or None
    for no LSF; ``kernel_taps`` supplies a tabulated LSF (the reference's 'COS' / CustomKernel
    branch -> normalising 'extend' convolution; ``FWHM='COS'`` itself needs linetools' tables, which
    this package does not carry: pass their samples as ``kernel_taps``).

    ``normalize_kernel``: whether the Gaussian taps are divided by their sum (trap T2: for FWHM '6.5' the two
    differ by sum(taps) = 1 - 2.8e-5 on every pixel).  False is what ``Gaussian1DKernel(...).array`` holds under
    astropy 4.3.1 -- the only astropy the reference could be run with here, the version every golden fixture and
    SURVEY anchor was made with (``tests/golden/taps.npz``); True gives the same samples divided by their sum
    (pinned against the same fixture, ``tests/test_host_logic.py``).  The default, None, asks the astropy that is
    installed next to this package -- the one the reference itself would build its kernel with -- whether its
    ``Gaussian1DKernel`` is sum-normalised (``astropy_normalizes_gaussian``) and follows it; where no astropy is
    importable it is False, the pinned behaviour.  ``VoigtModel.normalize_kernel`` records what was used.  The
    engine takes taps as data either way."""

    def __init__(self, config: FitConfiguration, FWHM: Union[str, float, None] = "6.5",
                 voigt_method: str = "wofz", kernel_taps: Optional[Sequence[float]] = None,
                 normalize_kernel: Optional[bool] = None):
        if voigt_method not in ("wofz", "fast"):
            raise ValueError(f"voigt_method must be one of ('wofz', 'fast'), got '{voigt_method}'")
        self.voigt_method = voigt_method
        self.config = config
        self.config.validate()
        if normalize_kernel is None:
            normalize_kernel = bool(astropy_normalizes_gaussian())        # (None -- no astropy here -- counts as False)
        self.normalize_kernel = bool(normalize_kernel)
        self.FWHM = config.instrumental_params.get("FWHM", FWHM)
        if kernel_taps is not None:
            self.taps, self.lsf_mode = np.asarray(kernel_taps, dtype=np.float64), L.LSF_ASTROPY_EXTEND
        elif self.FWHM is None:
            self.taps, self.lsf_mode = None, L.LSF_NONE
        else:
            if isinstance(self.FWHM, str) and self.FWHM.strip().upper() == "COS":
                # core/voigt_model.py:448-460 builds this kernel from linetools' COS tables
                raise ImportError("COS LSF requires linetools package; rbvfit_amd does not carry its tables -- "
                                  "pass the tabulated LSF samples as kernel_taps=")
            self.taps = gaussian_taps(float(self.FWHM), normalize=normalize_kernel)
            self.lsf_mode = L.LSF_SCIPY_NEAREST
        # line order: system -> ion_group -> transition -> component   (voigt_model.py:391-401)
        lam, gam, fv, zf, idx = [], [], [], [], []
        offset = 0
        for system in config.systems:
            for g in system.ion_groups:
                for wavelength in g.transitions:
                    info = atomic.lookup(wavelength, "closest")
                    for comp in range(g.components):
                        lam.append(info["wave"]); gam.append(info["gamma"]); fv.append(info["fval"])
                        zf.append(1.0 + system.redshift)
                        idx.append(offset + comp)                # :428-437
                offset += g.components
        self.total_components = offset
        self.n_lines = len(lam)
        self.atomic_lambda0 = np.array(lam, dtype=np.float64)
        self.atomic_gamma = np.array(gam, dtype=np.float32)
        self.atomic_f = np.array(fv, dtype=np.float32)
        self.z_factors = np.array(zf, dtype=np.float64)
        self.N_indices = np.array(idx, dtype=np.int64)
        self.b_indices = self.N_indices + self.total_components   # :441
        self.v_indices = self.N_indices + 2 * self.total_components   # :442

    def compile(self, verbose: bool = False, device_id: int = 0) -> "CompiledVoigtModel":
        data = CompiledModelData(self.atomic_lambda0.copy(), self.atomic_gamma.copy(), self.atomic_f.copy(),
                                 self.z_factors.copy(), self.N_indices.copy(), self.b_indices.copy(),
                                 self.v_indices.copy(), None if self.taps is None else self.taps.copy(),
                                 self.lsf_mode, self.n_lines, self.total_components, self.voigt_method)
        if verbose:
            print(f"Compiling VoigtModel: {3 * self.total_components} parameters, {self.n_lines} lines")
        return CompiledVoigtModel(data, device_id)

    def evaluate(self, theta, wavelength, return_components: bool = False, return_unconvolved: bool = False):
        """Analysis-time evaluation (core/voigt_model.py:509-558).  Note T10: the reference's
        ``evaluate`` always uses the exact Voigt function, whatever ``voigt_method`` says.
        ``return_components=True`` returns the reference's dictionary (voigt_model.py:232-259):
        ``flux`` (convolved unless ``return_unconvolved``), ``components`` (list of the L unconvolved
        per-line profiles exp(-tau_l)) and ``component_info`` (one dict per line: line_index, lambda0,
        gamma, f_value, z_total, N_value, b_value, v_value)."""
        data = CompiledModelData(self.atomic_lambda0, self.atomic_gamma, self.atomic_f, self.z_factors,
                                 self.N_indices, self.b_indices, self.v_indices, self.taps, self.lsf_mode,
                                 self.n_lines, self.total_components, "wofz")
        cm = CompiledVoigtModel(data)
        try:
            flux = cm.model_flux(theta, wavelength, convolved=not return_unconvolved)
            if not return_components:
                return flux
            theta = np.asarray(theta, dtype=np.float64)
            if theta.ndim != 1:
                raise ValueError("return_components=True takes a single theta, as the reference does")
            comps = cm.components(theta, wavelength)
            return {"flux": flux, "components": [comps[i] for i in range(self.n_lines)],
                    "component_info": component_info(data, theta)}
        finally:
            cm.close()


def component_info(data: "CompiledModelData", theta) -> list:
    """The per-line bookkeeping of ``_evaluate_compiled_model(..., return_components=True)``
    (voigt_model.py:240-253), formed with the reference's operations (:192-200)."""
    theta = np.asarray(theta, dtype=np.float64)
    N_linear = 10 ** theta[np.asarray(data.N_indices)]
    b_values = theta[np.asarray(data.b_indices)]
    v_values = theta[np.asarray(data.v_indices)]
    z_total = np.asarray(data.z_factors) * (1 + v_values / 299792.458) - 1
    return [{"line_index": i, "lambda0": float(data.atomic_lambda0[i]), "gamma": float(data.atomic_gamma[i]),
             "f_value": float(data.atomic_f[i]), "z_total": float(z_total[i]), "N_value": float(N_linear[i]),
             "b_value": float(b_values[i]), "v_value": float(v_values[i])} for i in range(int(data.n_lines))]


class CompiledVoigtModel:
    """GPU-backed ``model_flux(theta, wavelength)``.  ``theta`` may be (D,) -> (P,) as in the
    reference, or a batch (W, D) -> (W, P).  A model-only engine (unit weights) is cached per
    wavelength grid.  Not picklable across processes by design (no fork Pool with HIP)."""

    def __init__(self, data: CompiledModelData, device_id: int = 0):
        self.data = data
        self.device_id = device_id
        self._engines = {}

    def _engine_for(self, wavelength: np.ndarray) -> Engine:
        wave = np.ascontiguousarray(wavelength, dtype=np.float64)
        key = (wave.size, hashlib.blake2b(wave.tobytes(), digest_size=12).hexdigest())
        eng = self._engines.get(key)
        if eng is None:
            D = 3 * self.data.total_components
            eng = Engine(self.device_id)
            eng.set_bounds(np.full(D, -np.inf), np.full(D, np.inf))
            ones = np.ones_like(wave)
            eng.add_instrument(wave, ones, ones, np.zeros_like(wave), **self.data.engine_kwargs())
            if len(self._engines) >= 4:                      # small LRU: drop the oldest grid
                self._engines.pop(next(iter(self._engines))).close()
            self._engines[key] = eng
        return eng

    def close(self):
        for eng in self._engines.values():
            eng.close()
        self._engines = {}

    def model_flux(self, theta, wavelength, convolved: bool = True) -> np.ndarray:
        theta = np.asarray(theta, dtype=np.float64)
        out = self._engine_for(np.asarray(wavelength)).model_flux(0, theta, convolved=convolved)
        return out[0] if theta.ndim == 1 else out

    def __call__(self, theta, wavelength):
        return self.model_flux(theta, wavelength)

    def equivalent_width(self, theta, wavelength) -> np.ndarray:
        """``np.trapz(1 - model_flux(theta, wavelength), x=wavelength)`` per theta row (compute_cog.py:56-60) with the integral
        formed on the GPU: (W,) doubles come back instead of (W, P) rows.  Trapezoid weights c_i = (x_{i+1} - x_{i-1}) / 2
        (half intervals at the ends): EW = sum c_i - sum c_i flux_i."""
        theta = np.asarray(theta, dtype=np.float64)
        x = np.asarray(wavelength, dtype=np.float64)
        c = np.zeros_like(x)
        d = np.diff(x)
        c[:-1] += 0.5 * d
        c[1:] += 0.5 * d
        out = self._engine_for(x).model_flux_rowsum(0, theta, c, c0=float(np.sum(c)))
        return float(out[0]) if theta.ndim == 1 else out

    def components(self, theta, wavelength) -> np.ndarray:
        """Per-line unconvolved flux exp(-tau_l): (L, P) for one theta, (W, L, P) for a batch
        (``_evaluate_compiled_model(..., return_components=True)['components']``, voigt_model.py:232-238)."""
        theta = np.asarray(theta, dtype=np.float64)
        out = self._engine_for(np.asarray(wavelength)).model_flux_components(0, theta, self.data.n_lines)
        return out[0] if theta.ndim == 1 else out

    def __getstate__(self):
        raise TypeError("CompiledVoigtModel holds a GPU context and cannot be pickled; "
                        "run the sampler with use_pool=False (batched lnprob replaces the Pool)")
