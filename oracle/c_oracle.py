"""ctypes wrapper of oracle/voigt_oracle.c (plain-C restatement; TEST INFRASTRUCTURE ONLY)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libvoigt_oracle.so")
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


class _Inst(C.Structure):
    _fields_ = [("P", C.c_int), ("L", C.c_int), ("K", C.c_int), ("lsf_mode", C.c_int), ("method", C.c_int),
                ("wave", _dp), ("flux", _dp), ("inv_sigma2", _dp), ("log_inv_sigma2", _dp),
                ("lambda0", _dp), ("gamma", _dp), ("f", _dp), ("zfac", _dp),
                ("N_idx", _ip), ("b_idx", _ip), ("v_idx", _ip), ("taps", _dp)]


def load():
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "voigt_oracle.c")):
        subprocess.run(["make", "-s", "-C", _HERE], check=True)
    lib = C.CDLL(_SO)
    lib.vo_rew.restype = C.c_double
    lib.vo_rew.argtypes = [C.c_double, C.c_double]
    lib.vo_rew_array.argtypes = [C.c_int, _dp, _dp, _dp]
    lib.vo_lnprob_batch.argtypes = [C.POINTER(_Inst), C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, C.c_int]
    lib.vo_model_flux.argtypes = [C.POINTER(_Inst), _dp, _dp, _dp, C.c_int]
    return lib


class COracle:
    """Holds the arrays alive and exposes lnprob_batch / model_flux / rew."""

    def __init__(self, instruments, lb, ub):
        """instruments: list of oracle.voigt_oracle.OracleInstrument."""
        self.lib = load()
        self._keep = []
        arr = (_Inst * len(instruments))()
        f64 = lambda a: self._hold(np.ascontiguousarray(a, dtype=np.float64))
        i32 = lambda a: self._hold(np.ascontiguousarray(a, dtype=np.int32))
        for k, inst in enumerate(instruments):
            d = inst.data
            taps = f64(d.taps if d.taps is not None and len(d.taps) else np.zeros(1))
            arr[k] = _Inst(len(inst.wave), d.n_lines, 0 if d.lsf_mode == 0 else len(d.taps), int(d.lsf_mode),
                           1 if d.voigt_method == "fast" else 0,
                           self._p(f64(inst.wave)), self._p(f64(inst.flux)), self._p(f64(inst.inv_sigma2)),
                           self._p(f64(inst.log_inv_sigma2)), self._p(f64(d.atomic_lambda0)),
                           self._p(f64(np.asarray(d.atomic_gamma).astype(np.float64))),
                           self._p(f64(np.asarray(d.atomic_f).astype(np.float64))), self._p(f64(d.z_factors)),
                           self._pi(i32(d.N_indices)), self._pi(i32(d.b_indices)), self._pi(i32(d.v_indices)),
                           self._p(taps))
        self.insts, self.n = arr, len(instruments)
        self.lb, self.ub = f64(lb), f64(ub)
        self.P = [len(i.wave) for i in instruments]

    def _hold(self, a):
        self._keep.append(a)
        return a

    @staticmethod
    def _p(a):
        return a.ctypes.data_as(_dp)

    @staticmethod
    def _pi(a):
        return a.ctypes.data_as(_ip)

    def lnprob_batch(self, thetas, nthreads=1):
        th = np.ascontiguousarray(np.atleast_2d(thetas), dtype=np.float64)
        out = np.empty(len(th))
        self.lib.vo_lnprob_batch(self.insts, self.n, len(th), th.shape[1], self._p(th), self._p(self.lb),
                                 self._p(self.ub), self._p(out), int(nthreads))
        return out

    def model_flux(self, k, theta, convolved=True):
        th = np.ascontiguousarray(theta, dtype=np.float64)
        out = np.empty(self.P[k]); work = np.empty(2 * self.P[k])
        self.lib.vo_model_flux(C.byref(self.insts[k]), self._p(th), self._p(out), self._p(work), 1 if convolved else 0)
        return out


def rew(x, y):
    lib = load()
    x = np.ascontiguousarray(np.broadcast_to(x, np.broadcast(x, y).shape), dtype=np.float64).ravel()
    y = np.ascontiguousarray(np.broadcast_to(y, x.shape) if np.ndim(y) == 0 else y, dtype=np.float64).ravel()
    out = np.empty_like(x)
    lib.vo_rew_array(x.size, x.ctypes.data_as(_dp), y.ctypes.data_as(_dp), out.ctypes.data_as(_dp))
    return out
