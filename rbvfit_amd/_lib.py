"""ctypes binding of librbvfit_amd.so (the C ABI declared in include/rbvfit_amd.h).

There is deliberately NO fallback: if the HIP library is missing or cannot be loaded the import
of the product path fails loudly (``RbvfitAmdLibraryError``).  Nothing under ``oracle/`` is ever
imported from here.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RBVFIT_AMD_LIB") or os.path.join(_HERE, "lib", "librbvfit_amd.so")

VP_OK, VP_EINVAL, VP_EHIP, VP_ESTATE, VP_ENOMEM, VP_ENAN = 0, 1, 2, 3, 4, 5
LSF_NONE, LSF_SCIPY_NEAREST, LSF_ASTROPY_EXTEND = 0, 1, 2
VOIGT_WOFZ, VOIGT_FAST = 0, 1


class RbvfitAmdLibraryError(ImportError):
    pass


class RbvfitAmdError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"rbvfit_amd error {code}: {message}")
        self.code = code


_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_ctx = C.c_void_p

# name -> (restype, argtypes); must list every symbol of include/rbvfit_amd.h
SIGNATURES = {
    "vp_version": (C.c_char_p, []),
    "vp_device_count": (C.c_int, []),
    "vp_ctx_create": (C.c_int, [C.POINTER(_ctx), C.c_int]),
    "vp_ctx_destroy": (C.c_int, [_ctx]),
    "vp_set_option": (C.c_int, [_ctx, C.c_char_p, C.c_long]),
    "vp_set_bounds": (C.c_int, [_ctx, C.c_int, _dp, _dp]),
    "vp_add_instrument": (C.c_int, [_ctx, C.c_int, _dp, _dp, _dp, _dp, C.c_int, _dp, _dp, _dp, _dp,
                                    _ip, _ip, _ip, C.c_int, _dp, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "vp_update_spectrum": (C.c_int, [_ctx, C.c_int, _dp, _dp, _dp]),
    "vp_lnprob_batch": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),     # (theta, out: plain addresses)
    "vp_lnprob_batch_device": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vp_gather_create": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vp_gather_connect": (C.c_int, [_ctx, C.c_void_p, C.c_int]),
    "vp_gather_connect_local": (C.c_int, [_ctx, C.POINTER(_ctx), C.c_int]),
    "vp_lnprob_gather_device": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "vp_gather_wait": (C.c_int, [_ctx, C.c_void_p]),
    "vp_gather_state": (C.c_int, [_ctx, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]),
    "vp_gather_destroy": (C.c_int, [_ctx]),
    "vp_model_flux_batch": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int, _dp, _dp, C.c_int]),
    "vp_model_flux_rowsum": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int, _dp, _dp, C.c_double, C.c_int, _dp]),
    "vp_model_flux_batch_device": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                             C.c_int, C.c_void_p]),
    "vp_model_flux_components": (C.c_int, [_ctx, C.c_int, C.c_int, C.c_int, _dp, _dp]),
    "vp_voigt_h": (C.c_int, [_ctx, C.c_int, _dp, C.c_int, _dp, _dp]),
    "vp_stretch_run": (C.c_int, [_ctx, C.c_int, C.c_int, _dp, _dp, C.c_int, C.c_int, C.c_double, C.c_uint64, C.c_uint64,
                                 _dp, _dp, C.POINTER(C.c_int64)]),
    "vp_slice_run": (C.c_int, [_ctx, C.c_int, C.c_int, _dp, _dp, C.c_int, C.c_int, _dp, C.POINTER(C.c_int), C.c_double,
                               C.c_int, C.c_int, C.c_uint64, C.c_uint64, _dp, _dp, _dp, C.POINTER(C.c_int64)]),
    "vp_multi_create": (C.c_int, [C.POINTER(_ctx), C.c_int, C.POINTER(C.c_int)]),
    "vp_multi_destroy": (C.c_int, [_ctx]),
    "vp_multi_n_devices": (C.c_int, [_ctx]),
    "vp_multi_ctx": (_ctx, [_ctx, C.c_int]),
    "vp_multi_set_bounds": (C.c_int, [_ctx, C.c_int, _dp, _dp]),
    "vp_multi_add_instrument": (C.c_int, [_ctx, C.c_int, _dp, _dp, _dp, _dp, C.c_int, _dp, _dp, _dp, _dp,
                                          _ip, _ip, _ip, C.c_int, _dp, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "vp_multi_lnprob_batch": (C.c_int, [_ctx, C.c_int, C.c_int, _dp, _dp]),
    "vp_multi_stretch_run": (C.c_int, [_ctx, C.c_int, C.c_int, _dp, _dp, C.c_int, C.c_int, C.c_double, C.c_uint64, C.c_uint64,
                                       _dp, _dp, C.POINTER(C.c_int64)]),
    "vp_multi_slice_run": (C.c_int, [_ctx, C.c_int, C.c_int, _dp, _dp, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int),
                                     C.c_double, C.c_int, C.c_int, C.c_uint64, C.c_uint64, _dp, _dp, _dp, C.POINTER(C.c_int64)]),
    "vp_multi_last_error": (C.c_char_p, [_ctx]),
    "vp_philox4x32": (None, [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "vp_profile_enable": (C.c_int, [_ctx, C.c_int]),
    "vp_profile_read": (C.c_int, [_ctx, _dp, _dp, _dp, C.POINTER(C.c_int)]),
    "vp_ctx_stream": (C.c_void_p, [_ctx]),
    "vp_num_instruments": (C.c_int, [_ctx]),
    "vp_ndim": (C.c_int, [_ctx]),
    "vp_instrument_pixels": (C.c_int, [_ctx, C.c_int]),
    "vp_device_id": (C.c_int, [_ctx]),
    "vp_last_launch_kind": (C.c_int, [_ctx]),
    "vp_last_walker_split": (C.c_int, [_ctx]),
    "vp_prearm_counts": (C.c_int, [_ctx, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "vp_last_farfield_info": (C.c_int, [_ctx, C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "vp_last_error": (C.c_char_p, [_ctx]),
}

_lib = None


def _preload_shared_hip_runtime():
    """One HIP runtime per process.  PyTorch wheels bundle their own ``libamdhip64.so`` (SONAME
    ``libamdhip64.so.7``, found through an ``$ORIGIN`` rpath under the file name ``libamdhip64.so``).
    If this library pulled in ``/opt/rocm/lib/libamdhip64.so.7`` first, a later ``import torch`` would
    load the bundled copy as a SECOND runtime and fail with "No HIP GPUs are available"; the other
    order works because our NEEDED entry matches the SONAME already loaded.  So when a torch wheel is
    installed (found without importing it) its runtime is loaded first and both orders end up on the
    same copy.  ``RBVFIT_AMD_HIP_RUNTIME=system`` keeps the system runtime."""
    import sys
    if "torch" in sys.modules or os.environ.get("RBVFIT_AMD_HIP_RUNTIME", "").lower() == "system":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(path):
            C.CDLL(path, mode=C.RTLD_GLOBAL)
    except Exception:              # no torch / unusual layout: the system runtime is used
        pass


def load():
    """Load the shared library once; raise RbvfitAmdLibraryError if it is absent/unloadable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RbvfitAmdLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  rbvfit_amd has no CPU fallback.")
    _preload_shared_hip_runtime()
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # missing libamdhip64 etc.
        raise RbvfitAmdLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RbvfitAmdLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(lib, ctx, rc):
    if rc != VP_OK:
        msg = lib.vp_last_error(ctx)
        raise RbvfitAmdError(rc, msg.decode() if msg else "unknown error")
