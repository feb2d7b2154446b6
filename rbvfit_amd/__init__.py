"""rbvfit_amd -- MI355X-native Voigt forward model + log-likelihood engine behind rbvfit's
``CompiledVoigtModel.model_flux`` / ``vfit.lnprob`` interface.  See DESIGN.md."""
from ._lib import (RbvfitAmdError, RbvfitAmdLibraryError, LSF_NONE, LSF_SCIPY_NEAREST,
                   LSF_ASTROPY_EXTEND, VOIGT_WOFZ, VOIGT_FAST, LIB_PATH)
from .engine import Engine, MultiEngine, device_count
from . import model, vfit, sampler, dist, lsf, atomic, workloads, cog  # noqa: F401  (host mirror of the reference interface)

__version__ = "0.4.0"     # = RBVFIT_AMD_VERSION of include/rbvfit_amd.h = what vp_version() reports
__all__ = ["Engine", "MultiEngine", "device_count", "RbvfitAmdError", "RbvfitAmdLibraryError", "LIB_PATH",
           "LSF_NONE", "LSF_SCIPY_NEAREST", "LSF_ASTROPY_EXTEND", "VOIGT_WOFZ", "VOIGT_FAST"]
