// C ABI of the rbvfit_amd engine (see include/rbvfit_amd.h).  Host side: context, device
// residency of the static data, workspace, launches.  No torch types, no Python.
//
// ONE translation unit (the kernels are templates in headers; every instance is compiled once), laid out by concern:
//   capi_context.inc      Tuning knobs, Instrument, vp_ctx / vp_multi, workspace
//   capi_launch.inc       launch policy: which kernels a batch gets, enqueue_lnprob
//   capi_prearm.inc       pre-armed launches, the registry of live contexts
//   capi_setup.inc        vp_ctx_create ... vp_add_instrument, vp_lnprob_batch_device
//   capi_gather.inc       vp_gather_* (direct-write gather between ranks)
//   capi_lnprob_flux.inc  vp_lnprob_batch (host buffers), vp_model_flux_*
//   capi_samplers.inc     vp_stretch_run, vp_slice_run
//   capi_multi.inc        vp_multi_* (several device contexts, one process)
//   capi_misc.inc         test hooks, timing, introspection
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cerrno>
#include <sys/mman.h>
#include <unistd.h>
#include <chrono>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/rbvfit_amd.h"
#include "voigt_kernels.h"
#include "sampler_kernels.h"
#include "slice_kernels.h"

#include "capi_context.inc"
#include "capi_launch.inc"
#include "capi_prearm.inc"

extern "C" {

#include "capi_setup.inc"
#include "capi_gather.inc"
#include "capi_lnprob_flux.inc"
#include "capi_samplers.inc"
#include "capi_multi.inc"
#include "capi_misc.inc"

}  // extern "C"
