/*
 * rbvfit_amd.h -- C ABI of the MI355X-native Voigt forward-model + log-likelihood engine.
 *
 * This is the drop-in boundary for ONE path of rongmon/rbvfit: the per-walker evaluation
 *     CompiledVoigtModel.model_flux(theta, wave)      src/rbvfit/core/voigt_model.py:295-311
 *       -> _evaluate_compiled_model                   src/rbvfit/core/voigt_model.py:162-261
 *       -> _vectorized_voigt_tau                      src/rbvfit/core/voigt_model.py:100-159
 *     vfit.lnprior / lnlike / lnprob                  src/rbvfit/vfit_mcmc.py:291-353
 * evaluated for a whole batch of walkers per call on one MI355X (gfx950).
 *
 * The reference is pure Python, so "the reference's FFI for this path" is a ctypes binding;
 * INTEGRATION.md shows the stub a maintainer would add.  Plain pointers and sizes only.
 *
 * Conventions
 *   - every function returns 0 on success, a non-zero VP_E* code otherwise; the text of the
 *     last failure is available from vp_last_error().  Errors are never signalled through the
 *     numeric outputs.
 *   - numeric conventions INSIDE outputs follow the reference: a theta row outside [lb,ub] gets
 *     lnprob = -inf and its model is not evaluated (vfit_mcmc.py:291-295,348-353); NaN produced
 *     by the arithmetic (e.g. error == 0) propagates as NaN.
 *   - all arrays are C-contiguous; "host" pointers are ordinary process memory owned by the
 *     caller and only read/written during the call; "device" pointers are HIP device memory on
 *     the context's GPU.
 *   - the library copies everything passed to vp_set_bounds / vp_add_instrument.
 *   - a context may be used from any thread, one call at a time (internal mutex); it is NOT
 *     fork-safe (create it in the process that uses it; rbvfit callers pass use_pool=False).
 */
#ifndef RBVFIT_AMD_H
#define RBVFIT_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vp_ctx vp_ctx;

enum {
    VP_OK = 0,
    VP_EINVAL = 1,      /* bad argument (shape, index out of range, NULL) */
    VP_EHIP = 2,        /* a HIP runtime call failed */
    VP_ESTATE = 3,      /* call order (e.g. lnprob before bounds/instruments are set) */
    VP_ENOMEM = 4,
    VP_ENAN = 5      /* a proposal's lnprob was NaN (vp_stretch_run, vp_slice_run) */
};

/* LSF dispatch branches of core/voigt_model.py:220-230 */
enum {
    VP_LSF_NONE = 0,            /* kernel is None: no convolution                          (:221) */
    VP_LSF_SCIPY_NEAREST = 1,   /* Gaussian1DKernel -> ndimage.convolve1d(mode='nearest')  (:224) */
    VP_LSF_ASTROPY_EXTEND = 2   /* CustomKernel -> astropy convolve(boundary='extend'):
                                   taps are divided by their sum                     (:227,230) */
};

/* voigt_method of VoigtModel (core/voigt_model.py:359-364) */
enum { VP_VOIGT_WOFZ = 0, VP_VOIGT_FAST = 1 };

/* Number of HIP devices visible to the process (0 if none / HIP unusable). */
int vp_device_count(void);

/* Create a context bound to one GPU.  Replaces: construction of `vfit` (vfit_mcmc.py:127-197). */
int vp_ctx_create(vp_ctx** out, int device_id);
int vp_ctx_destroy(vp_ctx* ctx);

/* Box prior.  Replaces: vfit.lb / vfit.ub used by lnprior (vfit_mcmc.py:291-295).
 * Fixes D = ndim for the context. */
int vp_set_bounds(vp_ctx* ctx, int D, const double* lb, const double* ub);

/* Add one instrument = one entry of vfit.instrument_data after _compile_models
 * (vfit_mcmc.py:234-259) together with the CompiledModelData of its model
 * (core/voigt_model.py:265-280).
 *   P, wave/flux/inv_sigma2/log_inv_sigma2 : spectrum and precomputed weights (T4: computed by the
 *       host in the dtype the reference would use, then widened to double)
 *   L, lambda0/gamma/f/zfac                : per-line atomic_lambda0, atomic_gamma, atomic_f
 *       (float32-rounded then widened, rb_setline.py:42,44) and z_factors
 *   N_idx/b_idx/v_idx                      : theta indices per line (core/voigt_model.py:440-442)
 *   K, taps                                : kernel.array (K odd); K = 0 or taps = NULL with
 *                                            lsf_mode = VP_LSF_NONE means no LSF
 * NaN wavelength samples: such a pixel's model is NaN in the reference.  With VP_LSF_SCIPY_NEAREST (and without an LSF) it poisons
 * exactly the outputs its K taps reach and lnprob is NaN, as scipy's convolve1d does (:224).  With VP_LSF_ASTROPY_EXTEND the engine does what
 * astropy's convolve does by default (nan_treatment='interpolate', :227,230): the NaN pixel is left out and every output is
 * divided by the kernel weight of the samples that were used -- the pixel itself comes out as the weighted mean of its
 * neighbours, lnprob stays finite; a gap of K or more consecutive NaN samples keeps NaN outputs (tests/golden/nan_wave_*.npz,
 * nan_semantics.npz: vectors made by the reference).
 * Returns the instrument's index in *inst_index (may be NULL). */
int vp_add_instrument(vp_ctx* ctx, int P, const double* wave, const double* flux,
                      const double* inv_sigma2, const double* log_inv_sigma2,
                      int L, const double* lambda0, const double* gamma, const double* f,
                      const double* zfac, const int32_t* N_idx, const int32_t* b_idx,
                      const int32_t* v_idx, int K, const double* taps, int lsf_mode,
                      int voigt_method, int* inst_index);

/* Replace the observed flux / weights of an instrument (same P); tables and LSF unchanged. */
int vp_update_spectrum(vp_ctx* ctx, int inst, const double* flux, const double* inv_sigma2,
                       const double* log_inv_sigma2);

/* lnprob for a batch of walkers.  Replaces: map(vfit.lnprob, theta_rows) (vfit_mcmc.py:348-353,
 * the sampler fan-out of :408-440).  theta is row-major (W, D) host memory, out is (W,).
 * Batches of up to 1 MiB of theta are read / written by the kernels straight from / to a pinned staging buffer, and the call
 * returns as soon as every output row has arrived there (the rows are pre-set to a NaN bit pattern that no arithmetic produces and
 * each is written exactly once; inputs carrying that payload make the call wait on a stream-written completion word instead):
 * `out` is complete when it returns, as with any synchronous call.
 * Pre-armed launches (option "prearm": -1 default, 0 never, 1 always; "prearm_us", default 500): for batches that are one
 * walker_kernel launch, a call of the previous call's shape that came within prearm_us / 2 of its return leaves the launch for the NEXT batch
 * of this shape on the GPU.  It has done everything that does not depend on theta and waits, holding its compute units, for the
 * host to push that batch -- no longer than the caller's own rhythm suggests: 1.5 x the recent gap between its calls + 10 us (at
 * least 20 us, at most prearm_us; "prearm" = 1 waits the whole prearm_us), and a caller whose launches expire unused is not
 * pre-armed for again for 8, 16, ... calls.  Work that anything else puts on the GPU meanwhile (another library of the process,
 * another process) waits that long at most.  The next call then costs no launch and no read over PCIe: it writes
 * theta and a go word into the launch's slots in device memory through the PCIe BAR (GPUs without a large BAR: no pre-armed launches) and
 * waits for the rows.  Every other entry point of the context, every call on another context of this process on the same
 * GPU, vp_ctx_destroy and process exit send a waiting launch away first; a launch that expired, or a batch of another shape, falls back to the ordinary launch.  Results do not
 * depend on which way a batch was started.  vp_prearm_counts reports how often each happened. */
int vp_lnprob_batch(vp_ctx* ctx, int W, int D, const double* theta, double* out);

/* Same, operands already resident on the context's GPU; enqueued on `hip_stream` (a hipStream_t,
 * NULL = the context's own stream) and NOT synchronised: the caller orders later work on the
 * same stream or synchronises it.  Successive calls on one context must be stream-ordered
 * (they share the context's workspace). */
int vp_lnprob_batch_device(vp_ctx* ctx, int W, int D, const double* d_theta, double* d_out,
                           void* hip_stream);

/* ---- direct-write gather: the per-pass exchange of a walker-sharded ensemble, one process per GPU ----
 * Replaces, for the one-launch batches, the collective behind every pass of the reference's fan-out (vfit_mcmc.py:35-49,
 * 408-440: every worker's results return to the sampler): each rank owns a (world, W) vector of device memory that its peers
 * map (hipIpc handles, carried by the job's own channel -- rbvfit_amd.dist.DirectGather uses torch.distributed's object
 * all-gather).  vp_lnprob_gather_device is vp_lnprob_batch_device with the output written by the kernel itself into this rank's
 * block of EVERY rank's vector (8 bytes per walker and rank); the next pass's first launch raises this rank's flag in every
 * peer and waits, on the device, for its peers' flags of the pass before -- the dependency a blocking all-gather
 * states, without a collective launch or a host wait.  (Batches that take several launches: the first one handshakes, the final
 * reduction writes into the vectors.)
 *   vp_gather_create    allocates this rank's vector and flags; handles_out: 2 x 64 bytes (hipIpcMemHandle_t of both), zeros
 *                       when world == 1.  1 <= world <= 8.
 *   vp_gather_connect   handles_all: world x 2 x 64 bytes, rank-major (this rank's own entry is not opened).  shared_device != 0:
 *                       some ranks share a GPU -- a launch that waits for its peers inside its workgroups needs their launches
 *                       to start while it holds its wave slots, which one GPU cannot promise to several ranks; a pass is then
 *                       [one-wave launch that waits for the peers' flags of the pass before] [the pass] [one-wave launch that
 *                       raises this rank's flag of this pass in every peer]: every wait depends on earlier-enqueued work only.
 *   vp_gather_wait      enqueues a one-wave kernel that returns once every rank's block of the LAST pass has landed here.
 *   vp_gather_state     the device pointer of this rank's (world, W) vector; *timed_out != 0 if a device-side wait gave up
 *                       (~ a second of polling: a peer that never ran), synchronises.  Either pointer may be NULL. */
int vp_gather_create(vp_ctx* ctx, int W, int world, int rank, void* handles_out);
int vp_gather_connect(vp_ctx* ctx, const void* handles_all, int shared_device);
/* The same for ranks that are contexts of ONE process (several GPUs driven by one process, or -- tests -- several contexts on one
 * GPU: a process cannot open its own IPC handles): peers[r] is rank r's context (this rank's own entry is ignored), each with a
 * gather of the same W and world already created; their vectors are used through their device pointers (peer access between the
 * devices is enabled where it is not).  Every rank calls it; the peers must outlive this rank's gather. */
int vp_gather_connect_local(vp_ctx* ctx, vp_ctx* const* peers, int shared_device);
int vp_lnprob_gather_device(vp_ctx* ctx, int W, int D, const double* d_theta, void* hip_stream);
int vp_gather_wait(vp_ctx* ctx, void* hip_stream);
int vp_gather_state(vp_ctx* ctx, double** d_gathered, int* timed_out);
int vp_gather_destroy(vp_ctx* ctx);

/* Model flux for a batch.  Replaces: CompiledVoigtModel.model_flux per row
 * (core/voigt_model.py:295-311).  out is row-major (W, P) host memory.  convolved = 0 returns the
 * profile before the LSF (VoigtModel.evaluate(return_unconvolved=True), :509-558).  The prior is
 * not consulted (model_flux has none).  Instruments with >= 8 lines take the lines far from a 192-pixel block from the block's
 * expansion as the lnprob launches do, by the same batch-size rule (option "flux_farfield": -1 / 0 / 1; flux within the 1e-12 contract
 * either way). */
int vp_model_flux_batch(vp_ctx* ctx, int inst, int W, int D, const double* theta, double* out,
                        int convolved);
/* out[w] = c0 - sum_p weights[p] * model_flux(theta_w)[p]  for a batch: the model rows never leave the GPU, W doubles come back.
 * Replaces the equivalent-width integral of the curve-of-growth grid, np.trapz(1 - flux, x=wave) per (N, b) pair
 * (compute_cog.py:56-60, 75-86): weights = the trapezoid weights of the wavelength grid, c0 = their sum (rbvfit_amd.cog).
 * weights (P) and out (W) are host memory. */
int vp_model_flux_rowsum(vp_ctx* ctx, int inst, int W, int D, const double* theta, const double* weights, double c0,
                         int convolved, double* out);
int vp_model_flux_batch_device(vp_ctx* ctx, int inst, int W, int D, const double* d_theta,
                               double* d_out, int convolved, void* hip_stream);

/* Per-line ("component") profiles.  Replaces: _evaluate_compiled_model(return_components=True)
 * (core/voigt_model.py:232-259): out is row-major (W, L, P) host memory holding exp(-tau_l) of each
 * line l, UNCONVOLVED -- exactly what the reference returns despite its comment (SURVEY trap T11). */
int vp_model_flux_components(vp_ctx* ctx, int inst, int W, int D, const double* theta, double* out);

/* H(a_i, x_j) = Re w(x_j + i a_i) on the device for a grid (host buffers; out is row-major
 * (na, nx)).  Test hook for the Faddeeva tiers that replace scipy.special.wofz at the call site
 * core/voigt_model.py:156: the production tier logic is used (tier chosen per wavefront = 64
 * consecutive x_j of one a_i). */
int vp_voigt_h(vp_ctx* ctx, int na, const double* a, int nx, const double* x, double* out);

/* Device-resident ensemble sampler.  Replaces: the walker loop the reference delegates to emcee
 * (vfit_mcmc.py:408-423 EnsembleSampler construction, :536-540 run_mcmc): `nsteps` iterations of
 * the affine-invariant stretch move (scale `a`, emcee's default 2.0) in red-blue form -- each
 * half-ensemble is proposed, evaluated (one lnprob batch in HBM) and accepted/rejected on the
 * GPU; nothing crosses PCIe until the call returns.
 *   pos        (W, D) host, in: start positions, out: final positions.   W even, W >= 2.
 *   lnprob     (W) host, in (when have_lnprob != 0) / out: lnprob of pos.
 *   seed/step0 Philox4x32-10 key and the index of the first step: draws are a pure function of
 *              (seed, step, half, walker), so run(n1) then run(n2, step0 = n1) equals run(n1 + n2).
 *   chain      (nsteps, W, D) host or NULL;  chain_lnprob (nsteps, W) host or NULL: state after each step.
 *   naccepted  (W) host or NULL: accepted proposals per walker, ADDED to the values passed in.
 * Returns VP_ENAN (state and outputs undefined) if a proposal's lnprob is NaN -- emcee raises
 * "Probability function returned NaN" there.
 * Where a half-step is one launch and two half-ensembles of workgroups fit the GPU at once, consecutive half-steps are enqueued on
 * two streams of the context and overlap: a walker's workgroup waits (bounded; VP_EHIP if a wait gives up) for its partner's row
 * of the half-step before, not for that whole launch; with D <= 6 a walker's row, lnprob and version share one 64-byte line per
 * buffer, so that the load which shows a partner's version also holds its row (option "stretch_mailbox": 1 / 0).  The chain does not
 * depend on either (option "stretch_overlap": -1 / 0 / 1). */
int vp_stretch_run(vp_ctx* ctx, int W, int D, double* pos, double* lnprob, int have_lnprob, int nsteps, double a,
                   uint64_t seed, uint64_t step0, double* chain, double* chain_lnprob, int64_t* naccepted);

/* Device-resident ensemble slice sampler: the walker loop rbvfit delegates to zeus when the fitter is built with
 * sampler='zeus' (vfit_mcmc.py:425-440 EnsembleSampler construction, :536-540 run_mcmc): `nsteps` iterations of
 * ensemble slice sampling with the differential move.  Per iteration the ensemble is split at random in two
 * halves; every walker of the active half slices along mu * 2.38/sqrt(2 D) * (X_l - X_m) (l != m from the other
 * half) with stepping-out and shrinking.  The ragged sets of still-active walkers are compacted ON the GPU: each
 * round is one lnprob batch of slice_rows * W/2 rows ("slice_rows" option, 2 ... 8, default 2: W rows; at most 4096 --
 * the next 2 ... 8 trial points of every active walker first, the rest prior-rejected filler).  The loop itself runs
 * on the GPU as well: one single-workgroup kernel between two batches consumes the results and decides what the next
 * batch is (more candidates, the other half-ensemble, the next iteration with its chain row and mu tuning, or nothing
 * more), so the host only enqueues (batch, round kernel) pairs, a few ahead of the device, and synchronises once per
 * segment of iterations (the whole call unless the chain is larger than 256 MB).  While the call runs the calling
 * thread polls a word of mapped host memory.
 *   pos, lnprob, have_lnprob, seed, step0, chain, chain_lnprob: as in vp_stretch_run.  W even, 4 <= W <= 4096.
 *   mu         in: initial scale (zeus: 1.0), out: scale after the run.
 *   tune       != 0: adapt mu after every iteration (mu *= 2 n_expansions / (n_expansions + n_contractions)),
 *              and stop adapting after `patience` consecutive iterations with |ratio - 1| < tolerance
 *              (zeus: tolerance 0.05, patience 5).  The value carries the tuning state from call to call, so that
 *              run(n1) then run(n2) equals run(n1 + n2) while tuning is on: in, 1 + the number of consecutive
 *              in-tolerance iterations so far (1 for a fresh run); out, the same, or 0 once tuning has stopped.
 *   maxsteps   cap on the stepping-out expansions per slice (zeus: 10000).
 *   mu_history (nsteps) host or NULL: mu after each iteration.
 *   n_evals    out (or NULL): lnprob evaluations spent (trial points), ADDED to the value passed in.
 * Returns VP_ENAN if a trial point's lnprob (or the start state's) is NaN or the start state's is not finite. */
int vp_slice_run(vp_ctx* ctx, int W, int D, double* pos, double* lnprob, int have_lnprob, int nsteps, double* mu,
                 int* tune, double tolerance, int patience, int maxsteps, uint64_t seed, uint64_t step0,
                 double* chain, double* chain_lnprob, double* mu_history, int64_t* n_evals);

/* Philox4x32-10 block function used by vp_stretch_run / vp_slice_run (host evaluation; known-answer tests). */
void vp_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

/* Optional per-kernel timing with HIP events recorded on the stream the kernels are launched on
 * (used by bench.py for the roofline figure).  While enabled, every lnprob batch records events
 * around its prep / tile / finalize launches; vp_profile_read waits for them, returns the summed
 * milliseconds per kernel kind and the number of tile-kernel launches, and clears the record. */
int vp_profile_enable(vp_ctx* ctx, int enable);
int vp_profile_read(vp_ctx* ctx, double* prep_ms, double* tile_ms, double* finalize_ms,
                    int* n_tile_launches);

/* Tuning / experiment knobs of ONE context (none changes a result beyond the last bits: the grouping of the chi^2 sum,
 * and with "farfield" the far wings of lines that are many block widths away, each to 1e-16 in optical depth).
 * Defaults come from the RBVFIT_AMD_<NAME> environment variables, read once in vp_ctx_create -- never on
 * the per-call path.  Names: "geom" (-1 by batch size, 0 two-pass tiles, 1 one-pass tiles), "finalize"
 * (-1 auto, 0 own launch, 1 ticket), "walker" (-1 by batch size, 0 never, 1 whenever possible: the whole
 * batch as ONE launch, workgroup = walker), "prep_rpw", "zerocopy_max", "no_zerocopy", "no_fused_accept",
 * "slice_rows", "slice_seg", "tile_lpt", "walker_clusters", "prearm", "prearm_us" (vp_lnprob_batch), "flux_walker" (vp_model_flux_batch*: 1 / 0); "span", "waves", "no_multipole", "multipole_min", "lds_pad" and "farfield" (-1: per-block far-field
 * expansions for instruments with >= 8 lines in batches large enough to pay for the extra launch, 0 never, 1 whenever possible) apply to instruments added afterwards ("farfield" also at run time: 0 leaves the tables of an instrument unused, 1 uses them for every batch).
 * Unknown name -> VP_EINVAL.  (RBVFIT_AMD_VERBOSE in the environment makes vp_add_instrument print its far-field coverage
 * estimate to stderr; RBVFIT_AMD_LIB selects another build of the library in the Python loader.) */
int vp_set_option(vp_ctx* ctx, const char* name, long value);

/* ---- several GPUs from one process (the torch-free, RCCL-free form of the walker sharding) ----
 * A vp_multi owns one vp_ctx per listed device (a device may be listed more than once) with identical static
 * data.  vp_multi_lnprob_batch cuts the (W, D) batch into contiguous blocks of ceil(W / n_devices) rows -- the
 * partition of rbvfit_amd.dist.shard_bounds; trailing devices get fewer rows or none (ragged zeus batches with
 * W < n_devices) --, enqueues every block on its device before waiting for any, and each block's lnprob is
 * copied straight into its slice of `out`: no collective, the gather is the D2H copies.  Replaces the reference's
 * fork Pool fan-out (vfit_mcmc.py:35-49, 408-440) for callers that bind the C ABI without torch.  The per-device
 * contexts are reachable through vp_multi_ctx (e.g. for vp_set_option); errors name the failing device slot. */
typedef struct vp_multi vp_multi;
int vp_multi_create(vp_multi** out, int n_devices, const int* device_ids);
int vp_multi_destroy(vp_multi* m);
int vp_multi_n_devices(const vp_multi* m);
vp_ctx* vp_multi_ctx(vp_multi* m, int i);
int vp_multi_set_bounds(vp_multi* m, int D, const double* lb, const double* ub);
int vp_multi_add_instrument(vp_multi* m, int P, const double* wave, const double* flux, const double* inv_sigma2,
                            const double* log_inv_sigma2, int L, const double* lambda0, const double* gamma,
                            const double* f, const double* zfac, const int32_t* N_idx, const int32_t* b_idx,
                            const int32_t* v_idx, int K, const double* taps, int lsf_mode, int voigt_method,
                            int* inst_index);
int vp_multi_lnprob_batch(vp_multi* m, int W, int D, const double* theta, double* out);

/* vp_stretch_run for ONE ensemble sharded over the contexts of `m` (BASELINE config 4; the reference fans one ensemble over
 * its workers, vfit_mcmc.py:425-440, 536-540): every context holds the whole ensemble, proposes / evaluates / accepts its
 * block of ceil(W/2 / G) rows of the active half and writes the rows it moved into every replica through peer-mapped
 * pointers ((D + 1) doubles per moved walker); half-steps are ordered by events between the contexts' streams, the host
 * does not wait until the call ends (or a chain chunk is copied out).  Arguments, results and the chain are those of
 * vp_stretch_run FOR ANY G, bit for bit: the draws are keyed by the walker's index in the whole ensemble and every block is
 * evaluated with the launch structure the whole half-ensemble would get on one context.  At most 8 contexts. */
int vp_multi_stretch_run(vp_multi* m, int W, int D, double* pos, double* lnprob, int have_lnprob, int nsteps, double a,
                         uint64_t seed, uint64_t step0, double* chain, double* chain_lnprob, int64_t* naccepted);

/* vp_slice_run for ONE ensemble on the contexts of `m`: every context keeps the whole sampler state and runs the same
 * (deterministic) control kernels on it; each round's lnprob batch of trial rows is cut into blocks of ceil(B / G) rows,
 * one per context, whose results are written into every replica (one double per row), with an event barrier per round.
 * Arguments and results are those of vp_slice_run for any G, bit for bit.  At most 8 contexts. */
int vp_multi_slice_run(vp_multi* m, int W, int D, double* pos, double* lnprob, int have_lnprob, int nsteps, double* mu,
                       int* tune, double tolerance, int patience, int maxsteps, uint64_t seed, uint64_t step0,
                       double* chain, double* chain_lnprob, double* mu_history, int64_t* n_evals);
const char* vp_multi_last_error(const vp_multi* m);

/* The context's own stream (a hipStream_t, created non-blocking): what hip_stream == NULL selects in the
 * *_device entry points.  Lets a host framework order its own work against it (e.g.
 * torch.cuda.ExternalStream(handle).wait_stream(...)).  Batches of DIFFERENT contexts enqueued on their own streams run side by
 * side on the GPU -- independent ensembles fill each other's launch entries: measured 28.8 M evals/s for two 512-walker C1
 * ensembles against 24.2 M for one, 27.6 M against 16.5 M at 256 (bench.py, "two_ensembles_side_by_side"); two streams taken from
 * torch's pool did not overlap on ROCm 7.2 / torch 2.10, so pass NULL (or this handle) where that matters. */
void* vp_ctx_stream(const vp_ctx* ctx);

/* Introspection */
int vp_num_instruments(const vp_ctx* ctx);
int vp_ndim(const vp_ctx* ctx);
int vp_instrument_pixels(const vp_ctx* ctx, int inst);
int vp_device_id(const vp_ctx* ctx);
/* Launch structure the last lnprob batch used: 0 = prep_lines_kernel + tile_kernel (+ finalize_kernel),
 * 1 = walker_kernel (the whole batch in one launch: one instrument, or up to four with identical line tables), 2 = as 0 with farfield_kernel between preparation and tiles
 * (far lines from per-block expansions). */
int vp_last_launch_kind(const vp_ctx* ctx);
/* Workgroups per walker of the last walker_kernel launch: 0 = one (the ordinary form), 2 / 4 / 8 = its split form, in which a
 * walker is several workgroups of one-pass tiles on several compute units (option "walker_split": 0 never -- the default --, N
 * always N, -1 eight for batches of <= 32 walkers, where it measures 8 % faster: the reference's default is 50 walkers,
 * vfit_mcmc.py:127-135).  A row's value is then that of the one-pass tile launches (option "geom" = 1) bit for bit, whatever the
 * number of groups; it can differ in the last bit from the ordinary form's (two-pass tile sums), which is why it is opt-in. */
int vp_last_walker_split(const vp_ctx* ctx);
/* What the far-field expansions of the last lnprob batch covered (first instrument that took any; test / diagnosis hook, it
 * synchronises and copies the masks back): *variant = 0 none, 1 farfield_kernel<6,false> (lines outside clusters and whole
 * clusters, |x| >= 30), 2 farfield_kernel<9,true> (narrow-pixel instruments: also the MEMBERS of clusters too near for their
 * multipole, line by line, |x| >= 14); *covered = (walker, block, line) triples taken from the expansions, *covered_members =
 * those of them whose line is a member of a multipole cluster (with variant 1 they entered with their whole cluster, with
 * variant 2 also one by one), *pairs = walkers x blocks x lines.  Any pointer may be NULL. */
int vp_last_farfield_info(vp_ctx* ctx, int* variant, int64_t* covered, int64_t* covered_members, int64_t* pairs);

/* Pre-armed launches of vp_lnprob_batch (option "prearm", see vp_lnprob_batch): how many calls were started through one
 * (*used), how many such launches gave up waiting (*expired: the caller took longer than "prearm_us") and how many were sent away
 * because another entry point -- or a batch of another shape -- came first (*cancelled).  Any pointer may be NULL. */
int vp_prearm_counts(vp_ctx* ctx, int64_t* used, int64_t* expired, int64_t* cancelled);

/* Text of the last error on this context (or of the last failed vp_ctx_create when ctx is NULL).
 * Valid until the next call on the same context/thread. */
const char* vp_last_error(const vp_ctx* ctx);

/* Library version string: "rbvfit_amd " RBVFIT_AMD_VERSION " (gfx950, hip)".  One number for the header, the library and the
 * Python package (rbvfit_amd.__version__); tests/test_cabi_symbols.py checks that they agree. */
#define RBVFIT_AMD_VERSION "0.4.0"
const char* vp_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RBVFIT_AMD_H */
