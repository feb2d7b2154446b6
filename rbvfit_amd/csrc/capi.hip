// C ABI of the rbvfit_amd engine (see include/rbvfit_amd.h).  Host side: context, device
// residency of the static data, workspace, launches.  No torch types, no Python.
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

namespace {

thread_local std::string g_create_error;
#ifdef VP_STAMPS
// diagnostic build: wall-clock stamps of the last vp_lnprob_batch call (ns since its entry): 0 theta staged, 1 launches enqueued,
// 2 first output row seen, 3 all rows seen / completion, 4 copied out
double g_host_stamps[8];
std::chrono::steady_clock::time_point g_host_t0;
#define VP_HSTAMP(k) g_host_stamps[k] = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - g_host_t0).count()
#else
#define VP_HSTAMP(k) do { } while (0)
#endif

// Tuning / experiment knobs.  Defaults come from RBVFIT_AMD_* environment variables read ONCE, when the
// context is created (never on the per-call path); vp_set_option changes them per context.
struct Tuning {
    int prep_rpw = 64;          // records per wave of prep_lines_kernel
    int geom = -1;              // tile geometry: -1 by batch size, 0 two-pass tiles, 1 one-pass tiles
    int finalize = -1;          // final reduction: -1 by batch size, 0 own launch, 1 ticket in the tile kernel
    int walker = -1;            // walker_kernel (one launch per batch): -1 by batch size, 0 never, 1 whenever possible
    long zerocopy_max = 1l << 20;    // bytes of theta up to which the host entry reads pinned host memory directly
    int no_zerocopy = 0;
    int no_multipole = 0;       // (read when an instrument is added)
    int multipole_min = 3;      // smallest cluster (components of one transition) that gets a multipole record
                                // (2 measured slower on C1: 35.6 vs 32.2 us per 512-walker pass -- the attempt costs an
                                // un-prefetched record fetch per cluster and pass, and the prep launch grows)
    int span = 0;               // evaluated pixels per tile, 0 = default (read when an instrument is added)
    int waves = 0;              // waves per tile workgroup, 0 = default (read when an instrument is added)
    long lds_pad = 0;           // extra LDS bytes per tile workgroup: occupancy experiments
    int no_fused_accept = 0;    // device sampler: separate accept / propose launches
    int farfield = -1;          // far-field expansions in the tile launches: -1 for instruments with >= 8 lines and batches
                                // with enough covered (walker, block, item) triples (enqueue_lnprob), 0 never, 1 whenever possible (the instrument's
                                // block tables are made when it is added: 0 at that time rules them out for good)
    int no_shared_prep = 0;     // instruments with the previous one's line tables prepare their records again anyway
    int walker_clusters = 0;    // walker_kernel on instruments with multipole clusters: 0 = walk the member lines one by one
                                // (no cluster records: the plain instance), 1 = cluster records formed in the workgroup
                                // (that instance spills to scratch)
    int multi_sync = 0;         // vp_multi_stretch_run: how the contexts' half-steps are ordered (read from context 0): 0 events between
                                // their streams (default), 1 flags polled inside the kernels -- no host call and no cross-queue wait per
                                // half-step; tested with the contexts on ONE device, where it measured no faster than events because the
                                // contexts' kernels share the CUs (C1, 2 contexts: 49.9 vs 39.5 us per half-step; C3 at 2048 walkers:
                                // 257 vs 272); across devices it uses system-scope fences and has never run
    int host_spin = 2;          // how the host-buffer entries learn that their batch is done: 0 hipStreamSynchronize; 1 spin on a host-mapped
                                // completion word the stream writes behind the batch (512-walker C1 call 42.7 -> ~39.5 us); 2 zero-copy batches:
                                // spin on the OUTPUT ROWS themselves -- they are pre-set to a NaN pattern no arithmetic produces and every
                                // row is written exactly once, by the launch that finishes it: no completion packet behind the kernel and no
                                // hipStreamWriteValue32 call in front of it (falls back to 1 when an input carries that pattern)
    int no_ff_members = 0;      // (read when an instrument is added) never take cluster members into the far-field expansions one by one
    int tile_multi = -1;        // tiles of several instruments in one launch (tile_kernel_multi): -1 by batch size, 0 never, 1 whenever possible
    int gather_plain = 0;       // vp_gather_create: ordinary device memory for the gathered vector and flags instead of fine-grained
    int tile_lpt = 1;           // (read when an instrument is added) tile launches hand out the tiles with the most line cores
                                // first (0: in grid order)
    int slice_rows = 2;         // device slice sampler: rows of a round's lnprob batch per walker of the half-ensemble (2 ... 8)
    int flux_farfield = -1;     // vp_model_flux_batch[_device]: far lines from the blocks' expansions as in the lnprob launches: -1 by batch
                                // size (the lnprob rule), 0 never, 1 whenever the instrument has the tables
    int walker_prio = -1;       // walker_kernel: raised issue priority for the waves with line cores: -1 where workgroups share a CU, 0 never, 1 always
    long walker_perm_hex = 0;   // (experiments) an explicit WalkerArgs::wperm
    int walker_perm = -1;       // walker_kernel deals its tiles to the waves by estimated cost (WalkerArgs::wperm): -1 for batches of at most
                                // one workgroup per CU, 0 never (wave k takes tile k), 1 always
    int stretch_overlap = -1;   // vp_stretch_run, half-steps as one launch each: -1 consecutive half-steps on two streams, ordered walker by
                                // walker through version words (StretchArgs::ovl), where two half-ensembles fit the CUs at once; 0 never
                                // (every half-step behind the one before, one stream); 1 whenever the half-steps are one launch each
    int prearm = -1;            // vp_lnprob_batch, batches that are ONE walker_kernel launch: the launch for the NEXT call is put on the GPU
                                // while this call's runs, waits there for its theta (WalkerArgs::arm_*) and starts the moment the host has
                                // pushed it into the launch's slots (device memory, written through the PCIe BAR) -- no launch call, no
                                // command-processor latency, no read over PCIe and none of the kernel's theta-independent entry between the
                                // caller's theta and the arithmetic.  -1: when the previous call came within prearm_us / 2 of the
                                // one before returning (a sampler's loop), 0 never, 1 after every eligible call
    int prearm_us = 500;        // the LONGEST a pre-armed launch waits for its batch before it leaves (the GPU is held meanwhile); the wait it is
                                // actually given follows the caller's rhythm (vp_ctx::Prearm::gap_ema_us): 1.5 x the recent gap between calls + 10 us
    int flux_walker = 1;        // vp_model_flux_batch[_device]: batches the walker kernel would take as lnprob batches as ONE launch (0: prep + tile launches)
    int stretch_mailbox = 1;    // vp_stretch_run's overlapped half-steps (stretch_overlap) keep a walker's row, lnprob and version in one 64-byte
                                // line per buffer where D <= 6 (StretchArgs::ovl = 2); 0 = separate arrays
    int walker_split = 0;       // walker_kernel's split form (WalkerArgs::split: a walker = several workgroups of one-pass tiles): 0 never
                                // (default), N > 0 always N workgroups per walker where the form applies at all, -1 by batch size (eight
                                // where W x 8 workgroups leave every CU at most one: W <= 32 on MI355X).  Measured gain: 15.3 -> 14.0 us
                                // per pass for <= 32 walkers on C1, nothing above (walker_split_for); off by default because it makes a
                                // row's last bit depend on whether its batch has <= 32 rows (one-pass against two-pass tile sums) --
                                // a caller who recomputes a chain's lnprob in one large batch would no longer get the chain's bits
    int slice_seg = 0;          // device slice sampler: iterations per segment (the stretch the device works through without the
                                // host), 0 = as many as the chain chunk and the table of random splits allow (tests: small values)
};

struct Knob { const char* name; const char* env; int is_long; size_t off; };
#define VP_KNOB(field, envname, is_long) {#field, envname, is_long, offsetof(Tuning, field)}
const Knob g_knobs[] = {
    VP_KNOB(prep_rpw, "RBVFIT_AMD_PREP_RPW", 0), VP_KNOB(geom, "RBVFIT_AMD_GEOM", 0),
    VP_KNOB(finalize, "RBVFIT_AMD_FINALIZE", 0), VP_KNOB(walker, "RBVFIT_AMD_WALKER", 0),
    VP_KNOB(zerocopy_max, "RBVFIT_AMD_ZEROCOPY_MAX", 1),
    VP_KNOB(no_zerocopy, "RBVFIT_AMD_NO_ZEROCOPY", 0), VP_KNOB(no_multipole, "RBVFIT_AMD_NO_MULTIPOLE", 0), VP_KNOB(multipole_min, "RBVFIT_AMD_MULTIPOLE_MIN", 0),
    VP_KNOB(span, "RBVFIT_AMD_SPAN", 0), VP_KNOB(waves, "RBVFIT_AMD_WAVES", 0), VP_KNOB(lds_pad, "RBVFIT_AMD_LDS_PAD", 1),
    VP_KNOB(no_fused_accept, "RBVFIT_AMD_NO_FUSED_ACCEPT", 0), VP_KNOB(slice_rows, "RBVFIT_AMD_SLICE_ROWS", 0), VP_KNOB(walker_clusters, "RBVFIT_AMD_WALKER_CLUSTERS", 0), VP_KNOB(no_shared_prep, "RBVFIT_AMD_NO_SHARED_PREP", 0),
    VP_KNOB(farfield, "RBVFIT_AMD_FARFIELD", 0), VP_KNOB(multi_sync, "RBVFIT_AMD_MULTI_SYNC", 0), VP_KNOB(tile_multi, "RBVFIT_AMD_TILE_MULTI", 0), VP_KNOB(no_ff_members, "RBVFIT_AMD_NO_FF_MEMBERS", 0), VP_KNOB(host_spin, "RBVFIT_AMD_HOST_SPIN", 0),
    VP_KNOB(tile_lpt, "RBVFIT_AMD_TILE_LPT", 0), VP_KNOB(gather_plain, "RBVFIT_AMD_GATHER_PLAIN", 0), VP_KNOB(slice_seg, "RBVFIT_AMD_SLICE_SEG", 0),
    VP_KNOB(flux_farfield, "RBVFIT_AMD_FLUX_FARFIELD", 0), VP_KNOB(stretch_overlap, "RBVFIT_AMD_STRETCH_OVERLAP", 0), VP_KNOB(stretch_mailbox, "RBVFIT_AMD_STRETCH_MAILBOX", 0), VP_KNOB(flux_walker, "RBVFIT_AMD_FLUX_WALKER", 0), VP_KNOB(walker_perm, "RBVFIT_AMD_WALKER_PERM", 0), VP_KNOB(walker_prio, "RBVFIT_AMD_WALKER_PRIO", 0), VP_KNOB(walker_perm_hex, "RBVFIT_AMD_WALKER_PERM_HEX", 1),
    VP_KNOB(prearm, "RBVFIT_AMD_PREARM", 0), VP_KNOB(prearm_us, "RBVFIT_AMD_PREARM_US", 0), VP_KNOB(walker_split, "RBVFIT_AMD_WALKER_SPLIT", 0),
};
void set_knob(Tuning& t, const Knob& k, long v) {
    char* base = reinterpret_cast<char*>(&t) + k.off;
    if (k.is_long) *reinterpret_cast<long*>(base) = v; else *reinterpret_cast<int*>(base) = (int)v;
}
Tuning tuning_from_env() {
    Tuning t;
    for (const Knob& k : g_knobs)
        if (const char* e = getenv(k.env))      // set but empty counts as 1; decimal ("010" is ten), hexadecimal for the explicit tile deal alone
            set_knob(t, k, *e ? (std::strcmp(k.name, "walker_perm_hex") == 0 ? (long)strtoull(e, nullptr, 16) : strtol(e, nullptr, 10)) : 1);
    return t;
}

struct Instrument {
    vp::InstDev dev{};
    vp::InstDev dev_s{};         // same instrument, one-pass tiles: used for small batches (lnprob only)
    vp::InstDev dev_w{};         // walker_kernel's geometry: single-wave tiles of 384 evaluated pixels whatever the LSF
                                 // length (= dev for K <= 33; longer LSFs: more halo per tile, but one launch)
    size_t lds_w = 0;            // LDS bytes of one such tile; 0: the LSF is too long for single-wave tiles
    size_t lds_s = 0;            // LDS bytes of one single-wave ONE-pass tile (dev_s where nwaves == 1): walker_kernel's split form; 0: none
    int* split_hint = nullptr;   // (dev_s.ntiles) the split form's "tile met line cores before" hints (InstDev::core_hint of that launch)
    unsigned long long wperm = 0xFEDCBA9876543210ull;   // walker_kernel (this instrument alone): tile of wave k in nibble k (WalkerArgs)

    vp::LinesDev lines{};
    double sum_logw = 0.0;
    std::vector<void*> allocs;   // device allocations owned by this instrument
    double* d_flux = nullptr;
    double* d_w = nullptr;
    size_t lds_bytes = 0;
    // host copies for the "can any in-bounds walker leave the fast domain?" analysis
    std::vector<double> h_lambda0, h_gamma;
    std::vector<int> h_bidx;
    bool needs_generic = true;
    bool nanfix = false;         // astropy-branch LSF and NaN samples in the wavelength grid: InstDev::rbot (tile launches' NANFIX instances)
    int nwaves = 1;              // waves per tile workgroup (1, 2 or 4)
    bool ff_on = false;          // far-field expansions (farfield_kernel + the FF instance of tile_kernel) for lnprob
    double ff_cover = 0.0;       // estimated share of (block, line) pairs the expansions cover (host, prior-box centre)
    int ff_items = 0;            // multipole clusters + lines outside clusters: what a pass of the tile kernel walks
    std::vector<double> h_lines; // lambda0 | gamma | f | zfac: with h_idx, what the line records of a walker depend on
    std::vector<int> h_idx;      // N_idx | b_idx | v_idx | method | multipole settings
    unsigned long long member_mask[2] = {0ull, 0ull};   // lines that are members of multipole clusters (L <= 128: what the
                                       // far-field masks are compared with, vp_last_farfield_info)
    bool same_lines_as_prev = false;   // this instrument's records ARE the previous instrument's (same line tables):
                                       // its record-preparation launch is skipped (C3: two instruments, one physics)
};

}  // namespace

struct vp_ctx {
    int device = 0;
    pid_t pid = 0;               // the process that created the context (a forked child must not touch its mappings: prearm_leave_all_at_exit)
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;        // vp_stretch_run's overlapped half-steps: the odd half-steps (made on first use)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int num_cus = 0;
    mutable std::mutex mu;
    mutable std::string err;
    Tuning tune;
    size_t lds_limit = 65536;    // dynamic LDS a workgroup may ask for on this device
    int D = 0;
    double* d_lb = nullptr;
    double* d_ub = nullptr;
    std::vector<Instrument> inst;
    // per-batch workspace (grown on demand)
    int capW = 0;
    int capL = 0;          // max records per walker (lines + multipole clusters) the workspace was sized for
    int cap_tiles = 0;
    double* d_theta = nullptr;   // (capW, D)
    double* d_out = nullptr;     // (capW)
    double* d_lc = nullptr;      // (capW, capL, LC_STRIDE)
    double* d_partial = nullptr; // (capW, total_tiles)
    int* d_flags = nullptr;      // (capW)
    unsigned int* d_ticket = nullptr;  // (capW) arrival counters of the fused final reduction
    int* d_genflag = nullptr;    // 2 x (capW) walkers with lines outside the fast domain (per instrument pass): two buffers in turn -- the
                                 // record-preparation launch that fills one clears the other for the launch after it (genflag_acquire)
    unsigned gen_seq = 0;
    int gen_clean[2] = {0, 0};   // leading entries of each buffer known to be zero
    int* h_gen_any = nullptr;    // mapped host memory, one word: has any launch of this context (with these bounds and instruments) flagged a
    int* h_gen_any_dev = nullptr;   // walker?  (what the generic launches are sized by: tile_generic_kernel's grid)
    double* d_ff = nullptr;      // (capW x cap_ffblk, FF_STRIDE) far-field expansions, instrument after instrument
    int cap_ffblk = 0;
    std::vector<double> h_lb;    // host copy of the lower bounds
    std::vector<double> h_ub;    // ... and of the upper bounds
    int* d_tile_off = nullptr;   // 2 x (n_inst + 1): tile offsets of the full-size and of the one-pass geometry
    double* d_sum_logw = nullptr;
    bool meta_dirty = true;
    int total_tiles = 0;         // max over the two geometries (workspace size)
    int total_tiles_g[2] = {0, 0};
    int total_tiles_w = 0;       // tiles of all instruments in the single-wave geometry (tile_kernel_multi)
    double* h_pinned = nullptr;  // staging for theta / out
    double* h_pinned_dev = nullptr;   // ... as the device sees it (zero-copy batches)
    size_t h_pinned_bytes = 0;
    // completion of a host-buffer call without an interrupt: the stream writes a sequence number to a host-mapped word
    // behind the batch (hipStreamWriteValue32), the calling thread spins on it (host_wait)
    uint32_t* h_done = nullptr;
    uint32_t done_seq = 0;
    bool done_armed = false;     // the last batch enqueued carries a completion write
    // host_spin = 2: the last batch's output rows are what the host polls (lnprob_host_begin / host_wait)
    bool sentinel_armed = false;
    int sentinel_W = 0;
    const double* sentinel_out = nullptr;
    // pre-armed launch of the next vp_lnprob_batch call (Tuning::prearm)
    struct Prearm {
        uint32_t* h = nullptr;        // pinned host block the LAUNCH writes: h[16] "expired", h[32] "stuck"
        uint32_t* h_dev = nullptr;    // ... as the device sees it
        int bar = -1;                 // can the CPU write device memory (large BAR)?  -1 not asked yet
        double* slots = nullptr;      // fine-grained device memory the CPU writes through the BAR: slot_doubles per row (WalkerArgs::arm_slots)
        int slot_doubles = 0, slot_rows = 0;
        const double* theta_src = nullptr;    // the caller's theta of the batch in flight (only the slots have it so far)
        bool live = false;            // a launch is waiting on the stream (or has expired there)
        bool dirty = false;           // a pre-armed launch has been put on live_stream since the last fence: it writes rehearsal records into the
                                      // shared workspace even when it is sent away, so work on ANOTHER stream must be ordered behind it (foreign_stream_fence)
        hipEvent_t ev = nullptr;
        uint32_t seq = 0;             // ... with this sequence number
        int W = 0;                    // ... for this many rows
        hipStream_t cur = nullptr;    // the stream of the launch that serves the current call
        hipStream_t live_stream = nullptr;    // the stream the waiting launch is on
        bool inflight = false;        // the batch host_wait is waiting for was started through a pre-armed launch (seq_inflight)
        uint32_t seq_inflight = 0;
        bool have_last = false;       // last_return is set
        int last_W = -1;              // rows of the previous vp_lnprob_batch call
        std::chrono::steady_clock::time_point last_return;
        int misses = 0;               // consecutive launches that expired although their batch had been pushed
        // being a good neighbour: a waiting launch holds every CU, and only this library's own entry points can send it away --
        // anything else the process (or another process) puts on the GPU waits it out.  So it waits no longer than the caller's
        // own rhythm suggests, and a caller whose launches expire unused is left alone for a while.
        double gap_ema_us = 0.0;      // running mean of the time between a call's return and the next call (calls that came within prearm_us)
        int budget_us = 0;            // how long the launch now waiting was told to wait
        int cooldown = 0;             // calls still to pass before the next launch is pre-armed (set when one expired unused)
        int cooldown_next = 8;        // ... and what the next unused expiry will set it to (doubles, back to 8 after 16 used in a row)
        int used_streak = 0;
        int64_t used = 0, expired = 0, cancelled = 0;     // (vp_prearm_counts)
    } arm;
    bool sentinel_unsafe = false;    // some static input (bounds, spectra, line tables, taps) carries the sentinel's NaN payload
    // direct-write gather of the lnprob vector between the ranks of a multi-process job (vp_gather_*)
    struct Gather {
        int W = 0, world = 0, rank = 0, seq = 0;
        bool connected = false, finegrained = false;
        bool local_peers = false;              // the peers' vectors are plain device pointers of contexts of this process (no IPC mapping to close)
        bool shared_device = false;            // some ranks share a GPU: the handshake is a launch of its own (replicas_handshake)
        double* buf = nullptr;                 // (world, W) this rank's gathered vector
        int* flags = nullptr;                  // (MAX_REPLICAS) flags[r] = the last pass whose block from rank r has landed here
        unsigned int* done = nullptr;          // workgroup counters of the running launch (PUB_GROUPS + 1) | timeout flag (int)
        double* peer_buf[vp::MAX_REPLICAS] = {};
        int* peer_flags[vp::MAX_REPLICAS] = {};
    } gather;
    const vp::Replicas* gather_rep = nullptr;   // set around an enqueue_lnprob whose results go into the ranks' gathered vectors
    // model_flux / voigt_h scratch
    double* d_scratch = nullptr;
    size_t scratch_bytes = 0;
    // optional per-kernel timing with HIP events on the launch stream (bench.py roofline leg)
    int policy_W = 0;            // > 0: the launch structure of a batch is chosen as for THIS many rows (a block of a larger batch
                                 // that other contexts share: same structure, hence the same bits, as the whole batch on one context)
    int last_kind = 0;           // launch structure of the last lnprob batch: 0 prep + tile (+ finalize), 1 walker_kernel
    int last_split = 0;          // ... and, for walker_kernel, the workgroups per walker of its split form (0: the ordinary form)
    // the far-field expansions the last lnprob batch made for its FIRST instrument that took any (vp_last_farfield_info):
    // where they lie in the workspace, rows, blocks per row, which farfield_kernel instance, which instrument
    struct LastFF { const double* ff = nullptr; int W = 0, nbk = 0, members = 0, inst = -1; } last_ff;
    bool profiling = false;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    struct Span { size_t a, b; int kind; };   // kind: 0 prep, 1 tile, 2 finalize
    std::vector<Span> spans;
};

struct vp_multi {
    std::vector<vp_ctx*> ctx;
    std::mutex mu;
    std::string err;
    bool no_peer = false;        // some pair of devices cannot map each other's memory: no sharded device-resident sampler
    std::vector<hipEvent_t> ev;  // one per context: the half-step barrier of vp_multi_stretch_run
    bool broken = false;         // a setup call failed half-way and could not be undone: the contexts differ, every later call fails
};

namespace {

int fail(const vp_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg; else g_create_error = msg;
    return code;
}

#define HIP_TRY(c, expr)                                                                     \
    do {                                                                                     \
        hipError_t e__ = (expr);                                                             \
        if (e__ != hipSuccess)                                                               \
            return fail((c), VP_EHIP, std::string(#expr) + ": " + hipGetErrorString(e__));   \
    } while (0)

template <class T>
int upload(vp_ctx* c, Instrument* ins, const T* host, size_t n, T** out) {
    T* d = nullptr;
    HIP_TRY(c, hipMalloc((void**)&d, (n ? n : 1) * sizeof(T)));
    if (ins) ins->allocs.push_back(d);
    if (n) HIP_TRY(c, hipMemcpy(d, host, n * sizeof(T), hipMemcpyHostToDevice));
    *out = d;
    return VP_OK;
}

double neumaier_sum(const double* v, int n) {
    double s = 0.0, comp = 0.0;
    for (int i = 0; i < n; ++i) {
        double t = s + v[i];
        if (std::fabs(s) >= std::fabs(v[i])) comp += (s - t) + v[i]; else comp += (v[i] - t) + s;
        s = t;
    }
    return s + comp;
}

// The NaN the output rows of a zero-copy host batch are pre-set to (host_spin = 2).  Hardware-made NaNs are the default quiet NaN
// and NaN inputs travel through arithmetic with their payload (quieted, sign possibly flipped): an output can only equal
// the pattern if an INPUT carried this payload, and every entry point that takes doubles looks for it.
constexpr uint64_t VP_SENTINEL_BITS = 0x7FF8A5C3965A3C69ull;
inline bool carries_sentinel(const double* v, size_t n) {
    bool hit = false;
    for (size_t i = 0; i < n; ++i) {
        uint64_t b;
        std::memcpy(&b, v + i, 8);
        hit |= ((b & 0x7FFFFFFFFFFFFFFFull) | 0x0008000000000000ull) == VP_SENTINEL_BITS;
    }
    return hit;
}

int ensure_scratch(vp_ctx* c, size_t bytes) {
    if (bytes <= c->scratch_bytes) return VP_OK;
    if (c->d_scratch) HIP_TRY(c, hipFree(c->d_scratch));
    c->d_scratch = nullptr; c->scratch_bytes = 0;
    HIP_TRY(c, hipMalloc((void**)&c->d_scratch, bytes));
    c->scratch_bytes = bytes;
    return VP_OK;
}

int ensure_pinned(vp_ctx* c, size_t bytes) {
    if (bytes <= c->h_pinned_bytes) return VP_OK;
    if (c->h_pinned) HIP_TRY(c, hipHostFree(c->h_pinned));
    c->h_pinned = nullptr; c->h_pinned_bytes = 0; c->h_pinned_dev = nullptr;
    HIP_TRY(c, hipHostMalloc((void**)&c->h_pinned, bytes, hipHostMallocDefault));
    c->h_pinned_bytes = bytes;
    return VP_OK;
}

int ensure_workspace(vp_ctx* c, int W) {
    int maxL = 1;
    for (auto& in : c->inst) maxL = std::max(maxL, in.dev.L + in.dev.NCm);
    if (c->meta_dirty) {
        const size_t n1 = c->inst.size() + 1;
        std::vector<int> off(3 * n1, 0);                   // full-size | one-pass | single-wave (dev_w) geometry
        std::vector<double> slw(n1, 0.0);
        for (size_t k = 0; k < c->inst.size(); ++k) {
            off[k + 1] = off[k] + c->inst[k].dev.ntiles;
            off[n1 + k + 1] = off[n1 + k] + c->inst[k].dev_s.ntiles;
            off[2 * n1 + k + 1] = off[2 * n1 + k] + c->inst[k].dev_w.ntiles;
            slw[k] = c->inst[k].sum_logw;
        }
        c->total_tiles_g[0] = off[n1 - 1];
        c->total_tiles_g[1] = off[2 * n1 - 1];
        c->total_tiles_w = off[3 * n1 - 1];
        c->total_tiles = std::max({c->total_tiles_g[0], c->total_tiles_g[1], c->total_tiles_w});
        if (c->d_tile_off) HIP_TRY(c, hipFree(c->d_tile_off));
        if (c->d_sum_logw) HIP_TRY(c, hipFree(c->d_sum_logw));
        HIP_TRY(c, hipMalloc((void**)&c->d_tile_off, off.size() * sizeof(int)));
        HIP_TRY(c, hipMalloc((void**)&c->d_sum_logw, slw.size() * sizeof(double)));
        HIP_TRY(c, hipMemcpy(c->d_tile_off, off.data(), off.size() * sizeof(int), hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(c->d_sum_logw, slw.data(), slw.size() * sizeof(double), hipMemcpyHostToDevice));
        c->meta_dirty = false;
    }
    int ffblk = 0;
    for (auto& in : c->inst)
        if (in.ff_on) ffblk += std::max(in.dev.ntiles * in.dev.ff_nblk, in.dev_s.ntiles * in.dev_s.ff_nblk);   // (every instrument its own stretch)
    if (W <= c->capW && maxL <= c->capL && c->total_tiles <= c->cap_tiles && ffblk <= c->cap_ffblk) return VP_OK;
    // a stream-ordered previous call may still be using the old buffers
    HIP_TRY(c, hipDeviceSynchronize());
    const int newW = std::max(W, c->capW);
    for (void* p : {(void*)c->d_theta, (void*)c->d_out, (void*)c->d_lc, (void*)c->d_partial, (void*)c->d_flags, (void*)c->d_ticket, (void*)c->d_genflag,
                    (void*)c->d_ff})
        if (p) HIP_TRY(c, hipFree(p));
    c->d_theta = c->d_out = c->d_lc = c->d_partial = nullptr; c->d_flags = nullptr; c->d_ticket = nullptr; c->d_genflag = nullptr; c->capW = 0;
    c->d_ff = nullptr; c->cap_ffblk = 0;
    if (ffblk > 0) HIP_TRY(c, hipMalloc((void**)&c->d_ff, (size_t)newW * ffblk * vp::FF_STRIDE * sizeof(double)));
    HIP_TRY(c, hipMalloc((void**)&c->d_theta, (size_t)newW * std::max(c->D, 1) * sizeof(double)));
    HIP_TRY(c, hipMalloc((void**)&c->d_out, (size_t)newW * sizeof(double)));
    // (at least 1024 rows of records: walker_kernel's split form gives every workgroup of a walker its own rows -- W x split <= 512 --
    //  and the stretch sampler keeps two half-steps in flight)
    HIP_TRY(c, hipMalloc((void**)&c->d_lc, (size_t)std::max(newW, 1024) * maxL * vp::LC_STRIDE * sizeof(double)));
    HIP_TRY(c, hipMalloc((void**)&c->d_partial, (size_t)newW * std::max(c->total_tiles, 1) * sizeof(double)));
    HIP_TRY(c, hipMalloc((void**)&c->d_flags, (size_t)newW * sizeof(int)));
    HIP_TRY(c, hipMemset(c->d_flags, 0, (size_t)newW * sizeof(int)));
    HIP_TRY(c, hipMalloc((void**)&c->d_ticket, (size_t)newW * sizeof(unsigned int)));
    HIP_TRY(c, hipMemset(c->d_ticket, 0, (size_t)newW * sizeof(unsigned int)));
    HIP_TRY(c, hipMalloc((void**)&c->d_genflag, 2 * (size_t)newW * sizeof(int)));
    HIP_TRY(c, hipMemset(c->d_genflag, 0, 2 * (size_t)newW * sizeof(int)));
    c->gen_clean[0] = c->gen_clean[1] = newW;
    c->capW = newW; c->capL = maxL; c->cap_tiles = c->total_tiles; c->cap_ffblk = ffblk;
    return VP_OK;
}

template <int OUT, bool GENERIC>
void launch_tile(const Instrument& in, const double* lc, const int* flags, double* out, int stride, int offset,
                 int W, hipStream_t s, const vp::FinalizeArgs& fin, const int* genflag, const vp::InstDev* geom = nullptr,
                 int grid_z = 1, double* ff = nullptr, int gen_slots = 0, const double* theta_rebuild = nullptr, int D = 0) {
    const vp::InstDev& dev = geom ? *geom : in.dev;
    dim3 grid(W, dev.ntiles, grid_z);
    dim3 block(64 * in.nwaves);
    if (dev.rbot && OUT != 2) {           // NaN wavelength samples on the astropy branch: the instances that renormalise (InstDev::rbot)
        constexpr int O = OUT == 1 ? 1 : 0;
        if (GENERIC) {
            const dim3 gg(gen_slots > 0 ? std::min(W, gen_slots) : W, dev.ntiles, grid_z);
            hipLaunchKernelGGL((vp::tile_generic_kernel<O, false, true>), gg, block, in.lds_bytes, s, dev, lc, flags, out, stride, offset, fin, genflag, W,
                               in.lines, (const double*)nullptr, 0);
        } else if (dev.method == VP_VOIGT_FAST) {
            hipLaunchKernelGGL((vp::tile_kernel<1, O, false, false, true>), grid, block, in.lds_bytes, s, dev, lc, flags, out, stride, offset, fin, genflag);
        } else {
            hipLaunchKernelGGL((vp::tile_kernel<0, O, false, false, true>), grid, block, in.lds_bytes, s, dev, lc, flags, out, stride, offset, fin, genflag);
        }
        return;
    }
#ifndef VP_NO_TILE1
    if (OUT == 0 && !GENERIC && in.nwaves == 1 && grid_z == 1) {        // single-wave tiles: the kernel compiled for them alone
        vp::Tile1Args a{dev, lc, flags, genflag, vp::TileTail{out, stride, offset, fin}};
        a.I.ff = ff;
        if (dev.method == VP_VOIGT_FAST) hipLaunchKernelGGL((vp::tile_kernel1<1, false>), grid, block, in.lds_bytes, s, a);
        else if (ff) hipLaunchKernelGGL((vp::tile_kernel1<0, true>), grid, block, in.lds_bytes, s, a);
        else hipLaunchKernelGGL((vp::tile_kernel1<0, false>), grid, block, in.lds_bytes, s, a);
        return;
    }
#endif
    if (ff && OUT != 2 && !GENERIC && dev.method == VP_VOIGT_WOFZ) {     // far lines from the blocks' expansions
        vp::InstDev d2 = dev;
        d2.ff = ff;
        hipLaunchKernelGGL((vp::tile_kernel<0, OUT == 1 ? 1 : 0, false, true>), grid, block, in.lds_bytes, s, d2, lc, flags, out, stride, offset, fin, genflag);
        return;
    }
    if (dev.method == VP_VOIGT_FAST) {
        if (!GENERIC)
            hipLaunchKernelGGL((vp::tile_kernel<1, OUT, false>), grid, block, in.lds_bytes, s, dev, lc, flags, out,
                               stride, offset, fin, genflag);
    } else if (GENERIC) {
        // (almost always an empty launch: a small grid whose workgroups walk the flagged walkers -- tile_generic_kernel)
        //  where the batch before flagged none; one workgroup per walker otherwise, as a fit with damped lines needs them)
        const dim3 gg(gen_slots > 0 ? std::min(W, gen_slots) : W, dev.ntiles, grid_z);
        if (OUT == 1 && theta_rebuild)
            hipLaunchKernelGGL((vp::tile_generic_kernel<1, true>), gg, block, in.lds_bytes, s, dev, lc, flags, out, stride, offset, fin, genflag, W,
                               in.lines, theta_rebuild, D);
        else
            hipLaunchKernelGGL((vp::tile_generic_kernel<OUT, false>), gg, block, in.lds_bytes, s, dev, lc, flags, out, stride, offset, fin, genflag, W,
                               in.lines, (const double*)nullptr, 0);
    } else {
        hipLaunchKernelGGL((vp::tile_kernel<0, OUT, false>), grid, block, in.lds_bytes, s, dev, lc, flags, out,
                           stride, offset, fin, genflag);
    }
}

// Can a walker inside the prior box have a line outside the fast domain (a > 0.1 or b <= 0)?
// a = gamma lambda0 / (4 pi 1e13 b) is largest at the lower bound of b.  A 10 % margin covers the
// difference between this estimate and the device arithmetic.
void analyse_generic(vp_ctx* c, Instrument& in) {
    in.needs_generic = false;
    for (size_t l = 0; l < in.h_lambda0.size(); ++l) {
        const double blo = c->h_lb.empty() ? 0.0 : c->h_lb[in.h_bidx[l]];
        if (!(blo > 0.0)) { in.needs_generic = true; return; }
        const double amax = std::fabs(in.h_gamma[l]) * in.h_lambda0[l] / (12.566370614359172 * 1e13 * blo);
        if (!(amax < 0.09) || in.h_gamma[l] < 0.0) { in.needs_generic = true; return; }
    }
}

size_t prof_mark(vp_ctx* c, hipStream_t s) {
    if (c->ev_used == c->ev_pool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return (size_t)-1;
        c->ev_pool.push_back(e);
    }
    const size_t i = c->ev_used++;
    (void)hipEventRecord(c->ev_pool[i], s);
    return i;
}

// enqueue the whole lnprob pipeline for device-resident theta / out: per instrument a prep launch
// (line records; the first one also applies the box prior and writes -inf rows) and a tile launch;
// the last-arriving tile workgroup of each walker performs the final reduction.
// Record preparation launch: one lane per record, 64 records per wave (fewer per wave measured no
// faster even at 512 walkers x 4 lines; the prep_rpw knob overrides for experiments).
// The generic-path flags of a record-preparation launch: the buffer whose turn it is, zero in its first W entries (a memset only
// where the launch before could not vouch for that), and the other buffer, which this launch clears for the next one.
struct GenFlags { int* use; int* clear; int* any; int* any_clear; int slots; };
static int genflag_acquire(vp_ctx* c, int W, hipStream_t s, GenFlags* out) {
    const int b = (int)(c->gen_seq & 1u), o = b ^ 1;
    if (!c->h_gen_any) {
        if (hipHostMalloc((void**)&c->h_gen_any, 64, hipHostMallocMapped) == hipSuccess &&
            hipHostGetDevicePointer((void**)&c->h_gen_any_dev, c->h_gen_any, 0) == hipSuccess) {
            c->h_gen_any[0] = 0;
        } else { c->h_gen_any = nullptr; c->h_gen_any_dev = nullptr; (void)hipGetLastError(); }
    }
    // No batch of this context has flagged a walker so far (ONE word, set by the launches, never cleared by them: a fit with
    // damped lines says so within its first batches and keeps one workgroup per walker from then on; batches enqueued ahead of
    // the GPU would make anything finer unreliable): this batch's generic launch will most likely be empty -- a small grid
    const bool quiet = c->h_gen_any && __atomic_load_n(&c->h_gen_any[0], __ATOMIC_RELAXED) == 0;
    int* use = c->d_genflag + (size_t)b * c->capW;
    if (c->gen_clean[b] < W) HIP_TRY(c, hipMemsetAsync(use, 0, (size_t)W * sizeof(int), s));
    c->gen_clean[b] = 0;                                   // (written by this launch)
    if (c->gen_clean[o] < W) c->gen_clean[o] = W;          // (cleared by this launch)
    ++c->gen_seq;
    *out = GenFlags{use, c->d_genflag + (size_t)o * c->capW, c->h_gen_any_dev, nullptr, quiet ? vp::GEN_SLOTS : 0};
    return VP_OK;
}
static void launch_prep(const vp_ctx* c, const Instrument& in, const double* d_theta, int W, int do_flags, double* d_out,
                        int* genflag, hipStream_t s, int* genflag_clear = nullptr, int* gen_any = nullptr, int* gen_any_clear = nullptr) {
    const long nline = (long)W * in.dev.L, ncl = (long)W * in.dev.NCm;
    const int rpw = std::max(1, std::min(64, c->tune.prep_rpw));
    vp::PrepGrid g;
    g.rpw = rpw;
    g.nb_line = (int)((nline + rpw - 1) / rpw);
    g.nb_cl = (int)((ncl + rpw - 1) / rpw);
    g.cl_wpw = 0;
    if (in.lines.M > 0 && in.lines.M <= 64 && c->tune.prep_rpw >= 64) {      // cluster records by a lane per member
        g.cl_wpw = 64 / in.lines.M;
        g.nb_cl = (W + g.cl_wpw - 1) / g.cl_wpw;
    }
    g.nb_flag = do_flags ? W : 0;
    // (direct-write gather: the pass's first launch -- the one that applies the box prior -- handshakes with the peers)
    const vp::Replicas rep = (c->gather_rep && do_flags) ? *c->gather_rep : vp::Replicas{};
    hipLaunchKernelGGL(vp::prep_lines_kernel, dim3(g.nb_line + g.nb_cl + g.nb_flag), dim3(64), 0, s, d_theta, W, c->D,
                       in.lines, c->d_lb, c->d_ub, c->d_lc, c->d_flags, do_flags, d_out, genflag, g, rep, genflag_clear, gen_any, gen_any_clear);
}

// walker_kernel: the whole batch in ONE launch (workgroup = walker, wave = tile).  Possible for a single
// instrument with single-wave tiles, at most 16 of them, whose prior box keeps every line in the fast
// domain.  A walker's workgroup holds 12 wave slots of one CU for as long as its slowest tile runs, two fit on a
// CU, so the launch time is a step function of the number of 256-workgroup layers: measured on C1 (us per pass,
// walker kernel / prep + tile + finalize launches) 64: 18.2 / 22.3, 128: 18.3 / 23.5, 256: 18.3 / 24.9,
// 320: 24.8 / 26.9, 384: 24.8 / 29.3, 512: 25.0 / 31.0, 640: 38.5 / 35.5, 768: 38.9 / 36.4, 1024: 47.3 / 43.1.
// By default it is therefore used whenever the batch fits the CUs at once (two layers on C1); larger batches keep
// the one-wave workgroups, whose slots the hardware refills one by one as tiles finish.  Instruments with multipole
// clusters (>= 3 components of a transition) run it WITHOUT cluster records, their members as ordinary lines: forming
// the records in the workgroup costs a long per-lane chain and scratch; measured on 4096 pixels, us per pass, walker
// kernel / launches: MgII doublet x 3 components 256 walkers 22.5 / 32.1, 512: 29.3 / 39.2; x 4: 28.5 / 36.5 and
// 38.6 / 49.1; FeII 4 transitions x 4: 32.2 / 38.2 and 43.7 / 51.8.
// (one or two instruments: a second one must have the first one's line tables -- the walker's records serve both)
int walker_tiles(const vp_ctx* c) {
    int n = 0;
    for (auto& in : c->inst) n += in.dev_w.ntiles;
    return n;
}
size_t walker_wave_lds(const vp_ctx* c) {
    size_t b = 0;
    for (auto& in : c->inst) b = std::max(b, in.lds_w);
    return b;
}
size_t walker_lds_bytes(const vp_ctx* c) {           // tiles | tile sums, prior flag, spare | sampler form: 4 scalars, X_k, Y (64 each)
    return (size_t)walker_tiles(c) * walker_wave_lds(c) + (walker_tiles(c) + 2 + 4 + 128) * sizeof(double);
}

#ifndef VP_WALKER_MAX_LINES
#define VP_WALKER_MAX_LINES 40        // above: the launches win (multipoles, far-field expansions, finer scheduling) -- 5000 pixels,
#endif                                // 256 walkers, us per pass, walker kernel / launches: 20 lines 38.7 / 49.6, 40: 67.9 / 69.9, 64: 105.8 / 99.0
bool walker_applies(const vp_ctx* c, int W) {
    if (c->tune.walker == 0 || c->inst.empty() || c->inst.size() > 4 || c->D > 64) return false;
    const Instrument& in = c->inst[0];
    for (auto& k : c->inst) {
        if (k.nanfix) return false;          // (NaN wavelength samples on the astropy branch: the tile launches' NANFIX instances)
        if (k.lds_w == 0 || k.dev.method != in.dev.method) return false;
        if (k.dev.method == VP_VOIGT_WOFZ && k.needs_generic) return false;
    }
    for (size_t k = 1; k < c->inst.size(); ++k)
        if (!c->inst[k].same_lines_as_prev || c->tune.walker_clusters) return false;
    const int nt = walker_tiles(c);
    if (nt > vp::WALKER_THREADS_MAX / 64) return false;
    if (walker_lds_bytes(c) > c->lds_limit) return false;
    if (c->tune.walker == 1) return true;
    if (in.dev.L > VP_WALKER_MAX_LINES || (in.dev.NCm > 0 && c->tune.walker_clusters)) return false;
    // LSFs of more than 33 taps: the launches use 2- or 4-wave tile workgroups, the walker kernel single-wave tiles with
    // more halo -- measured on 2100 pixels, us per pass, launches / walker kernel: 45 taps 64 walkers 17.8 / 18.1,
    // 256: 20.5 / 18.2, 512: 26.9 / 22.0; 101 taps 19.4 / 22.1, 23.9 / 22.2, 35.0 / 32.1
    for (auto& k : c->inst)
        if (k.nwaves != 1 && W < 192) return false;
    // a CU holds per_cu walker workgroups at once (24 wave slots / waves per walker, LDS permitting); the batch lies
    // on the 256 CUs in layers of 256 workgroups and the launch takes as long as the fullest CU's layers
    const int per_cu = std::max(1, std::min(24 / std::max(1, nt), (int)(c->lds_limit / walker_lds_bytes(c))));
    // (short spectra: more workgroups fit on a CU and more layers stay ahead of the launches -- C1 cut to 1400 pixels, 4
    // tiles, 1024 walkers 22.0 vs 29.2 us; 2100 pixels, 6 tiles: 768 walkers 22.7 vs 30.1, 1024: 39.3 vs 33.5; 2800 pixels,
    // 8 tiles: 768 walkers 26.4 vs 33.6, 1024: 38.0 vs 37.6)
    const int layers = (W + 255) / 256;
    // (two workgroups per CU -- C1's 12 tiles --: a batch that fills a second round of 512 workgroups is still better off than
    //  through the launches -- C1, us per pass, walker kernel / launches: 768 walkers 35.5 / 36.1, 896: 41.7 / 42.5, 1024: 41.9 /
    //  44.75; 576: 35.2 / 29.9, 640: 35.3 / 32.1, 1280: 55.5 / 51.6)
    if (per_cu == 2 && layers == 4) return true;
    return layers <= per_cu && (layers <= 3 || layers <= per_cu - 2);
}

int walker_prio_for(vp_ctx* c, int W);
int ctx_num_cus(vp_ctx* c) {
    if (c->num_cus == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, c->device) == hipSuccess) c->num_cus = n;
    }
    return c->num_cus;
}
// walker_kernel's split form (WalkerArgs::split): how many workgroups a walker of a W-row batch gets, 0 = the ordinary form.
// Where a batch leaves CUs without a workgroup the launch is as long as its heaviest wave's chain (profiles/r05_C1_budget.txt:
// entry 5.0 + phase-A skeleton 3.0 + line cores 3.3 us of 15.5 at 256 walkers); one-pass tiles shorten the chain behind the entry,
// and several workgroups per walker put them on the idle CUs.
int walker_split_for(vp_ctx* c, int W) {
    if (c->tune.walker_split == 0 || c->inst.size() != 1 || c->tune.walker_clusters || c->gather_rep) return 0;
    const Instrument& in = c->inst[0];
    if (in.lds_s == 0 || !in.split_hint) return 0;
    const int nt = in.dev_s.ntiles, ncu = ctx_num_cus(c);
    int G = c->tune.walker_split;
    if (G < 0) {
        // Measured on C1 (us per pass, ordinary form / 2 / 4 / 8 groups; profiles/r05_experiments/split_sweep.txt): 16 walkers 15.25 /
        // 15.90 / 15.47 / 13.99, 32: 15.29 / 15.88 / 15.51 / 14.05, 50: 15.35 / 15.93 / 15.55 / 15.89, 64: 15.37 / 16.00 / 15.71 / 16.21,
        // 128: 15.42 / 16.06 / 18.52 / 19.77, 256: 15.42 / 27.99 (two 13-wave workgroups do not fit a CU's wave slots together) --
        // the form pays only where every group gets a CU's SIMDs to itself (one wave per SIMD): the launch is a 5.5 us entry plus
        // the heaviest wave's own latency chain (60 % of a lone wave's cycles are s_waitcnt), which one-pass tiles shorten by a
        // third, not by half.  So: eight groups where W x 8 workgroups still leave every CU at most one, nothing above.
        if (ncu <= 0 || (long)W * 8 > ncu) return 0;
        G = 8;
    }
    G = std::min(G, nt);
    if (G < 2 || (long)W * G > 1024) return 0;
    const int nwpg = (nt + G - 1) / G;
    if (nwpg > vp::WALKER_THREADS_MAX / 64) return 0;
    if ((size_t)nwpg * in.lds_s + (size_t)(nwpg + 2 + 4 + 128) * sizeof(double) > c->lds_limit) return 0;
    return G;
}
struct SplitShape { int G, waves; size_t lds; };
SplitShape split_shape(const vp_ctx* c, int G) {
    const Instrument& in = c->inst[0];
    const int nwpg = (in.dev_s.ntiles + G - 1) / G;
    return SplitShape{G, nwpg, (size_t)nwpg * in.lds_s + (size_t)(nwpg + 2 + 4 + 128) * sizeof(double)};
}

// (clusters: without their multipole records the members are ordinary lines -- a few more wing evaluations per pass
// against a cluster preparation chain inside every workgroup)
// split > 0: the split form -- W x split workgroups on the one-pass geometry; split_row0: first row of the tile-sum / ticket
// workspace this launch may use (the record rows come with a.lc)
template <bool SAMPLER, bool ARMED = false>
void launch_walker_any(vp_ctx* c, int W, vp::WalkerArgs a, const vp::StretchArgs& st, hipStream_t s, int split = 0, int split_row0 = 0) {
    const Instrument& in = c->inst[0];
    if (split > 0) {
        const SplitShape sh = split_shape(c, split);
        vp::InstDev d0 = in.dev_s;
        vp::LinesDev t0 = in.lines;
        d0.NCm = 0; t0.NCm = 0;
        d0.core_hint = in.split_hint;
        d0.ff = nullptr;
        a.wave_lds = (int)(in.lds_s / sizeof(double));
        a.split = split; a.split_row0 = split_row0; a.split_part = c->d_partial; a.split_ticket = c->d_ticket;
        a.wperm = 0xFEDCBA9876543210ull;
        if (!SAMPLER) a.prio = walker_prio_for(c, W * split);
        const dim3 grid(W * split), block(64 * sh.waves);
        if (in.dev.method == VP_VOIGT_FAST) hipLaunchKernelGGL((vp::walker_kernel<1, false, SAMPLER, ARMED, 0, true>), grid, block, sh.lds, s, d0, t0, a, st);
        else hipLaunchKernelGGL((vp::walker_kernel<0, false, SAMPLER, ARMED, 0, true>), grid, block, sh.lds, s, d0, t0, a, st);
        return;
    }
    const dim3 grid(W), block(64 * walker_tiles(c));
    const size_t lds = walker_lds_bytes(c);
    vp::InstDev d0 = in.dev_w;
    vp::LinesDev t0 = in.lines;
    const bool keep_clusters = in.dev.NCm > 0 && c->tune.walker_clusters && !SAMPLER && c->inst.size() == 1;
    if (!keep_clusters) { d0.NCm = 0; t0.NCm = 0; }
    if (c->inst.size() > 1) {
        vp::InstDev dk[4] = {d0, d0, d0, d0};
        vp::WalkerMore tb{};
        int tsum = d0.ntiles;
        for (size_t k = 1; k < c->inst.size(); ++k) {
            dk[k] = c->inst[k].dev_w;
            dk[k].NCm = 0;
            tb.t[k - 1] = tsum;
            tb.slw[k - 1] = c->inst[k].sum_logw;
            tsum += dk[k].ntiles;
        }
        for (size_t k = c->inst.size(); k < 4; ++k) tb.t[k - 1] = tsum;        // (no tiles)
        const bool fast = in.dev.method == VP_VOIGT_FAST;
        if (c->inst.size() == 2) {
            if (fast) hipLaunchKernelGGL((vp::walker_kernel2<1, SAMPLER, ARMED>), grid, block, lds, s, dk[0], dk[1], tb, t0, a, st);
            else hipLaunchKernelGGL((vp::walker_kernel2<0, SAMPLER, ARMED>), grid, block, lds, s, dk[0], dk[1], tb, t0, a, st);
        } else {
            if (fast) hipLaunchKernelGGL((vp::walker_kernel4<1, SAMPLER, ARMED>), grid, block, lds, s, dk[0], dk[1], dk[2], dk[3], tb, t0, a, st);
            else hipLaunchKernelGGL((vp::walker_kernel4<0, SAMPLER, ARMED>), grid, block, lds, s, dk[0], dk[1], dk[2], dk[3], tb, t0, a, st);
        }
        return;
    }
    if (in.dev.method == VP_VOIGT_FAST) hipLaunchKernelGGL((vp::walker_kernel<1, false, SAMPLER, ARMED>), grid, block, lds, s, d0, t0, a, st);
    else if (keep_clusters) hipLaunchKernelGGL((vp::walker_kernel<0, true, false, ARMED>), grid, block, lds, s, d0, t0, a, st);
    else hipLaunchKernelGGL((vp::walker_kernel<0, false, SAMPLER, ARMED>), grid, block, lds, s, d0, t0, a, st);
}

// which deal of tiles to waves a walker launch of W workgroups gets (Tuning::walker_perm)
unsigned long long walker_perm_for(vp_ctx* c, int W) {
    const unsigned long long ident = 0xFEDCBA9876543210ull;
    if (c->inst.size() != 1 || c->tune.walker_perm == 0) return ident;
    if (c->tune.walker_perm_hex != 0) return (unsigned long long)c->tune.walker_perm_hex;     // (experiments: an explicit deal)
    if (c->tune.walker_perm > 0) return c->inst[0].wperm;
    if (c->num_cus == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, c->device) == hipSuccess) c->num_cus = n;
    }
    return (c->num_cus > 0 && W <= c->num_cus) ? c->inst[0].wperm : ident;
}
// raised issue priority for the waves with line cores: where workgroups share a CU (WalkerArgs::prio)
int walker_prio_for(vp_ctx* c, int W) {
    if (c->tune.walker_prio >= 0) return c->tune.walker_prio ? 1 : 0;
    if (c->num_cus == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, c->device) == hipSuccess) c->num_cus = n;
    }
    return (c->num_cus > 0 && W <= c->num_cus) ? 0 : 1;
}

void launch_walker(vp_ctx* c, int W, const double* d_theta, double* d_out, hipStream_t s, const vp::Replicas* gather = nullptr,
                   bool armed = false) {
    vp::WalkerArgs a{d_theta, c->d_lb, c->d_ub, c->d_lc, d_out, c->inst[0].sum_logw, c->D, (int)(walker_wave_lds(c) / sizeof(double)),
                     walker_prio_for(c, W), walker_perm_for(c, W)};
    if (armed) {
        a.arm_slots = c->arm.slots;
        a.arm_slot_doubles = c->arm.slot_doubles;
        a.arm_host = c->arm.h_dev;
        a.arm_seq = c->arm.seq;
        // (100 MHz clock.  The caller's own rhythm: one and a half times its recent gap between calls + 10 us, at least 20 us, at most prearm_us --
        //  itself at most 100 ms, far below the other waves' own bound of ~1 s of polling, SYNC_SPIN_LIMIT.  "prearm" = 1 -- tests,
        //  experiments -- waits the whole prearm_us.)
        const int cap = std::min(std::max(1, c->tune.prearm_us), 100000);
        c->arm.budget_us = c->tune.prearm > 0 ? cap : std::min(cap, std::max(20, (int)(1.5 * c->arm.gap_ema_us) + 10));
        a.arm_ticks = 100 * c->arm.budget_us;
    }
    vp::StretchArgs st{};
    if (gather) st.rep = *gather;            // (the plain form's only use of the sampler arguments: where the results go)
    const int split = gather ? 0 : walker_split_for(c, c->policy_W > 0 ? c->policy_W : W);
    c->last_split = split;
    if (armed) launch_walker_any<false, true>(c, W, a, st, s, split);
    else launch_walker_any<false>(c, W, a, st, s, split);
}

// One stretch-move half-step of the active half (nS walkers) as ONE launch: proposal, lnprob and accept/reject inside
// each walker's workgroup (walker_kernel<.., SAMPLER = true>).
// (lc_row0: first row of the record workspace this launch may use -- two half-steps in flight at once, vp_stretch_run's
//  overlapped form, must not share rows)
void launch_walker_stretch(vp_ctx* c, int nS, const vp::StretchArgs& st, hipStream_t s, int lc_row0 = 0, int split = 0) {
    const size_t nrec = (size_t)(c->inst[0].dev.L + c->inst[0].dev.NCm) * vp::LC_STRIDE;
    if (split > 0) {        // (every workgroup of a walker has its own record rows: the second launch in flight starts behind nS x split)
        vp::WalkerArgs a{nullptr, c->d_lb, c->d_ub, c->d_lc + (size_t)lc_row0 * split * nrec, nullptr, c->inst[0].sum_logw, c->D, 0,
                         (st.ovl && c->tune.walker_prio < 0) ? 1 : walker_prio_for(c, nS * split), 0ull};
        launch_walker_any<true>(c, nS, a, st, s, split, lc_row0);
        return;
    }
    const vp::WalkerArgs a{nullptr, c->d_lb, c->d_ub, c->d_lc + (size_t)lc_row0 * nrec, nullptr, c->inst[0].sum_logw, c->D,
                           (int)(walker_wave_lds(c) / sizeof(double)),
                           (st.ovl && c->tune.walker_prio < 0) ? 1 : walker_prio_for(c, nS),     // (overlapped half-steps share the CUs)
                           walker_perm_for(c, nS)};
    launch_walker_any<true>(c, nS, a, st, s);
}

int enqueue_lnprob(vp_ctx* c, int W, const double* d_theta, double* d_out, hipStream_t s) {
    int tile_off = 0;
    const bool prof = c->profiling;
    size_t m0 = prof ? prof_mark(c, s) : 0;
    c->last_kind = 0;
    c->last_ff = vp_ctx::LastFF{};
    const int Wp = c->policy_W > 0 ? c->policy_W : W;       // rows the launch structure is chosen for
    if (walker_applies(c, Wp)) {
        c->last_kind = 1;
        launch_walker(c, W, d_theta, d_out, s, c->gather_rep);
        if (prof) {
            size_t m1 = prof_mark(c, s);
            c->spans.push_back({m0, m1, 1});
        }
        HIP_TRY(c, hipGetLastError());
        return VP_OK;
    }
    // Geometry: a batch whose full-size tiles would leave most wave slots empty is cut into one-pass
    // tiles instead (twice the workgroups, half the per-wave latency): measured better up to 384
    // walkers x 12 tiles (30.2 vs 31.1 us), equal at 448, worse at 512 (256 CUs x 4 SIMDs x 6 waves =
    // 6144 slots).
    int sel = ((long)Wp * c->total_tiles_g[0] <= 4800) ? 1 : 0;
    if (c->tune.geom >= 0) sel = c->tune.geom ? 1 : 0;
    const int ntot = c->total_tiles_g[sel];
    // Final reduction (bit-identical either way, see tile_kernel): fused into the tile kernel -- the
    // last-arriving tile of a walker, by ticket -- while the batch leaves wave slots empty; a small launch of
    // its own (one lane per walker) once the batch fills them, where the ticket's L2 round trips at the end of
    // every tile wave cost more than a launch.  Measured on C1 (us per pass, own launch / ticket): 256
    // walkers 26.5 / 26.7, 512: 32.3 / 33.0-33.9, 2048: 88.9 / 92.5, 8192: 314 / 329.
    int fmode = ((long)Wp * c->total_tiles_g[0] < 6144) ? 1 : 0;
    if (c->tune.finalize >= 0) fmode = c->tune.finalize ? 1 : 0;      // 0: own launch, 1: ticket
    if (c->gather_rep) fmode = 0;                    // (the finalize launch is what writes into the ranks' gathered vectors)
    const bool fused = fmode != 0;
    const vp::Replicas frep = c->gather_rep ? *c->gather_rep : vp::Replicas{};
    const vp::FinalizeArgs fin{c->d_ticket, c->d_tile_off + sel * (c->inst.size() + 1), c->d_sum_logw,
                               d_out, (int)c->inst.size(), ntot, fmode};
    // ---- several instruments with one set of records, small batch: all their tiles in ONE launch (tile_kernel_multi)
    {
        const size_t ni = c->inst.size();
        bool multi = !prof && ni >= 2 && ni <= 4 && c->tune.tile_multi != 0 && !c->tune.no_shared_prep;
        size_t lds = 0;
        long waves = 0;
        for (size_t k = 0; multi && k < ni; ++k) {
            const Instrument& in = c->inst[k];
            if (k > 0 && !in.same_lines_as_prev) multi = false;
            if (in.lds_w == 0 || in.dev.method != c->inst[0].dev.method || in.nanfix) multi = false;
            if (in.needs_generic && in.dev.method == VP_VOIGT_WOFZ) multi = false;
            lds = std::max(lds, in.lds_w);
            waves += (long)Wp * in.dev_w.ntiles;
        }
        // (measured on C3, two instruments of 8192 pixels = 52 single-wave tiles per walker, us per pass, launches / one launch:
        //  64 walkers 67.5 / 47.6, 256: 89.2 / 92.5, 512: 143.1 / 143.9, 1024: 229 / 251, 2048: 401 / 468 -- the long LSF's tiles carry
        //  more halo as single waves, which only a batch that leaves the GPU mostly idle does not feel)
        if (multi && c->tune.tile_multi < 0 && waves > 6144) multi = false;
        if (multi) {
            c->last_kind = 3;
            const Instrument& in0 = c->inst[0];
            if (prof) (void)0;
            launch_prep(c, in0, d_theta, W, 1, d_out, (int*)nullptr, s);
            vp::InstDev dk[4] = {in0.dev_w, in0.dev_w, in0.dev_w, in0.dev_w};
            vp::TileMulti tb{};
            int tsum = 0;
            for (size_t k = 0; k < ni; ++k) {
                dk[k] = c->inst[k].dev_w;
                tb.off[k] = tsum;
                if (k > 0) tb.t[k - 1] = tsum;
                tsum += dk[k].ntiles;
            }
            for (size_t k = ni; k < 4; ++k) { tb.t[k - 1] = tsum; tb.off[k] = tsum; }
            const int nt = c->total_tiles_w;
            // (final reduction by its own launch: with these many single-wave tiles per walker the ticket's round trips at the
            //  end of every wave cost more -- C3 at 64 walkers 56.9 us by ticket, 47.6 by launch; scripts/structure_check.py)
            int fm = 0;
            if (c->tune.finalize >= 0) fm = c->tune.finalize ? 1 : 0;
            if (c->gather_rep) fm = 0;
            const vp::FinalizeArgs fw{c->d_ticket, c->d_tile_off + 2 * (ni + 1), c->d_sum_logw, d_out, (int)ni, nt, fm};
            const dim3 grid(W, nt), block(64);
            if (in0.dev.method == VP_VOIGT_FAST)
                hipLaunchKernelGGL((vp::tile_kernel_multi<1>), grid, block, lds, s, dk[0], dk[1], dk[2], dk[3], tb, (int)ni, c->d_lc, c->d_flags, c->d_partial, nt, fw);
            else
                hipLaunchKernelGGL((vp::tile_kernel_multi<0>), grid, block, lds, s, dk[0], dk[1], dk[2], dk[3], tb, (int)ni, c->d_lc, c->d_flags, c->d_partial, nt, fw);
            if (!fm) {
                vp::FinalizeByValue bv{};
                bv.n_inst = (int)ni;
                for (size_t k = 0; k < ni; ++k) { bv.tile_off[k] = tb.off[k]; bv.sum_logw[k] = c->inst[k].sum_logw; }
                bv.tile_off[ni] = nt;
                hipLaunchKernelGGL((vp::finalize_kernel<true>), dim3((W + 63) / 64), dim3(64), 0, s, c->d_partial, nt, W, c->d_flags, fw, bv, frep);
            }
            HIP_TRY(c, hipGetLastError());
            return VP_OK;
        }
    }
    bool ff_made = false;                            // this instrument's expansions came with the previous one's launch
    GenFlags gf{c->d_genflag, nullptr, nullptr, nullptr, 0};     // (the flags of the records in the workspace: shared by instruments that share those)
    int rc_gf = VP_OK;
    for (size_t k = 0; k < c->inst.size(); ++k) {
        const Instrument& in = c->inst[k];
        const bool gen = in.needs_generic && in.dev.method == VP_VOIGT_WOFZ;
        if (!(k > 0 && in.same_lines_as_prev && !c->tune.no_shared_prep)) {       // (same line tables as the previous instrument: its records and
                                                       // generic-path flags are still in the workspace)
            if (gen && (rc_gf = genflag_acquire(c, W, s, &gf))) return rc_gf;
            launch_prep(c, in, d_theta, W, k == 0 ? 1 : 0, d_out, gen ? gf.use : (int*)nullptr, s, gen ? gf.clear : (int*)nullptr,
                        gen ? gf.any : (int*)nullptr, gen ? gf.any_clear : (int*)nullptr);
        }
        const vp::InstDev& geom = sel ? in.dev_s : in.dev;
        // (the extra launch costs ~20 us; what it saves grows with walkers x blocks x lines -- measured on C2, us per pass
        // without / with: 128 walkers 54.8 / 60.5, 256 (0.42e6): 72.7 / 71.5, 512: 111.8 / 98.3, 1024: 188.8 / 153.1,
        // 2048: 349.0 / 260.5.  Below that the lines are walked directly unless "farfield" = 1 asks for the expansions.
        // What the expansions save is one evaluation per pass for every ITEM they cover (a multipole cluster or a line
        // outside clusters: C2 has 9, a 40-line FeII fit of 8 clusters 8, one of 20 lines 4), ff_cover of them (estimated
        // when the instrument is added); the crossover sits near 2e5 covered (walker, block, item) triples: C2 at 256
        // walkers (0.195e6) 72.7 / 71.5 us, the 40-line FeII fit at 1024 walkers (0.17e6) 164.9 / 161.1, the 20-line one
        // at 1024 (0.09e6) 90.4 / 93.2.
        // (round 3, after the launch itself got cheaper -- records staged in LDS, faster record preparation in front of it --: on
        //  from 1.5e5 covered triples: C2 at 256 walkers (1.95e5) 57.2 / 54.9, C4 at 64 (0.9e5) 119.5 / 130.8.)
        // (instruments that share their records come in pairs whose expansions are ONE launch: the pair's triples count together
        //  -- C3, us per pass without / with: 128 walkers 66.4 / 70.3, 256: 68.5 / 67.0, 512: 103.8 / 94.4)
        auto ff_score = [&](size_t i) {
            const Instrument& x = c->inst[i];
            const vp::InstDev& gx = sel ? x.dev_s : x.dev;
            return x.ff_on ? x.ff_cover * (double)Wp * gx.ntiles * gx.ff_nblk * x.ff_items : 0.0;
        };
        auto ff_wanted = [&](size_t i) {
            const Instrument& x = c->inst[i];
            if (!x.ff_on || !c->d_ff || c->tune.farfield == 0) return false;
            if (c->tune.farfield > 0) return true;
            double score = ff_score(i);
            if (!c->tune.no_shared_prep && !prof) {
                size_t r0 = i;
                while (r0 > 0 && c->inst[r0].same_lines_as_prev) --r0;
                const size_t partner = r0 + ((i - r0) ^ 1);
                const bool in_run = partner < c->inst.size() && (partner < i ? true : c->inst[partner].same_lines_as_prev) &&
                                    (partner > i || c->inst[i].same_lines_as_prev);
                if (in_run && (sel ? c->inst[partner].dev_s : c->inst[partner].dev).ff_members == (sel ? x.dev_s : x.dev).ff_members)
                    score += ff_score(partner);
            }
            // (instruments whose cluster members enter the expansions line by line pay more for the launch: C4 at 64 walkers
            //  -- 1.8e5 -- 119.5 / 130.3 us, at 512 walkers 636 / 529)
            return score >= ((sel ? x.dev_s : x.dev).ff_members ? 3.0e5 : 1.5e5);
        };
        // every instrument's expansions have their own stretch of the workspace (W x its blocks), in instrument order
        size_t ff_off = 0;
        for (size_t i = 0; i < k; ++i) {
            const vp::InstDev& gi = sel ? c->inst[i].dev_s : c->inst[i].dev;
            if (c->inst[i].ff_on) ff_off += (size_t)W * gi.ntiles * gi.ff_nblk * vp::FF_STRIDE;
        }
        double* ff = ff_wanted(k) ? c->d_ff + ff_off : nullptr;
        if (ff && !ff_made) {                            // the blocks' far-field expansions from the records just made
            c->last_kind = 2;
            vp::InstDev g2 = geom;
            g2.ff = ff;
            const int nbk = geom.ntiles * geom.ff_nblk;
            if (c->last_ff.inst < 0) c->last_ff = vp_ctx::LastFF{ff, W, nbk, g2.ff_members, (int)k};
            const size_t ffl = vp::farfield_lds_bytes(in.lines.L, in.lines.NCm);
            // the next instrument's too, in the same launch, when it has these line tables (the same records)
            const Instrument* nx = (k + 1 < c->inst.size() && !prof) ? &c->inst[k + 1] : nullptr;
            if (nx && nx->same_lines_as_prev && !c->tune.no_shared_prep && ff_wanted(k + 1) && (sel ? nx->dev_s : nx->dev).ff_members == g2.ff_members) {
                vp::InstDev g3 = sel ? nx->dev_s : nx->dev;
                g3.ff = ff + (size_t)W * nbk * vp::FF_STRIDE;
                const int nbk1 = g3.ntiles * g3.ff_nblk, nbx0 = (nbk + 63) / 64, nbx1 = (nbk1 + 63) / 64;
                if (g2.ff_members) hipLaunchKernelGGL((vp::farfield_kernel2<9, true>), dim3(nbx0 + nbx1, W), dim3(64 * vp::FF_WAVES), ffl, s, g2, g3, nbx0, in.lines, c->d_lc, W);
                else hipLaunchKernelGGL((vp::farfield_kernel2<6, false>), dim3(nbx0 + nbx1, W), dim3(64 * vp::FF_WAVES), ffl, s, g2, g3, nbx0, in.lines, c->d_lc, W);
                ff_made = true;                          // (consumed by the next instrument's pass of this loop)
            } else {
                if (g2.ff_members) hipLaunchKernelGGL((vp::farfield_kernel<9, true>), dim3((nbk + 63) / 64, W), dim3(64 * vp::FF_WAVES), ffl, s, g2, in.lines, c->d_lc, W);
                else hipLaunchKernelGGL((vp::farfield_kernel<6, false>), dim3((nbk + 63) / 64, W), dim3(64 * vp::FF_WAVES), ffl, s, g2, in.lines, c->d_lc, W);
            }
        } else if (ff) {
            c->last_kind = 2;
            ff_made = false;
        }
        size_t m1 = prof ? prof_mark(c, s) : 0;
        launch_tile<0, false>(in, c->d_lc, c->d_flags, c->d_partial, ntot, tile_off, W, s, fin,
                              gen ? gf.use : (const int*)nullptr, &geom, 1, ff);
        if (gen)
            launch_tile<0, true>(in, c->d_lc, c->d_flags, c->d_partial, ntot, tile_off, W, s, fin, gf.use, &geom, 1, nullptr, gf.slots);
        if (prof) {
            size_t m2 = prof_mark(c, s);
            c->spans.push_back({m0, m1, 0});
            c->spans.push_back({m1, m2, 1});
            m0 = m2;
        }
        tile_off += geom.ntiles;
    }
    if (!fused) {
        vp::FinalizeByValue bv{};
        const size_t ni = c->inst.size();
        if (ni <= (size_t)vp::FIN_MAX_INST) {
            bv.n_inst = (int)ni;
            int off = 0;
            for (size_t k = 0; k < ni; ++k) {
                bv.tile_off[k] = off;
                off += (sel ? c->inst[k].dev_s : c->inst[k].dev).ntiles;
                bv.sum_logw[k] = c->inst[k].sum_logw;
            }
            bv.tile_off[ni] = off;
            hipLaunchKernelGGL((vp::finalize_kernel<true>), dim3((W + 63) / 64), dim3(64), 0, s, c->d_partial, ntot, W, c->d_flags, fin, bv, frep);
        } else {
            hipLaunchKernelGGL((vp::finalize_kernel<false>), dim3((W + 63) / 64), dim3(64), 0, s, c->d_partial, ntot, W, c->d_flags, fin, bv, frep);
        }
        if (prof) {
            size_t m3 = prof_mark(c, s);
            c->spans.push_back({m0, m3, 2});
        }
    }
    HIP_TRY(c, hipGetLastError());
    return VP_OK;
}

// ---- pre-armed launches (Tuning::prearm, WalkerArgs::arm_*) -------------------------------------------------------------
// (all called with c->mu held)
// what the host writes into the slots (through the BAR: stores only, never a load; the word goes last)
void prearm_push(vp_ctx* c, int W, const double* theta, int code) {
    const int n = c->arm.slot_doubles, D = c->D;
    const uint64_t word = ((uint64_t)c->arm.seq << 2) | (uint64_t)code;
    double* s = c->arm.slots;
    if (!theta) {
        for (int w = 0; w < W; ++w) reinterpret_cast<volatile uint64_t*>(s + (size_t)w * n)[n - 1] = word;
    } else if (n == 8) {
        // one 64-byte line per row: the row, padding, the word -- written in address order, so the write-combining buffer goes out
        // as one burst
        // (no fence between row and word -- it would cut the burst in two --: the word carries a fold of the row's bits in its upper
        //  half and the waiting workgroup takes the row only when it matches, arm_wait)
        for (int w = 0; w < W; ++w) {
            volatile double* q = s + (size_t)w * 8;
            const double* t = theta + (size_t)w * D;
            uint64_t h = 0;
            int k = 0;
            for (; k < D; ++k) { uint64_t b; std::memcpy(&b, t + k, 8); h ^= b; q[k] = t[k]; }
            for (; k < 7; ++k) q[k] = 0.0;
            reinterpret_cast<volatile uint64_t*>(q)[7] = word | ((uint64_t)(uint32_t)(h ^ (h >> 32)) << 32);
        }
    } else {
        for (int w = 0; w < W; ++w) std::memcpy(s + (size_t)w * n, theta + (size_t)w * D, (size_t)D * sizeof(double));
        __builtin_ia32_sfence();            // rows before words (write-combining buffers of different lines go out in any order)
        for (int w = 0; w < W; ++w) reinterpret_cast<volatile uint64_t*>(s + (size_t)w * n)[n - 1] = word;
    }
    __builtin_ia32_sfence();
}
// tell a waiting launch to leave: it has read nothing that depends on a batch and written nothing
void prearm_cancel(vp_ctx* c) {
    if (!c->arm.live) return;
    prearm_push(c, c->arm.W, nullptr, vp::ARM_LEAVE);
    c->arm.live = false;
    ++c->arm.cancelled;
}
// may the NEXT call of this shape be started through a pre-armed launch?
bool prearm_eligible(vp_ctx* c, int W, size_t theta_bytes) {
    return c->tune.prearm != 0 && c->arm.bar != 0 && c->tune.host_spin >= 2 && !c->sentinel_unsafe && !c->gather_rep && !c->profiling && c->policy_W == 0 &&
           theta_bytes <= (size_t)std::max(0l, c->tune.zerocopy_max) && !c->tune.no_zerocopy && walker_applies(c, W);
}
int prearm_launch(vp_ctx* c, int W) {
    if (c->arm.bar < 0) {
        // device memory the CPU can write: every GPU of this class has its whole memory behind the PCIe BAR; without it there
        // are no pre-armed launches
        int v = 0;
        c->arm.bar = (hipDeviceGetAttribute(&v, hipDeviceAttributeIsLargeBar, c->device) == hipSuccess && v) ? 1 : 0;
        (void)hipGetLastError();
        if (!c->arm.bar) return VP_OK;
    }
    if (!c->arm.h) {
        if (hipHostMalloc((void**)&c->arm.h, 256, hipHostMallocMapped) != hipSuccess) { c->arm.h = nullptr; c->arm.bar = 0; (void)hipGetLastError(); return VP_OK; }
        std::memset(c->arm.h, 0, 256);
        HIP_TRY(c, hipHostGetDevicePointer((void**)&c->arm.h_dev, c->arm.h, 0));
    }
    const int n = 8 * ((c->D + 1 + 7) / 8);
    if (!c->arm.slots || c->arm.slot_rows < W || c->arm.slot_doubles != n) {
        // (no launch is waiting here: a batch of another shape has sent it away -- hipFree waits for it)
        if (c->arm.slots) HIP_TRY(c, hipFree(c->arm.slots));
        c->arm.slots = nullptr;
        const size_t bytes = (size_t)W * n * sizeof(double);
        if (hipExtMallocWithFlags((void**)&c->arm.slots, bytes, hipDeviceMallocFinegrained) != hipSuccess) {
            c->arm.slots = nullptr; c->arm.bar = 0; (void)hipGetLastError(); return VP_OK;
        }
        HIP_TRY(c, hipMemset(c->arm.slots, 0, bytes));
        HIP_TRY(c, hipDeviceSynchronize());
        {   // is the allocation really in this process's address space?  (asked of the kernel, not found out by a fault)
            const uintptr_t pg = (uintptr_t)sysconf(_SC_PAGESIZE), a0 = (uintptr_t)c->arm.slots & ~(pg - 1);
            unsigned char vec[1];
            if (mincore((void*)a0, 1, vec) != 0 && errno == ENOMEM) {
                (void)hipFree(c->arm.slots);
                c->arm.slots = nullptr; c->arm.bar = 0;
                return VP_OK;
            }
        }
        c->arm.slot_rows = W;
        c->arm.slot_doubles = n;
    }
    c->arm.seq = (c->arm.seq + 1) & 0x3fffffffu;
    if (c->arm.seq == 0) c->arm.seq = 1;
    hipStream_t s = c->stream;
    double* dp = c->h_pinned_dev;
    launch_walker(c, W, dp, dp + (size_t)W * c->D, s, nullptr, true);
    HIP_TRY(c, hipGetLastError());
    c->arm.live = true;
    c->arm.dirty = true;
    c->arm.live_stream = s;
    c->arm.W = W;
    return VP_OK;
}
// An entry point about to enqueue on a stream of the caller's: a pre-armed launch -- waiting, sent away or expired -- may still be
// writing its rehearsal records into the context's record workspace on the context's own stream; order the caller's stream behind it.
int foreign_stream_fence(vp_ctx* c, hipStream_t s) {
    if (!c->arm.dirty || !c->arm.live_stream || s == c->arm.live_stream) return VP_OK;       // (same stream: stream order does it)
    if (!c->arm.ev) HIP_TRY(c, hipEventCreateWithFlags(&c->arm.ev, hipEventDisableTiming));
    HIP_TRY(c, hipEventRecord(c->arm.ev, c->arm.live_stream));
    HIP_TRY(c, hipStreamWaitEvent(s, c->arm.ev, 0));
    c->arm.dirty = false;
    return VP_OK;
}
// Contexts alive in this process: a program that ends without vp_ctx_destroy must not leave a launch waiting on the GPU while the
// runtime is taken down under it -- at exit every waiting launch is told to leave (CPU stores into its slots, no HIP call).
std::mutex g_ctx_mu;
std::vector<vp_ctx*> g_ctxs;
void prearm_leave_all_at_exit() {
    std::lock_guard<std::mutex> g(g_ctx_mu);
    bool any = false;
    const pid_t me = getpid();
    for (vp_ctx* c : g_ctxs)
        // (a forked child that leaves through exit() runs this handler too: the parent's device mappings are not in its address
        //  space -- ROCm marks them MADV_DONTFORK -- and the launch is the parent's to send away)
        if (c->pid == me && c->mu.try_lock()) {
            any |= c->arm.live;
            prearm_cancel(c);
            c->mu.unlock();
        }
    if (any) std::this_thread::sleep_for(std::chrono::microseconds(100));      // (a workgroup polls its slot every ~0.5 us)
}
void ctx_register(vp_ctx* c) {
    std::lock_guard<std::mutex> g(g_ctx_mu);
    static bool hooked = false;
    if (!hooked) { std::atexit(prearm_leave_all_at_exit); hooked = true; }     // (behind the HIP runtime's own handlers: runs before them)
    c->pid = getpid();
    g_ctxs.push_back(c);
}
void ctx_unregister(vp_ctx* c) {
    std::lock_guard<std::mutex> g(g_ctx_mu);
    g_ctxs.erase(std::remove(g_ctxs.begin(), g_ctxs.end(), c), g_ctxs.end());
}
// A launch that waits holds its compute units: work of ANOTHER context of this process on the same GPU would sit behind it until it
// expires.  Every entry point therefore sends the other contexts' waiting launches away first (one short lock when there are none).
void prearm_cancel_others(vp_ctx* c) {
    std::lock_guard<std::mutex> g(g_ctx_mu);
    if (g_ctxs.size() < 2) return;
    for (vp_ctx* o : g_ctxs)
        if (o != c && o->device == c->device && o->mu.try_lock()) {       // (a context busy in a call of its own has no launch waiting)
            prearm_cancel(o);
            o->mu.unlock();
        }
}
struct CtxGuard {                 // every entry but vp_lnprob_batch itself: lock the context and send a waiting launch away
    std::lock_guard<std::mutex> g;
    explicit CtxGuard(vp_ctx* c) : g(c->mu) { prearm_cancel(c); prearm_cancel_others(c); }
};

// (called with c->mu held)
int check_batch_args(vp_ctx* c, int W, int D, const void* a, const void* b) {
    if (c->D <= 0) return fail(c, VP_ESTATE, "vp_set_bounds has not been called");
    if (c->inst.empty()) return fail(c, VP_ESTATE, "no instrument has been added");
    if (D != c->D) return fail(c, VP_EINVAL, "theta has D=" + std::to_string(D) + " but the context was set up with D=" + std::to_string(c->D));
    if (W < 0) return fail(c, VP_EINVAL, "negative batch size");
    if (W > 0 && (!a || !b)) return fail(c, VP_EINVAL, "NULL theta/out");
    return VP_OK;
}

}  // namespace

extern "C" {

const char* vp_version(void) { return "rbvfit_amd " RBVFIT_AMD_VERSION " (gfx950, hip)"; }

int vp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* vp_last_error(const vp_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int vp_ctx_create(vp_ctx** out, int device_id) {
    if (!out) return fail(nullptr, VP_EINVAL, "out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, VP_EHIP, std::string("no usable HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0"));
    if (device_id < 0 || device_id >= n) return fail(nullptr, VP_EINVAL, "device_id out of range");
    vp_ctx* c = new (std::nothrow) vp_ctx();
    if (!c) return fail(nullptr, VP_ENOMEM, "out of host memory");
    c->device = device_id;
    c->tune = tuning_from_env();
    if ((e = hipSetDevice(device_id)) != hipSuccess || (e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) {
        std::string m = std::string("context creation failed: ") + hipGetErrorString(e);
        delete c;
        return fail(nullptr, VP_EHIP, m);
    }
    // Dynamic LDS a workgroup may ask for: the device's per-workgroup limit (160 KiB on MI355X).  The
    // tile and walker kernels are told they may use all of it (a no-op where the runtime does not need it).
    int lds_max = 0;
    if (hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, device_id) == hipSuccess && lds_max > 0)
        c->lds_limit = (size_t)lds_max;
    for (const void* f : {(const void*)vp::walker_kernel<0, false, false>, (const void*)vp::walker_kernel<0, true, false>,
                          (const void*)vp::walker_kernel<1, false, false>, (const void*)vp::walker_kernel<0, false, true>,
                          (const void*)vp::walker_kernel<1, false, true>,
                          (const void*)vp::walker_kernel<0, false, false, true>, (const void*)vp::walker_kernel<1, false, false, true>,
                          (const void*)vp::walker_kernel<0, true, false, true>, (const void*)vp::walker_kernel<0, false, false, false, 1>,
                          (const void*)vp::walker_kernel<0, false, false, false, 0, true>, (const void*)vp::walker_kernel<1, false, false, false, 0, true>,
                          (const void*)vp::walker_kernel<0, false, true, false, 0, true>, (const void*)vp::walker_kernel<1, false, true, false, 0, true>,
                          (const void*)vp::walker_kernel<0, false, false, true, 0, true>, (const void*)vp::walker_kernel<1, false, false, true, 0, true>,
                          (const void*)vp::walker_kernel2<0, false, true>, (const void*)vp::walker_kernel2<1, false, true>,
                          (const void*)vp::walker_kernel4<0, false, true>, (const void*)vp::walker_kernel4<1, false, true>,
                          (const void*)vp::tile_kernel1<0, true>, (const void*)vp::tile_kernel1<0, false>, (const void*)vp::tile_kernel1<1, false>,
                          (const void*)vp::tile_generic_kernel<0, false>, (const void*)vp::tile_generic_kernel<1, false>, (const void*)vp::tile_generic_kernel<1, true>,
                          (const void*)vp::tile_generic_kernel<2, false>, (const void*)vp::tile_kernel<0, 1, false, true>,
                          (const void*)vp::walker_kernel2<0, false>, (const void*)vp::walker_kernel2<0, true>,
                          (const void*)vp::walker_kernel2<1, false>, (const void*)vp::walker_kernel2<1, true>,
                          (const void*)vp::walker_kernel4<0, false>, (const void*)vp::walker_kernel4<0, true>,
                          (const void*)vp::walker_kernel4<1, false>, (const void*)vp::walker_kernel4<1, true>,
                          (const void*)vp::tile_kernel<0, 0, false, true>, (const void*)vp::tile_kernel_multi<0>, (const void*)vp::tile_kernel_multi<1>,
                          (const void*)vp::tile_kernel<0, 0, false>, (const void*)vp::tile_kernel<0, 1, false>, (const void*)vp::tile_kernel<0, 2, false>,
                          (const void*)vp::tile_kernel<1, 0, false>, (const void*)vp::tile_kernel<1, 1, false>,
                          (const void*)vp::tile_kernel<1, 2, false>})
        (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_limit);
    (void)hipGetLastError();
    ctx_register(c);
    *out = c;
    return VP_OK;
}

int vp_set_option(vp_ctx* c, const char* name, long value) {
    if (!c) return VP_EINVAL;
    CtxGuard g(c);
    if (!name) return fail(c, VP_EINVAL, "vp_set_option: NULL name");
    for (const Knob& k : g_knobs)
        if (std::strcmp(k.name, name) == 0) {
            set_knob(c->tune, k, value);
            return VP_OK;
        }
    return fail(c, VP_EINVAL, std::string("vp_set_option: unknown option '") + name + "'");
}

static void gather_release(vp_ctx* c) {
    vp_ctx::Gather& g = c->gather;
    for (int r = 0; r < g.world && r < vp::MAX_REPLICAS; ++r) {
        if (r == g.rank) continue;
        if (g.local_peers) continue;           // (plain pointers into contexts of this process: theirs to free)
        if (g.peer_buf[r]) (void)hipIpcCloseMemHandle(g.peer_buf[r]);
        if (g.peer_flags[r]) (void)hipIpcCloseMemHandle(g.peer_flags[r]);
    }
    if (g.buf) (void)hipFree(g.buf);
    if (g.flags) (void)hipFree(g.flags);
    if (g.done) (void)hipFree(g.done);
    (void)hipGetLastError();
    g = vp_ctx::Gather{};
}

int vp_ctx_destroy(vp_ctx* c) {
    if (!c) return VP_OK;
    ctx_unregister(c);
    { std::lock_guard<std::mutex> g(c->mu); prearm_cancel(c); }
    hipSetDevice(c->device);
    hipDeviceSynchronize();
    if (c->arm.h) hipHostFree(c->arm.h);
    if (c->arm.ev) hipEventDestroy(c->arm.ev);
    if (c->arm.slots) hipFree(c->arm.slots);
    if (c->h_gen_any) hipHostFree(c->h_gen_any);
    for (auto& in : c->inst) for (void* p : in.allocs) hipFree(p);
    for (void* p : {(void*)c->d_lb, (void*)c->d_ub, (void*)c->d_theta, (void*)c->d_out, (void*)c->d_lc, (void*)c->d_partial,
                    (void*)c->d_flags, (void*)c->d_ticket, (void*)c->d_genflag, (void*)c->d_tile_off, (void*)c->d_sum_logw, (void*)c->d_scratch,
                    (void*)c->d_ff})
        if (p) hipFree(p);
    if (c->h_pinned) hipHostFree(c->h_pinned);
    if (c->h_done) hipHostFree(c->h_done);
    gather_release(c);
    for (hipEvent_t e : c->ev_pool) hipEventDestroy(e);
    if (c->ev_fork) hipEventDestroy(c->ev_fork);
    if (c->ev_join) hipEventDestroy(c->ev_join);
    if (c->stream2) hipStreamDestroy(c->stream2);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
    return VP_OK;
}

int vp_set_bounds(vp_ctx* c, int D, const double* lb, const double* ub) {
    if (!c) return VP_EINVAL;
    CtxGuard g(c);
    if (c->h_gen_any) c->h_gen_any[0] = 0;          // (new prior box: nothing known about walkers outside the fast domain)
    if (D <= 0 || !lb || !ub) return fail(c, VP_EINVAL, "vp_set_bounds: D must be positive and lb/ub non-NULL");
    if (!c->inst.empty() && D != c->D) return fail(c, VP_ESTATE, "vp_set_bounds: D differs from the D the instruments were validated against");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipDeviceSynchronize());
    if (c->d_lb) HIP_TRY(c, hipFree(c->d_lb));
    if (c->d_ub) HIP_TRY(c, hipFree(c->d_ub));
    c->d_lb = c->d_ub = nullptr;
    int rc;
    if ((rc = upload<double>(c, nullptr, lb, D, &c->d_lb))) return rc;
    if ((rc = upload<double>(c, nullptr, ub, D, &c->d_ub))) return rc;
    if (D != c->D) {   // drop the workspace, it is resized lazily
        for (void* p : {(void*)c->d_theta, (void*)c->d_out, (void*)c->d_lc, (void*)c->d_partial, (void*)c->d_flags})
            if (p) hipFree(p);
        c->d_theta = c->d_out = c->d_lc = c->d_partial = nullptr; c->d_flags = nullptr; c->capW = 0;
    }
    c->D = D;
    if (carries_sentinel(lb, D) || carries_sentinel(ub, D)) c->sentinel_unsafe = true;
    c->h_lb.assign(lb, lb + D);
    c->h_ub.assign(ub, ub + D);
    for (auto& in : c->inst) analyse_generic(c, in);
    return VP_OK;
}

int vp_add_instrument(vp_ctx* c, int P, const double* wave, const double* flux, const double* inv_sigma2,
                      const double* log_inv_sigma2, int L, const double* lambda0, const double* gamma,
                      const double* f, const double* zfac, const int32_t* N_idx, const int32_t* b_idx,
                      const int32_t* v_idx, int K, const double* taps, int lsf_mode, int voigt_method,
                      int* inst_index) {
    if (!c) return VP_EINVAL;
    CtxGuard g(c);
    if (c->h_gen_any) c->h_gen_any[0] = 0;
    if (c->D <= 0) return fail(c, VP_ESTATE, "vp_add_instrument: call vp_set_bounds first (theta indices are validated against D)");
    if (P <= 0 || !wave || !flux || !inv_sigma2 || !log_inv_sigma2) return fail(c, VP_EINVAL, "vp_add_instrument: empty or NULL spectrum");
    if (L <= 0 || !lambda0 || !gamma || !f || !zfac || !N_idx || !b_idx || !v_idx) return fail(c, VP_EINVAL, "vp_add_instrument: empty or NULL line tables");
    if (voigt_method != VP_VOIGT_WOFZ && voigt_method != VP_VOIGT_FAST) return fail(c, VP_EINVAL, "vp_add_instrument: unknown voigt_method");
    if (lsf_mode < VP_LSF_NONE || lsf_mode > VP_LSF_ASTROPY_EXTEND) return fail(c, VP_EINVAL, "vp_add_instrument: unknown lsf_mode");
    if (lsf_mode != VP_LSF_NONE && (K <= 0 || !taps)) return fail(c, VP_EINVAL, "vp_add_instrument: lsf_mode set but no taps");
    if (lsf_mode != VP_LSF_NONE && (K % 2) == 0) return fail(c, VP_EINVAL, "vp_add_instrument: the number of taps must be odd");
    if (K > 2049) return fail(c, VP_EINVAL, "vp_add_instrument: more than 2049 LSF taps is not supported");
    if (L > 4096) return fail(c, VP_EINVAL, "vp_add_instrument: more than 4096 lines per instrument is not supported");
    for (int l = 0; l < L; ++l) {
        if (N_idx[l] < 0 || N_idx[l] >= c->D || b_idx[l] < 0 || b_idx[l] >= c->D || v_idx[l] < 0 || v_idx[l] >= c->D)
            return fail(c, VP_EINVAL, "vp_add_instrument: theta index of line " + std::to_string(l) + " outside [0, D)");
    }
    if (carries_sentinel(wave, P) || carries_sentinel(flux, P) || carries_sentinel(inv_sigma2, P) || carries_sentinel(lambda0, L) ||
        carries_sentinel(gamma, L) || carries_sentinel(f, L) || carries_sentinel(zfac, L) || (taps && K > 0 && carries_sentinel(taps, K)))
        c->sentinel_unsafe = true;
    HIP_TRY(c, hipSetDevice(c->device));
    Instrument in;
    int rc;
    // flipped (and, for the astropy branch, normalised) taps: out[p] = sum_j kflip[j] f[p - halo_lo + j]
    std::vector<double> kflip;
    int Kuse = 1, cidx = 0;
    if (lsf_mode == VP_LSF_NONE) {
        kflip.assign(1, 1.0);
    } else {
        Kuse = K; cidx = K / 2;
        double norm = 1.0;
        if (lsf_mode == VP_LSF_ASTROPY_EXTEND) { norm = 0.0; for (int j = 0; j < K; ++j) norm += taps[j]; }
        kflip.resize(K);
        for (int j = 0; j < K; ++j) {
            double t = taps[K - 1 - j];
            kflip[j] = (lsf_mode == VP_LSF_ASTROPY_EXTEND) ? t / norm : t;
        }
    }
    std::vector<double> ginv(P);
    for (int p = 0; p < P; ++p) ginv[p] = 1.0 / wave[p];
    double *d_wave, *d_ginv, *d_flux, *d_w, *d_k, *d_l0, *d_g, *d_f, *d_z, *d_fr0;
    std::vector<double> freq0(L);
    for (int l = 0; l < L; ++l) freq0[l] = vp::C_FREQ / lambda0[l];     // voigt_model.py:143
    int *d_n, *d_b, *d_v;
#define UP(T, h, n, d) if ((rc = upload<T>(c, &in, h, n, &d))) { for (void* p : in.allocs) hipFree(p); return rc; }
    UP(double, wave, P, d_wave) UP(double, ginv.data(), P, d_ginv) UP(double, flux, P, d_flux) UP(double, inv_sigma2, P, d_w)
    kflip.resize((kflip.size() + 7) & ~(size_t)7, 0.0);       // zero-padded to whole groups of 8 (what the LSF loop reads)
    UP(double, kflip.data(), kflip.size(), d_k)
    UP(double, lambda0, L, d_l0) UP(double, freq0.data(), L, d_fr0) UP(double, gamma, L, d_g) UP(double, f, L, d_f) UP(double, zfac, L, d_z)
    UP(int, N_idx, L, d_n) UP(int, b_idx, L, d_b) UP(int, v_idx, L, d_v)
    in.d_flux = d_flux; in.d_w = d_w;
    // multipole clusters: maximal runs of >= 3 (and <= 64) consecutive lines with the same rest
    // wavelength and redshift factor = the components of one transition (voigt_model.py:391-401)
    std::vector<int> cl_mp(L, -1), cl_end(L, 0), cl_first, cl_count;
    for (int l = 0; l < L;) {
        int e = l + 1;
        while (e < L && lambda0[e] == lambda0[l] && zfac[e] == zfac[l]) ++e;
        for (int k = l; k < e; ++k) cl_end[k] = e;
        const bool enable = !c->tune.no_multipole;
        if (enable && e - l >= c->tune.multipole_min && e - l <= 64 && voigt_method == VP_VOIGT_WOFZ) {
            cl_mp[l] = (int)cl_first.size();
            cl_first.push_back(l);
            cl_count.push_back(e - l);
        }
        l = e;
    }
    const int NCm = (int)cl_first.size();
    for (int k = 0; k < NCm; ++k)
        for (int l = cl_first[k]; l < cl_first[k] + cl_count[k] && l < 128; ++l) in.member_mask[l >> 6] |= 1ull << (l & 63);
    int *d_clmp, *d_clend, *d_clfirst, *d_clcount;
    UP(int, cl_mp.data(), L, d_clmp) UP(int, cl_end.data(), L, d_clend)
    UP(int, cl_first.data(), cl_first.size(), d_clfirst) UP(int, cl_count.data(), cl_count.size(), d_clcount)
    std::vector<int4> mem_info;                      // per member of a cluster: cluster, line, first member, member count
    for (int k = 0; k < NCm; ++k) {
        const int m0 = (int)mem_info.size();
        for (int j = 0; j < cl_count[k]; ++j) mem_info.push_back(int4{k, cl_first[k] + j, m0, cl_count[k]});
    }
    int4* d_meminfo;
    UP(int4, mem_info.data(), mem_info.size(), d_meminfo)
    in.lines = vp::LinesDev{L, d_l0, d_fr0, d_g, d_f, d_z, d_n, d_b, d_v, NCm, d_clfirst, d_clcount, d_clmp, d_clend,
                            (int)mem_info.size(), d_meminfo};
    vp::InstDev& d = in.dev;
    d.P = P; d.L = L; d.K = Kuse; d.halo_lo = Kuse - 1 - cidx; d.method = voigt_method; d.line_sel = -1;
    d.NCm = NCm;
#undef UP
    // Tile geometry: one wave evaluates 2*RB chunks of 64 consecutive pixels (RB register-blocked per pass);
    // a workgroup is 1, 2 or 4 such waves.  Single-wave workgroups need no cross-wave barrier and
    // let the hardware balance the walkers' tiles; longer LSFs take wider tiles to keep the halo
    // (K-1 re-evaluated pixels per tile) a small fraction.
    int nwaves = Kuse <= 33 ? 1 : Kuse <= 65 ? 2 : 4;
    int span = 2 * 64 * vp::RB * nwaves;                 // two register-blocked passes per wave
    if (c->tune.waves > 0) nwaves = std::min(4, c->tune.waves) == 3 ? 2 : std::min(4, c->tune.waves);   // tuning experiments: 1, 2 or 4
    if (c->tune.span > 0) span = std::max(64, (c->tune.span / 64) * 64);                         // (multiple of 64)
    if (Kuse > 257) { span = std::min(8192, ((4 * Kuse + 63) / 64) * 64); nwaves = 4; }
    const int need = P + Kuse - 1;
    if (need < span) span = std::max(64, ((need + 63) / 64) * 64);
    in.nwaves = nwaves;
    d.span = span; d.TP = span - (Kuse - 1);
    d.ntiles = (P + d.TP - 1) / d.TP;
    d.wave = d_wave; d.ginv = d_ginv; d.flux = d_flux; d.w = d_w; d.kflip = d_k;
    d.rbot = nullptr;
    {
        // NaN wavelength samples make NaN model pixels.  The reference's Gaussian branch (scipy convolve1d) lets each poison the K
        // outputs around it; its CustomKernel branch (astropy convolve, nan_treatment='interpolate': voigt_model.py:227,230) leaves
        // them out and divides every output by the kernel weight of the samples it did use.  The pattern is static, so the
        // reciprocal of that weight is a per-pixel table (taps in the order the device applies them; a window of NaNs only: NaN).
        bool any_nan = false;
        for (int p = 0; p < P; ++p) any_nan |= wave[p] != wave[p];
        if (any_nan) {
            // (the other branches -- scipy's convolve1d, no LSF -- get the same instances with a table of 1 and NaN: a NaN sample
            //  poisons exactly the outputs whose K taps reach it.  Left to the arithmetic it would also poison the outputs the
            //  ZERO taps of the padded tap groups reach, 0 x NaN being NaN)
            std::vector<double> rbot(P);
            const int halo = d.halo_lo;
            for (int p = 0; p < P; ++p) {
                double bot = 0.0;
                bool hit = false;
                for (int j = 0; j < Kuse; ++j) {
                    const int q = std::min(std::max(p - halo + j, 0), P - 1);
                    if (wave[q] == wave[q]) bot += kflip[j]; else hit = true;
                }
                if (lsf_mode == VP_LSF_ASTROPY_EXTEND) rbot[p] = bot != 0.0 ? 1.0 / bot : std::nan("");
                else rbot[p] = hit ? std::nan("") : 1.0;
            }
            double* d_rbot;
            if ((rc = upload<double>(c, &in, rbot.data(), rbot.size(), &d_rbot))) { for (void* q : in.allocs) hipFree(q); return rc; }
            d.rbot = d_rbot;
            in.nanfix = true;
        }
    }
    d.core_hint = nullptr;                               // (per geometry, below)
    d.ff_tab = nullptr; d.ff = nullptr; d.ff_nblk = 0; d.ff_members = 0;
    in.dev_s = d;                                        // one-pass tiles for small batches (same LDS layout rules)
    {
        const int span_s = 64 * vp::RB * nwaves;
        if (span_s < d.span && span_s - (Kuse - 1) >= 64) {
            in.dev_s.span = span_s; in.dev_s.TP = span_s - (Kuse - 1);
            in.dev_s.ntiles = (P + in.dev_s.TP - 1) / in.dev_s.TP;
        }
    }
    // far-field expansions: per block of 64 RB evaluated pixels the centre and half-width of 1/wave (both geometries)
    in.ff_on = voigt_method == VP_VOIGT_WOFZ && L <= 128 && (c->tune.farfield > 0 || (c->tune.farfield < 0 && L >= 8)) && !in.nanfix;
    if (in.ff_on) {
        for (vp::InstDev* gd : {&in.dev, &in.dev_s}) {
            const int blk_px = 64 * vp::RB, nblk = (gd->span + blk_px - 1) / blk_px;
            std::vector<double> tab((size_t)gd->ntiles * nblk * 4, 0.0);
            for (int t = 0; t < gd->ntiles; ++t) {
                const int p0 = t * gd->TP, nout = std::min(p0 + gd->TP, P) - p0, n_eval = nout + Kuse - 1, q0 = p0 - gd->halo_lo;
                for (int b = 0; b < nblk; ++b) {
                    double* e = tab.data() + ((size_t)t * nblk + b) * 4;
                    const int i0 = b * blk_px;
                    if (i0 >= n_eval) { e[0] = 0.0; e[1] = -1.0; e[2] = 0.0; e[3] = 0.0; continue; }
                    const int i1 = std::min(i0 + blk_px, n_eval) - 1;
                    const int qa = std::min(std::max(q0 + i0, 0), P - 1), qb = std::min(std::max(q0 + i1, 0), P - 1);
                    double gmin = ginv[qa], gmax = ginv[qa];
                    for (int q = qa; q <= qb; ++q) { gmin = std::min(gmin, ginv[q]); gmax = std::max(gmax, ginv[q]); }
                    const double gc = 0.5 * (gmax + gmin), hw = 0.5 * (gmax - gmin);
                    e[0] = gc; e[1] = hw; e[2] = hw > 0.0 ? 1.0 / hw : 0.0; e[3] = gc * e[2];
                    if (!(hw >= 0.0) || !(std::fabs(gc) <= 1.79e308)) { e[1] = -1.0; e[2] = 0.0; e[3] = 0.0; }   // NaN pixels: no expansion
                }
            }
            double* d_tab;
            if ((rc = upload<double>(c, &in, tab.data(), tab.size(), &d_tab))) { for (void* p : in.allocs) hipFree(p); return rc; }
            gd->ff_tab = d_tab; gd->ff_nblk = nblk; gd->ff = nullptr;
            if (gd == &in.dev) {
                // how much of the instrument the expansions will cover, estimated for Doppler widths at the geometric
                // centre of their bounds and lines at rest in their systems: the acceptance rule of farfield_kernel in
                // velocity units (block half-width <= distance / 8, nearest pixel >= 30 Doppler widths away)
                const double ckms = 299792.458;
                long cov = 0, tot = 0;
                for (size_t bk = 0; bk < tab.size() / 4; ++bk) {
                    const double gc = tab[4 * bk], hw = tab[4 * bk + 1];
                    if (!(hw >= 0.0) || !(gc > 0.0)) continue;
                    const double hwv = ckms * hw / gc;
                    for (int l = 0; l < L; ++l) {
                        const double lam = lambda0[l] * zfac[l];
                        double bl = c->h_lb[b_idx[l]], bu = c->h_ub.size() == c->h_lb.size() ? c->h_ub[b_idx[l]] : bl;
                        const double beff = (bl > 0.0 && bu > 0.0) ? std::sqrt(bl * bu) : 20.0;
                        const double dv = ckms * std::fabs(std::log(1.0 / (gc * lam)));
                        ++tot;
                        if (hwv <= dv / 8.0 && dv - hwv >= 30.0 * beff) ++cov;
                    }
                }
                in.ff_cover = tot > 0 ? (double)cov / (double)tot : 0.0;
                {
                    // members of near clusters line by line (farfield_kernel): only where a block is narrow against the
                    // lines -- half-width <= 1/8 of a distance of ~30 Doppler widths of the narrowest line allowed
                    std::vector<double> hwvs;
                    for (size_t bk = 0; bk < tab.size() / 4; ++bk)
                        if (tab[4 * bk + 1] >= 0.0 && tab[4 * bk] > 0.0) hwvs.push_back(ckms * tab[4 * bk + 1] / tab[4 * bk]);
                    double bmin = 1e300;
                    for (int l = 0; l < L; ++l) {
                        const double bl = c->h_lb[b_idx[l]], bu = c->h_ub.size() == c->h_lb.size() ? c->h_ub[b_idx[l]] : bl;
                        bmin = std::min(bmin, (bl > 0.0 && bu > 0.0) ? std::sqrt(bl * bu) : 20.0);
                    }
                    int members = 0;
                    if (!hwvs.empty() && NCm > 0 && !c->tune.no_ff_members) {
                        std::nth_element(hwvs.begin(), hwvs.begin() + hwvs.size() / 2, hwvs.end());
                        members = hwvs[hwvs.size() / 2] <= (30.0 / 8.0) * bmin ? 1 : 0;
                    }
                    in.dev.ff_members = members;
                }
                {
                    int members = 0;
                    for (size_t k = 0; k < cl_count.size(); ++k) members += cl_count[k];
                    in.ff_items = (int)cl_count.size() + (L - members);
                }
                if (getenv("RBVFIT_AMD_VERBOSE")) fprintf(stderr, "[rbvfit_amd] instrument %zu: far-field cover estimate %.3f (%d lines, %zu blocks)\n",
                                                          c->inst.size(), in.ff_cover, L, tab.size() / 4);
            }
        }
    }
    in.dev_s.ff_members = in.dev.ff_members;
    in.lds_bytes = (size_t)(span + vp::FL_PAD + 4 + vp::DAW_LDS_DOUBLES + vp::EXP_LDS_DOUBLES + (span / 64) * ((L + 63) / 64)) * sizeof(double);
    {
        in.dev_w = in.dev;
        in.lds_w = in.lds_bytes;
        if (in.nwaves != 1) {
            const int span_w = 2 * 64 * vp::RB;
            if (span_w - (Kuse - 1) >= 64 && c->tune.span == 0) {
                in.dev_w.span = span_w; in.dev_w.TP = span_w - (Kuse - 1);
                in.dev_w.ntiles = (P + in.dev_w.TP - 1) / in.dev_w.TP;
                in.dev_w.ff_tab = nullptr; in.dev_w.ff_nblk = 0;
                in.lds_w = (size_t)(span_w + vp::FL_PAD + 4 + vp::DAW_LDS_DOUBLES + vp::EXP_LDS_DOUBLES + (span_w / 64) * ((L + 63) / 64)) * sizeof(double);
            } else {
                in.lds_w = 0;
            }
        }
    }
    // walker_kernel's split form runs on the ONE-pass geometry of single-wave tiles (dev_s), with hints of its own
    if (in.nwaves == 1 && in.dev_s.span < in.dev.span && in.dev_s.ntiles >= 2 && c->tune.span == 0) {
        in.lds_s = (size_t)(in.dev_s.span + vp::FL_PAD + 4 + vp::DAW_LDS_DOUBLES + vp::EXP_LDS_DOUBLES + (in.dev_s.span / 64) * ((L + 63) / 64)) * sizeof(double);
        std::vector<int> zeros(std::max(in.dev_s.ntiles, vp::TILE_ORDER_AT), 0);
        if ((rc = upload<int>(c, &in, zeros.data(), zeros.size(), &in.split_hint))) { for (void* p : in.allocs) hipFree(p); return rc; }
    }
    if (in.lds_bytes + (size_t)std::max(0l, c->tune.lds_pad) > c->lds_limit) {
        for (void* p : in.allocs) hipFree(p);
        return fail(c, VP_EINVAL, "vp_add_instrument: a " + std::to_string(Kuse) + "-tap LSF with " + std::to_string(L) +
                    " lines needs " + std::to_string(in.lds_bytes) + " B of LDS per tile workgroup, the device allows " +
                    std::to_string(c->lds_limit));
    }
    in.sum_logw = neumaier_sum(log_inv_sigma2, P);
    // Per geometry: walker_kernel's hints (zero), then the order in which tile_kernel hands out the tiles -- by estimated
    // cost, the most expensive first, so that a launch ends on cheap workgroups wherever the lines sit on the grid: cost =
    // the tile's pixels within +-300 km/s of a line centre (at v = 0), summed over the lines (where the line-core and
    // near-wing tiers run); ties, and grids without lines, keep the grid order.
    {
        std::vector<int> core_lo(L, 1), core_hi(L, 0);
        if (P > 1 && wave[P - 1] > wave[0] && c->tune.tile_lpt) {
            for (int l = 0; l < L; ++l) {
                const double wc = lambda0[l] * zfac[l], hw = wc * (300.0 / 299792.458);
                const int lo = (int)(std::lower_bound(wave, wave + P, wc - hw) - wave);
                const int hi = (int)(std::upper_bound(wave, wave + P, wc + hw) - wave) - 1;
                if (hi >= lo) { core_lo[l] = lo; core_hi[l] = hi; }
            }
        }
        vp::InstDev* geoms[3] = {&in.dev, &in.dev_s, &in.dev_w};
        for (int gi = 0; gi < 3; ++gi) {
            vp::InstDev* gd = geoms[gi];
            bool reused = false;
            for (int gj = 0; gj < gi && !reused; ++gj)
                if (geoms[gj]->TP == gd->TP && geoms[gj]->ntiles == gd->ntiles) { gd->core_hint = geoms[gj]->core_hint; reused = true; }
            if (reused) continue;
            std::vector<int> tabl(vp::TILE_ORDER_AT + gd->ntiles, 0);
            std::vector<long> cost(gd->ntiles, 0);
            bool any = false;
            for (int l = 0; l < L; ++l)
                for (int t = core_lo[l] / gd->TP; core_hi[l] >= core_lo[l] && t <= std::min(gd->ntiles - 1, core_hi[l] / gd->TP); ++t) {
                    cost[t] += std::min(core_hi[l], (t + 1) * gd->TP - 1) - std::max(core_lo[l], t * gd->TP) + 1;
                    any = true;
                }
            if (any && gd->ntiles > 1) {
                std::vector<int> idx(gd->ntiles);
                for (int t = 0; t < gd->ntiles; ++t) idx[t] = t;
                std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return cost[a] > cost[b]; });
                for (int t = 0; t < gd->ntiles; ++t) tabl[vp::TILE_ORDER_AT + t] = idx[t] + 1;
            }
            // walker_kernel's deal of the tiles to its waves by that cost (WalkerArgs::wperm): waves k, k + 4, k + 8 of a workgroup
            // share a SIMD, so the tiles go out in tiers of four, every other tier backwards -- each SIMD one tile of every tier --,
            // starting behind the waves that form the records (they get the cheapest tiles).  Measured on C1, us per pass, tile
            // order / dealt: 256 walkers 15.98 / 15.68; 512 walkers (two workgroups per CU) 21.5 / 21.9 -- there two waves with
            // line cores that share a SIMD keep it issuing between them, so the deal is used for single-layer batches only
            if ((gd == &in.dev_w || (gi < 2 && in.dev_w.TP == gd->TP && in.dev_w.ntiles == gd->ntiles)) && gd->ntiles <= 16) {
                const int nw = gd->ntiles, ntask = std::min(nw, 1 + (L + 3) / 4);
                std::vector<int> idx(nw), seq(nw);
                for (int t = 0; t < nw; ++t) idx[t] = t;
                if (any) std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return cost[a] > cost[b]; });
                for (int k = 0; k < nw; ++k) seq[k] = (ntask + k) % nw;       // the waves with entry tasks come last (cheapest tiles)
                unsigned long long perm = 0xFEDCBA9876543210ull;
                for (int k = 0; k < nw; ++k) {
                    const int tier = k / 4, lo = tier * 4, hi = std::min(nw, lo + 4) - 1;
                    const int tile = idx[(tier & 1) ? hi - (k - lo) : k];
                    perm = (perm & ~(15ull << (4 * seq[k]))) | ((unsigned long long)tile << (4 * seq[k]));
                }
                in.wperm = perm;
            }
            int* d_tabl;
            if ((rc = upload<int>(c, &in, tabl.data(), tabl.size(), &d_tabl))) { for (void* p : in.allocs) hipFree(p); return rc; }
            gd->core_hint = d_tabl;
        }
    }
    in.h_lambda0.assign(lambda0, lambda0 + L); in.h_gamma.assign(gamma, gamma + L); in.h_bidx.assign(b_idx, b_idx + L);
    in.h_lines.assign(lambda0, lambda0 + L); in.h_lines.insert(in.h_lines.end(), gamma, gamma + L);
    in.h_lines.insert(in.h_lines.end(), f, f + L); in.h_lines.insert(in.h_lines.end(), zfac, zfac + L);
    in.h_idx.assign(N_idx, N_idx + L); in.h_idx.insert(in.h_idx.end(), b_idx, b_idx + L); in.h_idx.insert(in.h_idx.end(), v_idx, v_idx + L);
    in.h_idx.push_back(voigt_method); in.h_idx.push_back(c->tune.no_multipole); in.h_idx.push_back(c->tune.multipole_min);
    if (!c->inst.empty()) {
        const Instrument& pv = c->inst.back();
        in.same_lines_as_prev = pv.h_idx == in.h_idx && pv.h_lines.size() == in.h_lines.size() &&
                                std::memcmp(pv.h_lines.data(), in.h_lines.data(), in.h_lines.size() * sizeof(double)) == 0;
    }
    analyse_generic(c, in);
    if (c->tune.lds_pad > 0) in.lds_bytes += (size_t)c->tune.lds_pad;   // occupancy experiments
    c->inst.push_back(std::move(in));
    c->meta_dirty = true;
    if (inst_index) *inst_index = (int)c->inst.size() - 1;
    return VP_OK;
}

int vp_update_spectrum(vp_ctx* c, int inst, const double* flux, const double* inv_sigma2, const double* log_inv_sigma2) {
    if (!c) return VP_EINVAL;
    CtxGuard g(c);
    if (inst < 0 || inst >= (int)c->inst.size()) return fail(c, VP_EINVAL, "vp_update_spectrum: instrument index out of range");
    if (!flux || !inv_sigma2 || !log_inv_sigma2) return fail(c, VP_EINVAL, "vp_update_spectrum: NULL array");
    Instrument& in = c->inst[inst];
    if (carries_sentinel(flux, in.dev.P) || carries_sentinel(inv_sigma2, in.dev.P)) c->sentinel_unsafe = true;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipDeviceSynchronize());
    HIP_TRY(c, hipMemcpy(in.d_flux, flux, (size_t)in.dev.P * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(in.d_w, inv_sigma2, (size_t)in.dev.P * sizeof(double), hipMemcpyHostToDevice));
    in.sum_logw = neumaier_sum(log_inv_sigma2, in.dev.P);
    c->meta_dirty = true;
    return VP_OK;
}

int vp_lnprob_batch_device(vp_ctx* c, int W, int D, const double* d_theta, double* d_out, void* hip_stream) {
    if (!c) return VP_EINVAL;
    CtxGuard g(c);
    int rc = check_batch_args(c, W, D, d_theta, d_out);
    if (rc) return rc;
    if (W == 0) return VP_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    if ((rc = ensure_workspace(c, W))) return rc;
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
    if ((rc = foreign_stream_fence(c, s))) return rc;
    return enqueue_lnprob(c, W, d_theta, d_out, s);
}

// ---- direct-write gather (the per-pass exchange of the walker sharding) ---------------------------------
// One process per GPU: every rank owns a (world, W) vector and the flags of its peers' passes; the memory is exported with
// hipIpcGetMemHandle, the handles travel through the job's own channel (torch.distributed's object all-gather in
// rbvfit_amd.dist.DirectGather) and every rank maps its peers' vectors.  A pass is then ONE launch: walker_kernel stores each
// walker's lnprob into its block of every rank's vector (8 bytes per walker and rank, through the mapped pointers); the next
// launch on the stream raises this rank's flag in every peer -- the pass before is complete when it starts -- and its workgroups
// wait for their peers' flags of that pass before they compute (vp::replicas_handshake): no collective launch, no host
// involvement, nothing counted at the end of a launch.
namespace {
__global__ void gather_wait_kernel(vp::Replicas R) { vp::replicas_handshake(R); }
// Ranks that share a GPU (vp_gather_connect's shared_device): a pass is [wait for the peers' flags of the pass before] [the pass's
// launches] [publish: raise this rank's flag of THIS pass in every peer].  A wait then only ever depends on launches that were
// enqueued before it -- whatever hardware queue the runtime put them on (several streams of one process can share a queue, and a
// launch that waits for one queued BEHIND it there would wait for ever; with the flag raised by the next pass's first launch,
// as ranks on GPUs of their own do it, eight ranks in one process did exactly that).
__global__ void gather_publish_kernel(vp::Replicas R) {
    const int lane = threadIdx.x & 63;
    if (lane < R.n && lane != R.me) __hip_atomic_store(R.flags[lane] + R.me, R.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void gather_waitonly_kernel(vp::Replicas R) {       // until every peer has published pass R.seq - 1
    const int lane = threadIdx.x & 63, need = R.seq - 1;
    if (need <= 0 || R.n <= 1) return;
    for (int spins = 0;; ++spins) {
        int f = need;
        if (lane < R.n && lane != R.me) f = __hip_atomic_load(R.flags[R.me] + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (__ballot(f < need) == 0ull) break;
        if (spins > vp::SYNC_SPIN_LIMIT) {
            if (lane == 0) __hip_atomic_store(R.timeout, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        __builtin_amdgcn_s_sleep(8);
    }
}
}

static vp::Replicas gather_replicas(const vp_ctx* c, int seq) {
    const vp_ctx::Gather& g = c->gather;
    vp::Replicas R{};
    R.n = g.world;
    for (int r = 0; r < g.world; ++r) {
        R.pos[r] = nullptr;
        R.lp[r] = (r == g.rank ? g.buf : g.peer_buf[r]) + (size_t)g.rank * g.W;
        R.flags[r] = r == g.rank ? g.flags : g.peer_flags[r];
    }
    R.done = g.done;
    R.timeout = reinterpret_cast<int*>(g.done + vp::PUB_GROUPS + 1);
    R.me = g.rank;
    R.seq = seq;
    R.sync = g.world > 1 ? 2 : 1;
    return R;
}

int vp_gather_create(vp_ctx* c, int W, int world, int rank, void* handles_out) {
    if (!c) return VP_EINVAL;
    CtxGuard g(c);
    if (W <= 0 || world <= 0 || world > vp::MAX_REPLICAS || rank < 0 || rank >= world || !handles_out)
        return fail(c, VP_EINVAL, "vp_gather_create: W > 0, 1 <= world <= " + std::to_string(vp::MAX_REPLICAS) + ", 0 <= rank < world and a handle buffer required");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipDeviceSynchronize());
    gather_release(c);
    vp_ctx::Gather& G = c->gather;
    G.W = W; G.world = world; G.rank = rank;
    const size_t bb = (size_t)world * W * sizeof(double), fb = 64 * sizeof(int);
    // fine-grained device memory where the runtime offers it (peers write while kernels here read), plain otherwise
    if (!c->tune.gather_plain && hipExtMallocWithFlags((void**)&G.buf, bb, hipDeviceMallocFinegrained) == hipSuccess &&
        hipExtMallocWithFlags((void**)&G.flags, fb, hipDeviceMallocFinegrained) == hipSuccess) {
        G.finegrained = true;
    } else {
        (void)hipGetLastError();
        if (G.buf) { (void)hipFree(G.buf); G.buf = nullptr; }
        if (G.flags) { (void)hipFree(G.flags); G.flags = nullptr; }
        HIP_TRY(c, hipMalloc((void**)&G.buf, bb));
        HIP_TRY(c, hipMalloc((void**)&G.flags, fb));
    }
    HIP_TRY(c, hipMalloc((void**)&G.done, 128));
    HIP_TRY(c, hipMemset(G.buf, 0, bb));
    HIP_TRY(c, hipMemset(G.flags, 0, fb));
    HIP_TRY(c, hipMemset(G.done, 0, 128));
    HIP_TRY(c, hipDeviceSynchronize());
    std::memset(handles_out, 0, 2 * sizeof(hipIpcMemHandle_t));
    if (world > 1) {
        hipIpcMemHandle_t h[2];
        hipError_t e = hipIpcGetMemHandle(&h[0], G.buf);
        if (e == hipSuccess) e = hipIpcGetMemHandle(&h[1], G.flags);
        if (e != hipSuccess && G.finegrained) {          // (fine-grained memory that cannot be exported: plain memory instead)
            (void)hipGetLastError();
            (void)hipFree(G.buf); (void)hipFree(G.flags);
            G.buf = nullptr; G.flags = nullptr; G.finegrained = false;
            HIP_TRY(c, hipMalloc((void**)&G.buf, bb));
            HIP_TRY(c, hipMalloc((void**)&G.flags, fb));
            HIP_TRY(c, hipMemset(G.buf, 0, bb));
            HIP_TRY(c, hipMemset(G.flags, 0, fb));
            HIP_TRY(c, hipDeviceSynchronize());
            e = hipIpcGetMemHandle(&h[0], G.buf);
            if (e == hipSuccess) e = hipIpcGetMemHandle(&h[1], G.flags);
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            gather_release(c);
            return fail(c, VP_EHIP, std::string("vp_gather_create: hipIpcGetMemHandle: ") + hipGetErrorString(e));
        }
        std::memcpy(handles_out, h, sizeof(h));
    } else {
        G.connected = true;
    }
    return VP_OK;
}

int vp_gather_connect(vp_ctx* c, const void* handles_all, int shared_device) {
    if (!c) return VP_EINVAL;
    CtxGuard g(c);
    vp_ctx::Gather& G = c->gather;
    if (!G.buf) return fail(c, VP_ESTATE, "vp_gather_connect: call vp_gather_create first");
    G.shared_device = shared_device != 0;
    if (G.connected) return VP_OK;
    if (!handles_all) return fail(c, VP_EINVAL, "vp_gather_connect: NULL handles");
    HIP_TRY(c, hipSetDevice(c->device));
    const hipIpcMemHandle_t* h = static_cast<const hipIpcMemHandle_t*>(handles_all);
    for (int r = 0; r < G.world; ++r) {
        if (r == G.rank) continue;
        void *pb = nullptr, *pf = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&pb, h[2 * r], hipIpcMemLazyEnablePeerAccess);
        if (e == hipSuccess) e = hipIpcOpenMemHandle(&pf, h[2 * r + 1], hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            if (pb) (void)hipIpcCloseMemHandle(pb);
            return fail(c, VP_EHIP, std::string("vp_gather_connect: hipIpcOpenMemHandle (rank ") + std::to_string(r) + "): " + hipGetErrorString(e));
        }
        G.peer_buf[r] = static_cast<double*>(pb);
        G.peer_flags[r] = static_cast<int*>(pf);
    }
    G.connected = true;
    return VP_OK;
}

int vp_gather_connect_local(vp_ctx* c, vp_ctx* const* peers, int shared_device) {
    if (!c) return VP_EINVAL;
    CtxGuard g(c);
    vp_ctx::Gather& G = c->gather;
    if (!G.buf) return fail(c, VP_ESTATE, "vp_gather_connect_local: call vp_gather_create first");
    G.shared_device = shared_device != 0;
    if (G.connected) return VP_OK;
    if (!peers) return fail(c, VP_EINVAL, "vp_gather_connect_local: NULL peers");
    HIP_TRY(c, hipSetDevice(c->device));
    for (int r = 0; r < G.world; ++r) {
        if (r == G.rank) continue;
        const vp_ctx* p = peers[r];
        // (the peer's gather is read without its lock: the caller sets all ranks up before any of them runs a pass)
        if (!p || p == c || !p->gather.buf || p->gather.W != G.W || p->gather.world != G.world || p->gather.rank != r)
            return fail(c, VP_EINVAL, "vp_gather_connect_local: peers[" + std::to_string(r) + "] has no gather of this shape with rank " + std::to_string(r));
        if (p->device != c->device) {
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, c->device, p->device) != hipSuccess || !can)
                return fail(c, VP_EHIP, "vp_gather_connect_local: device " + std::to_string(c->device) + " cannot map device " + std::to_string(p->device));
            hipError_t e = hipDeviceEnablePeerAccess(p->device, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return fail(c, VP_EHIP, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
            (void)hipGetLastError();
        }
        G.peer_buf[r] = p->gather.buf;
        G.peer_flags[r] = p->gather.flags;
    }
    G.local_peers = true;
    G.connected = true;
    return VP_OK;
}

int vp_lnprob_gather_device(vp_ctx* c, int W, int D, const double* d_theta, void* hip_stream) {
    if (!c) return VP_EINVAL;
    CtxGuard g(c);
    vp_ctx::Gather& G = c->gather;
    if (!G.connected) return fail(c, VP_ESTATE, "vp_lnprob_gather_device: no connected gather (vp_gather_create / vp_gather_connect)");
    int rc = check_batch_args(c, W, D, d_theta, G.buf);
    if (rc) return rc;
    if (W != G.W) return fail(c, VP_EINVAL, "vp_lnprob_gather_device: W differs from the gather's block size");
    HIP_TRY(c, hipSetDevice(c->device));
    if ((rc = ensure_workspace(c, W))) return rc;
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
    if ((rc = foreign_stream_fence(c, s))) return rc;
    vp::Replicas R = gather_replicas(c, ++G.seq);
    const bool shared = G.shared_device && G.world > 1;
    if (shared) {                                    // ranks that share a GPU: wait in front, publish behind (gather_publish_kernel)
        hipLaunchKernelGGL(gather_waitonly_kernel, dim3(1), dim3(64), 0, s, R);
        R.sync = 0;
    }
    c->gather_rep = &R;
    rc = enqueue_lnprob(c, W, d_theta, nullptr, s);
    c->gather_rep = nullptr;
    if (shared && rc == VP_OK) {
        hipLaunchKernelGGL(gather_publish_kernel, dim3(1), dim3(64), 0, s, gather_replicas(c, G.seq));
        HIP_TRY(c, hipGetLastError());
    }
    return rc;
}

int vp_gather_wait(vp_ctx* c, void* hip_stream) {
    if (!c) return VP_EINVAL;
    CtxGuard g(c);
    vp_ctx::Gather& G = c->gather;
    if (!G.connected) return fail(c, VP_ESTATE, "vp_gather_wait: no connected gather");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
    if (G.shared_device && G.world > 1) hipLaunchKernelGGL(gather_waitonly_kernel, dim3(1), dim3(64), 0, s, gather_replicas(c, G.seq + 1));
    else hipLaunchKernelGGL(gather_wait_kernel, dim3(1), dim3(64), 0, s, gather_replicas(c, G.seq + 1));
    HIP_TRY(c, hipGetLastError());
    return VP_OK;
}

int vp_gather_state(vp_ctx* c, double** d_gathered, int* timed_out) {
    if (!c) return VP_EINVAL;
    CtxGuard g(c);
    vp_ctx::Gather& G = c->gather;
    if (!G.buf) return fail(c, VP_ESTATE, "vp_gather_state: no gather");
    if (d_gathered) *d_gathered = G.buf;
    if (timed_out) {
        HIP_TRY(c, hipSetDevice(c->device));
        int t = 0;
        HIP_TRY(c, hipMemcpy(&t, G.done + vp::PUB_GROUPS + 1, sizeof(int), hipMemcpyDeviceToHost));     // (synchronises with the stream's work)
        *timed_out = t;
    }
    return VP_OK;
}

int vp_gather_destroy(vp_ctx* c) {
    if (!c) return VP_EINVAL;
    CtxGuard g(c);
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    gather_release(c);
    return VP_OK;
}

// (called with c->mu held) host-buffer lnprob in two halves, so that several contexts (vp_multi) can have their
// batches in flight at once: begin = stage theta + enqueue, end = wait + copy out
static int lnprob_host_begin(vp_ctx* c, int W, int D, const double* theta, int arm_next = 0) {
    int rc;
    const size_t tb = (size_t)W * D * sizeof(double), ob = (size_t)W * sizeof(double);
    // a launch waiting for exactly this shape of batch?  (anything else: it leaves before the workspace may move)
    bool use_armed = false;
    if (c->arm.live) {
        if (c->arm.W == W && arm_next >= 0 && prearm_eligible(c, W, tb) &&
            __atomic_load_n(&c->arm.h[vp::ARM_EXPIRED_WORD], __ATOMIC_ACQUIRE) != c->arm.seq)
            use_armed = true;
        else {
            if (__atomic_load_n(&c->arm.h[vp::ARM_EXPIRED_WORD], __ATOMIC_ACQUIRE) == c->arm.seq) {
                // it waited its whole budget for nothing -- the GPU was held for nobody: leave this caller alone for a while
                c->arm.live = false; ++c->arm.expired;
                c->arm.cooldown = c->arm.cooldown_next;
                c->arm.cooldown_next = std::min(4096, 2 * c->arm.cooldown_next);
                c->arm.used_streak = 0;
            }
            prearm_cancel(c);
        }
    }
    c->arm.inflight = false;
    HIP_TRY(c, hipSetDevice(c->device));
    if ((rc = ensure_workspace(c, W))) return rc;
    if ((rc = ensure_pinned(c, tb + ob))) return rc;
    if (!use_armed) std::memcpy(c->h_pinned, theta, tb);       // (a waiting launch gets theta pushed into its slots instead)
    double* h_out = c->h_pinned + (size_t)W * D;
    VP_HSTAMP(0);
    c->done_armed = false;
    c->sentinel_armed = false;
    // zerocopy_max defaults to 1 MiB; measured: 512 walkers x 6 parameters 50 us/call zero-copy vs 58 us with copies
    if (tb <= (size_t)std::max(0l, c->tune.zerocopy_max) && !c->tune.no_zerocopy) {
        // up to 1 MiB of theta: the kernels read theta from / write lnprob to the pinned (host-coherent)
        // buffer directly over PCIe -- no H2D/D2H copy commands on the latency path
        if (!c->h_pinned_dev) HIP_TRY(c, hipHostGetDevicePointer((void**)&c->h_pinned_dev, c->h_pinned, 0));
        double* dp = c->h_pinned_dev;
        // completion by the output rows themselves: every row is written exactly once per batch (out-of-bounds rows by the
        // launch that applies the prior, the others by the launch -- or the walker's last tile -- that finishes them)
        const bool poll_rows = c->tune.host_spin >= 2 && !c->sentinel_unsafe && !c->gather_rep && !carries_sentinel(theta, (size_t)W * D);
        if (use_armed && !poll_rows) { prearm_cancel(c); use_armed = false; std::memcpy(c->h_pinned, theta, tb); }
        if (poll_rows) {
            uint64_t* o = reinterpret_cast<uint64_t*>(h_out);
            for (int i = 0; i < W; ++i) o[i] = VP_SENTINEL_BITS;
            __atomic_thread_fence(__ATOMIC_RELEASE);
        }
        if (use_armed) {
            // the pattern rows are in place: push theta and the go words into the waiting launch's slots
            prearm_push(c, W, theta, vp::ARM_GO);
            c->arm.theta_src = theta;
            c->arm.live = false;
            c->arm.inflight = true;
            c->arm.cur = c->arm.live_stream;
            c->arm.seq_inflight = c->arm.seq;
            ++c->arm.used;
            if (++c->arm.used_streak >= 16) c->arm.cooldown_next = 8;
            c->last_kind = 1;
            c->last_ff = vp_ctx::LastFF{};
        } else {
            c->arm.cur = c->stream;
            if ((rc = enqueue_lnprob(c, W, dp, dp + (size_t)W * D, c->stream))) return rc;
        }
        if (poll_rows) {
            // the launch for the caller's NEXT batch of this shape, behind this one on the stream
            if (arm_next > 0 && prearm_eligible(c, W, tb) && (rc = prearm_launch(c, W))) return rc;
            c->sentinel_armed = true; c->sentinel_W = W; c->sentinel_out = h_out;
            VP_HSTAMP(1);
            return VP_OK;
        }
    } else {
        HIP_TRY(c, hipMemcpyAsync(c->d_theta, c->h_pinned, tb, hipMemcpyHostToDevice, c->stream));
        if ((rc = enqueue_lnprob(c, W, c->d_theta, c->d_out, c->stream))) return rc;
        HIP_TRY(c, hipMemcpyAsync(h_out, c->d_out, ob, hipMemcpyDeviceToHost, c->stream));
    }
    // completion word behind the batch (the command processor writes it once everything before it on the stream is done
    // and visible to the host); where the runtime refuses, host_wait falls back to hipStreamSynchronize
    if (c->tune.host_spin) {
        if (!c->h_done) {
            if (hipHostMalloc((void**)&c->h_done, 64, hipHostMallocMapped) != hipSuccess) { c->h_done = nullptr; (void)hipGetLastError(); }
            else *c->h_done = 0;
        }
        void* dptr = nullptr;
        if (c->h_done && hipHostGetDevicePointer(&dptr, c->h_done, 0) == hipSuccess &&
            hipStreamWriteValue32(c->stream, dptr, ++c->done_seq, 0) == hipSuccess)
            c->done_armed = true;
        else
            (void)hipGetLastError();
    }
    VP_HSTAMP(1);
    return VP_OK;
}
// wait for the batch lnprob_host_begin enqueued: spin on the completion word (no interrupt, no driver call on the way),
// hipStreamSynchronize when the word is not armed or has not come after 20 ms (a long batch: the interrupt is cheap then)
static int host_wait(vp_ctx* c) {
    if (c->sentinel_armed) {
        // rows land in any order; walk them once, waiting at the first one that is still the pattern
        c->sentinel_armed = false;
        const uint64_t* o = reinterpret_cast<const uint64_t*>(c->sentinel_out);
        const int W = c->sentinel_W;
        const auto t0 = std::chrono::steady_clock::now();
        int i = 0;
        // (a batch handed to a pre-armed launch: the launch may have given up waiting just as the rows and go words were on their way --
        //  it says so; some of its workgroups may have met their rows and run.  The batch is then started again the ordinary way:
        //  the launch armed behind it is sent away, the stream drains, the rows are set to the pattern again)
        auto relaunch = [&]() -> int {
            c->arm.inflight = false;
            prearm_cancel(c);
            HIP_TRY(c, hipSetDevice(c->device));
            // some workgroups of the expired launch may have met their rows and run: nothing of it may still write when the
            // rows are set to the pattern again
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            // (the split form counts a walker's workgroups in: groups of the expired launch that ran have left their tickets)
            if (c->d_ticket && c->capW > 0) HIP_TRY(c, hipMemset(c->d_ticket, 0, (size_t)c->capW * sizeof(unsigned int)));
            std::memcpy(c->h_pinned, c->arm.theta_src, (size_t)W * c->D * sizeof(double));
            uint64_t* orow = const_cast<uint64_t*>(o);
            for (int k = 0; k < W; ++k) orow[k] = VP_SENTINEL_BITS;
            __atomic_thread_fence(__ATOMIC_RELEASE);
            i = 0;
            c->arm.cur = c->stream;
            double* dp = c->h_pinned_dev;
            return enqueue_lnprob(c, W, dp, dp + (size_t)W * c->D, c->stream);
        };
        for (unsigned int spins = 0; i < W; ++spins) {
            while (i < W && __atomic_load_n(o + i, __ATOMIC_ACQUIRE) != VP_SENTINEL_BITS) {
#ifdef VP_STAMPS
                if (i == 0) VP_HSTAMP(2);
#endif
                ++i;
            }
            if (i == W) { VP_HSTAMP(3); if (c->arm.inflight) c->arm.misses = 0; return VP_OK; }
            __builtin_ia32_pause();
            if (c->arm.inflight && (spins & 63u) == 63u &&
                (__atomic_load_n(&c->arm.h[vp::ARM_EXPIRED_WORD], __ATOMIC_ACQUIRE) == c->arm.seq_inflight ||
                 __atomic_load_n(&c->arm.h[vp::ARM_STUCK_WORD], __ATOMIC_ACQUIRE) == c->arm.seq_inflight)) {      // (a workgroup gave up on its own: same cure)
                ++c->arm.expired; --c->arm.used;
                // (the go words were on their way and the launch still gave up: once is a caller that arrived at the last moment;
                //  three times in a row is a system where the pushes do not reach the launch in time -- no more pre-armed launches)
                if (++c->arm.misses >= 3) c->arm.bar = 0;
                int rc = relaunch();
                if (rc) return rc;
            }
            if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) break;
        }
        if (i == W) return VP_OK;
        HIP_TRY(c, hipSetDevice(c->device));
        HIP_TRY(c, hipStreamSynchronize(c->arm.inflight && c->arm.cur ? c->arm.cur : c->stream));     // (a long batch: the interrupt is cheap then; everything is written behind it)
        if (c->arm.inflight) {
            // ... unless workgroups of a pre-armed launch left on their own (arm_wait's last resort): rows still carry the pattern
            bool missing = false;
            for (int k = 0; k < W; ++k) missing |= __atomic_load_n(o + k, __ATOMIC_ACQUIRE) == VP_SENTINEL_BITS;
            if (missing) {
                c->arm.bar = 0;
                int rc = relaunch();
                if (rc) return rc;
                HIP_TRY(c, hipStreamSynchronize(c->stream));
            }
        }
        return VP_OK;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->done_armed) {
        const uint32_t want = c->done_seq;
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned int spins = 0;; ++spins) {
            if (__atomic_load_n(c->h_done, __ATOMIC_ACQUIRE) == want) { VP_HSTAMP(3); return VP_OK; }
            __builtin_ia32_pause();
            if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) break;
        }
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return VP_OK;
}
static int lnprob_host_end(vp_ctx* c, int W, int D, double* out) {
    int rc;
    if ((rc = host_wait(c))) return rc;
    std::memcpy(out, c->h_pinned + (size_t)W * D, (size_t)W * sizeof(double));
    VP_HSTAMP(4);
    return VP_OK;
}

int vp_lnprob_batch(vp_ctx* c, int W, int D, const double* theta, double* out) {
    if (!c) return VP_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    int rc = check_batch_args(c, W, D, theta, out);
    if (rc) return rc;
    if (W == 0) return VP_OK;
#ifdef VP_STAMPS
    g_host_t0 = std::chrono::steady_clock::now();
#endif
    prearm_cancel_others(c);
    // pre-arm the next call's launch?  (prearm = -1: when this call came quickly behind the last one's return -- a sampler's loop)
    int arm_next = 0;
    if (c->tune.prearm > 0) arm_next = 1;
    else if (c->tune.prearm < 0 && c->arm.have_last && c->arm.last_W == W) {    // (ragged batches -- a slice sampler's rounds -- never arm)
        const double gap = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - c->arm.last_return).count();
        if (gap < 0.5 * std::max(1, c->tune.prearm_us)) {
            c->arm.gap_ema_us = c->arm.gap_ema_us > 0.0 ? 0.75 * c->arm.gap_ema_us + 0.25 * gap : gap;
            arm_next = 1;
        } else {
            c->arm.gap_ema_us = 0.0;             // a pause: the rhythm starts again
        }
        if (arm_next && c->arm.cooldown > 0) { --c->arm.cooldown; arm_next = 0; }
    }
    c->arm.last_W = W;
    if ((rc = lnprob_host_begin(c, W, D, theta, arm_next))) return rc;
    rc = lnprob_host_end(c, W, D, out);
    c->arm.inflight = false;
    c->arm.last_return = std::chrono::steady_clock::now();
    c->arm.have_last = true;
    return rc;
}
int vp_prearm_counts(vp_ctx* c, int64_t* used, int64_t* expired, int64_t* cancelled) {
    if (!c) return VP_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    if (used) *used = c->arm.used;
    if (expired) *expired = c->arm.expired;
    if (cancelled) *cancelled = c->arm.cancelled;
    return VP_OK;
}

// (called with c->mu held) prep + tile launches that write the (W, P) model flux of one instrument
static int enqueue_model_flux(vp_ctx* c, int inst, int W, const double* d_theta, double* d_out, int convolved, hipStream_t s) {
    int rc;
    if ((rc = ensure_workspace(c, W))) return rc;
    const Instrument& in = c->inst[inst];
    const bool gen = in.dev.method == VP_VOIGT_WOFZ;      // model_flux has no prior box: theta may be anything
    GenFlags gfl{nullptr, nullptr, nullptr, nullptr, 0};
    if (gen && (rc = genflag_acquire(c, W, s, &gfl))) return rc;
    const vp::FinalizeArgs nofin{};
    const int* gf = gfl.use;
    // far lines from the blocks' expansions, as in the lnprob launches (convolved flux; same rule for when the extra launch pays)
    double* ff = nullptr;
    c->last_ff = vp_ctx::LastFF{};
    if (convolved && in.ff_on && c->d_ff && c->tune.flux_farfield != 0) {
        const double score = in.ff_cover * (double)W * in.dev.ntiles * in.dev.ff_nblk * in.ff_items;
        if (c->tune.flux_farfield > 0 || score >= (in.dev.ff_members ? 3.0e5 : 1.5e5)) ff = c->d_ff;
    }
    // Batches the walker kernel takes as lnprob batches (one instrument, no cluster records, the workgroups fit the CUs): the rows in
    // ONE launch, workgroup = walker -- records formed in the workgroup, no record-preparation launch in front (C1 at 512 rows
    // 28 -> 24 us); walkers with a line outside the fast domain are left to the generic launch behind it, as in the tile path.
    if (convolved && !ff && inst == 0 && c->inst.size() == 1 && in.dev.NCm == 0 && c->tune.flux_walker != 0 && !c->profiling &&
        !c->gather_rep && walker_applies(c, W) && in.dev.method == VP_VOIGT_WOFZ) {
        vp::WalkerArgs a{d_theta, c->d_lb, c->d_ub, c->d_lc, d_out, 0.0, c->D, (int)(walker_wave_lds(c) / sizeof(double)),
                         walker_prio_for(c, W), walker_perm_for(c, W)};
        a.flux_stride = in.dev.P;
        a.genflag = gfl.use; a.genflag_clear = gfl.clear; a.gen_any = gfl.any; a.gen_any_clear = gfl.any_clear;
        vp::InstDev d0 = in.dev_w;
        vp::LinesDev t0 = in.lines;
        d0.NCm = 0; t0.NCm = 0;
        hipLaunchKernelGGL((vp::walker_kernel<0, false, false, false, 1>), dim3(W), dim3(64 * walker_tiles(c)), walker_lds_bytes(c), s, d0, t0, a,
                           vp::StretchArgs{});
        launch_tile<1, true>(in, c->d_lc, nullptr, d_out, in.dev.P, 0, W, s, nofin, gf, nullptr, 1, nullptr, gfl.slots, d_theta, c->D);
        HIP_TRY(c, hipGetLastError());
        return VP_OK;
    }
    launch_prep(c, in, d_theta, W, 0, nullptr, gfl.use, s, gfl.clear, gfl.any, gfl.any_clear);
    if (ff) {
        vp::InstDev g2 = in.dev;
        g2.ff = ff;
        const int nbk = g2.ntiles * g2.ff_nblk;
        c->last_ff = vp_ctx::LastFF{ff, W, nbk, g2.ff_members, inst};
        const size_t ffl = vp::farfield_lds_bytes(in.lines.L, in.lines.NCm);
        if (g2.ff_members) hipLaunchKernelGGL((vp::farfield_kernel<9, true>), dim3((nbk + 63) / 64, W), dim3(64 * vp::FF_WAVES), ffl, s, g2, in.lines, c->d_lc, W);
        else hipLaunchKernelGGL((vp::farfield_kernel<6, false>), dim3((nbk + 63) / 64, W), dim3(64 * vp::FF_WAVES), ffl, s, g2, in.lines, c->d_lc, W);
    }
    if (convolved) {
        launch_tile<1, false>(in, c->d_lc, nullptr, d_out, in.dev.P, 0, W, s, nofin, gf, nullptr, 1, ff);
        if (gen) launch_tile<1, true>(in, c->d_lc, nullptr, d_out, in.dev.P, 0, W, s, nofin, gf, nullptr, 1, nullptr, gfl.slots);
    } else {
        launch_tile<2, false>(in, c->d_lc, nullptr, d_out, in.dev.P, 0, W, s, nofin, gf);
        if (gen) launch_tile<2, true>(in, c->d_lc, nullptr, d_out, in.dev.P, 0, W, s, nofin, gf, nullptr, 1, nullptr, gfl.slots);
    }
    HIP_TRY(c, hipGetLastError());
    return VP_OK;
}

int vp_model_flux_batch_device(vp_ctx* c, int inst, int W, int D, const double* d_theta, double* d_out, int convolved,
                               void* hip_stream) {
    if (!c) return VP_EINVAL;
    CtxGuard g(c);
    int rc = check_batch_args(c, W, D, d_theta, d_out);
    if (rc) return rc;
    if (inst < 0 || inst >= (int)c->inst.size()) return fail(c, VP_EINVAL, "vp_model_flux_batch: instrument index out of range");
    if (W == 0) return VP_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
    if ((rc = foreign_stream_fence(c, s))) return rc;
    return enqueue_model_flux(c, inst, W, d_theta, d_out, convolved, s);
}

int vp_model_flux_batch(vp_ctx* c, int inst, int W, int D, const double* theta, double* out, int convolved) {
    if (!c) return VP_EINVAL;
    CtxGuard g(c);          // held from staging theta to the copy-back: d_theta / d_scratch are the context's
    int rc = check_batch_args(c, W, D, theta, out);
    if (rc) return rc;
    if (inst < 0 || inst >= (int)c->inst.size()) return fail(c, VP_EINVAL, "vp_model_flux_batch: instrument index out of range");
    if (W == 0) return VP_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    if ((rc = ensure_workspace(c, W))) return rc;
    const size_t P = c->inst[inst].dev.P;
    if ((rc = ensure_scratch(c, (size_t)W * P * sizeof(double)))) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->d_theta, theta, (size_t)W * D * sizeof(double), hipMemcpyHostToDevice, c->stream));
    if ((rc = enqueue_model_flux(c, inst, W, c->d_theta, c->d_scratch, convolved, c->stream))) return rc;
    HIP_TRY(c, hipMemcpyAsync(out, c->d_scratch, (size_t)W * P * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return VP_OK;
}

int vp_model_flux_components(vp_ctx* c, int inst, int W, int D, const double* theta, double* out) {
    if (!c) return VP_EINVAL;
    CtxGuard g(c);
    int rc = check_batch_args(c, W, D, theta, out);
    if (rc) return rc;
    if (inst < 0 || inst >= (int)c->inst.size()) return fail(c, VP_EINVAL, "vp_model_flux_components: instrument index out of range");
    if (W == 0) return VP_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    if ((rc = ensure_workspace(c, W))) return rc;
    Instrument in = c->inst[inst];                 // local copy: line_sel = -2 makes the line a grid dimension
    in.allocs.clear();
    in.dev.line_sel = -2;
    const size_t P = in.dev.P, L = in.dev.L;
    // ONE launch per block of walkers evaluates all L per-line profiles (grid = walkers x tiles x lines) straight into
    // the (w, l, p) layout of `out`; blocks of at most ~1 GiB of output keep the device scratch bounded
    const size_t per_walker = L * P * sizeof(double);
    const int wblock = (int)std::max<size_t>(1, std::min<size_t>((size_t)W, ((size_t)1 << 30) / per_walker));
    if ((rc = ensure_scratch(c, (size_t)wblock * per_walker))) return rc;
    hipStream_t s = c->stream;
    HIP_TRY(c, hipMemcpyAsync(c->d_theta, theta, (size_t)W * D * sizeof(double), hipMemcpyHostToDevice, s));
    const bool gen = in.dev.method == VP_VOIGT_WOFZ;
    GenFlags gfl{nullptr, nullptr, nullptr, nullptr, 0};
    if (gen && (rc = genflag_acquire(c, W, s, &gfl))) return rc;
    launch_prep(c, in, c->d_theta, W, 0, nullptr, gfl.use, s, gfl.clear, gfl.any, gfl.any_clear);
    const vp::FinalizeArgs nofin{};
    const size_t nrec = (size_t)(in.dev.L + in.dev.NCm) * vp::LC_STRIDE;
    for (int w0 = 0; w0 < W; w0 += wblock) {
        const int nw = std::min(wblock, W - w0);
        const int* gf = gen ? gfl.use + w0 : (const int*)nullptr;
        launch_tile<2, false>(in, c->d_lc + (size_t)w0 * nrec, nullptr, c->d_scratch, (int)(L * P), 0, nw, s, nofin, gf, nullptr, (int)L);
        if (gen) launch_tile<2, true>(in, c->d_lc + (size_t)w0 * nrec, nullptr, c->d_scratch, (int)(L * P), 0, nw, s, nofin, gf, nullptr, (int)L,
                                      nullptr, gfl.slots);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipMemcpyAsync(out + (size_t)w0 * L * P, c->d_scratch, (size_t)nw * per_walker, hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(c, hipStreamSynchronize(s));
    return VP_OK;
}

int vp_stretch_run(vp_ctx* c, int W, int D, double* pos, double* lnprob, int have_lnprob, int nsteps, double a,
                   uint64_t seed, uint64_t step0, double* chain, double* chain_lnprob, int64_t* naccepted) {
    if (!c) return VP_EINVAL;
    CtxGuard g(c);
    int rc = check_batch_args(c, W, D, pos, lnprob);
    if (rc) return rc;
    if (W < 2 || (W & 1)) return fail(c, VP_EINVAL, "vp_stretch_run: the number of walkers must be even and >= 2");
    if (nsteps < 0 || !(a > 1.0)) return fail(c, VP_EINVAL, "vp_stretch_run: nsteps must be >= 0 and a > 1");
    if ((chain == nullptr) != (chain_lnprob == nullptr)) return fail(c, VP_EINVAL, "vp_stretch_run: chain and chain_lnprob go together");
    HIP_TRY(c, hipSetDevice(c->device));
    if ((rc = ensure_workspace(c, W))) return rc;
    hipStream_t s = c->stream;
    const int half = W / 2;
    // device state: pos (W,D) | lp (W) | prop (half,D) | lp_new (half) | zz (half) | second buffer of pos, lp (overlapped half-steps) |
    //               chain chunk | nacc (W) | nanflag, timeout | versions (W)
    //               mailbox lines 2 x W x 8 (overlapped half-steps with D <= 6: row, lnprob, version in one 64-byte line) |
    const size_t nd_state = (size_t)W * D + W + (size_t)half * D + 2 * (size_t)half + (size_t)W * D + W + 8 + 2 * (size_t)W * 8;
    const size_t row = (size_t)W * (D + 1);                       // doubles stored per step
    size_t chunk = chain ? std::max<size_t>(1, std::min<size_t>((size_t)std::max(nsteps, 1), ((size_t)256 << 20) / (row * sizeof(double)))) : 0;
    const size_t bytes = (nd_state + chunk * row) * sizeof(double) + (size_t)W * sizeof(long long) + 64 + (size_t)W * sizeof(int);
    if ((rc = ensure_scratch(c, bytes))) return rc;
    double* d_pos = c->d_scratch;
    double* d_lp = d_pos + (size_t)W * D;
    double* d_prop = d_lp + W;
    double* d_lpnew = d_prop + (size_t)half * D;
    double* d_zz = d_lpnew + half;
    double* d_pos1 = d_zz + half;                                 // the rows' second buffer (overlapped half-steps)
    double* d_lp1 = d_pos1 + (size_t)W * D;
    double* d_mail = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(d_lp1 + W) + 63) & ~(uintptr_t)63);    // 64-byte aligned lines
    double* d_chain = d_mail + 2 * (size_t)W * 8;                 // chunk * (W*D) then chunk * W
    long long* d_nacc = reinterpret_cast<long long*>(d_chain + chunk * row);
    int* d_nan = reinterpret_cast<int*>(d_nacc + W);             // [0] NaN flag, [1] a device-side wait gave up
    int* d_ver = d_nan + 16;                                      // (W) versions
    HIP_TRY(c, hipMemcpyAsync(d_pos, pos, (size_t)W * D * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemsetAsync(d_nacc, 0, (size_t)W * sizeof(long long) + 64 + (size_t)W * sizeof(int), s));
    if (have_lnprob) {
        for (int w = 0; w < W; ++w)
            if (lnprob[w] != lnprob[w]) return fail(c, VP_ENAN, "vp_stretch_run: the initial lnprob holds NaN (Probability function returned NaN)");
        HIP_TRY(c, hipMemcpyAsync(d_lp, lnprob, (size_t)W * sizeof(double), hipMemcpyHostToDevice, s));
    } else {
        // a walker that starts at NaN would never move (log u < NaN is false) and the run would still report success
        if ((rc = enqueue_lnprob(c, W, d_pos, d_lp, s))) return rc;
        hipLaunchKernelGGL(vp::nan_flag_kernel, dim3((W + 255) / 256), dim3(256), 0, s, d_lp, W, d_nan);
        int h_nan0 = 0;
        HIP_TRY(c, hipMemcpyAsync(&h_nan0, d_nan, sizeof(int), hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(s));
        if (h_nan0) return fail(c, VP_ENAN, "vp_stretch_run: the initial lnprob holds NaN (Probability function returned NaN)");
    }
    const int thr = 64;
    // whole half-step in one launch (proposal, lnprob, accept inside each walker's workgroup) where the walker kernel
    // applies to a half-ensemble batch and the instrument has no cluster records
    const bool one_launch = !c->tune.no_fused_accept && c->tune.walker != 0 && walker_applies(c, half) &&
                            (c->inst[0].dev.NCm == 0 || !c->tune.walker_clusters);
    const int split = one_launch ? walker_split_for(c, half) : 0;      // a walker of the half-step as several workgroups (WalkerArgs::split)
    const bool fuse = W <= 1024 && !c->tune.no_fused_accept;   // accept + next proposal in one launch
    const int wthr = ((W + 63) / 64) * 64;
    bool have_prop = false;                                   // is the proposal of the coming pass already enqueued?
    // Overlapped half-steps: consecutive half-steps on two streams, each walker waiting for its partner's version word only
    // (StretchArgs::ovl).  Both launches must be able to sit on the CUs together -- a workgroup that waits holds its slot, and
    // the one it waits for must never be left without one: two half-ensembles of workgroups within what the CUs hold at once.
    bool ovl = false;
    if (one_launch && c->tune.stretch_overlap != 0 && nsteps > 0) {
        if (c->num_cus == 0) {
            int n = 0;
            if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, c->device) == hipSuccess) c->num_cus = n;
        }
        const int nt = split > 0 ? split_shape(c, split).waves : walker_tiles(c);
        const size_t wl = split > 0 ? split_shape(c, split).lds : walker_lds_bytes(c);
        const long per_cu = std::max(1, std::min((split > 0 ? 28 : 24) / std::max(1, nt), (int)(c->lds_limit / wl)));
        ovl = c->tune.stretch_overlap > 0 || (per_cu >= 2 && 2l * half * std::max(1, split) <= per_cu * (long)c->num_cus);
        if (ovl && !c->stream2) {
            if (hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess ||
                hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); ovl = false; }
        }
    }
    // mailbox lines (StretchArgs::ovl == 2): row, lnprob and version of a walker in one 64-byte line per buffer, packed here
    const bool mail = ovl && D <= 6 && c->tune.stretch_mailbox != 0;
    std::vector<double> h_mail;
    if (mail) {
        h_mail.assign(2 * (size_t)W * 8, 0.0);
        std::vector<double> lp0(W);
        if (have_lnprob) std::memcpy(lp0.data(), lnprob, (size_t)W * sizeof(double));
        else {
            HIP_TRY(c, hipMemcpyAsync(lp0.data(), d_lp, (size_t)W * sizeof(double), hipMemcpyDeviceToHost, s));
            HIP_TRY(c, hipStreamSynchronize(s));
        }
        for (int w = 0; w < W; ++w) {
            std::memcpy(&h_mail[(size_t)w * 8], pos + (size_t)w * D, (size_t)D * sizeof(double));
            h_mail[(size_t)w * 8 + 6] = lp0[w];                    // ([7]: version 0 = all-zero bits)
        }
        HIP_TRY(c, hipMemcpyAsync(d_mail, h_mail.data(), h_mail.size() * sizeof(double), hipMemcpyHostToDevice, s));
    }
    hipStream_t s2 = ovl ? c->stream2 : s;
    // (whatever way this call ends -- an error return from the middle of the loop included -- nothing of it is left running on the
    //  second stream when the context's mutex is released)
    struct DrainSecond { hipStream_t q; bool on; ~DrainSecond() { if (on) (void)hipStreamSynchronize(q); } } drain_second{s2, ovl};
    if (ovl) {      // what the first stream has set up (rows, versions) before the second one's first launch
        HIP_TRY(c, hipEventRecord(c->ev_fork, s));
        HIP_TRY(c, hipStreamWaitEvent(s2, c->ev_fork, 0));
    }
    for (int done = 0; done < nsteps;) {
        const int n = chain ? (int)std::min<size_t>(chunk, (size_t)(nsteps - done)) : nsteps - done;
        for (int it = 0; it < n; ++it) {
            const uint64_t step = step0 + (uint64_t)(done + it);
            for (int h = 0; h < 2; ++h) {
                const int s0 = h ? half : 0, c0 = h ? 0 : half;
                if (one_launch) {
                    vp::StretchArgs sa{};
                    sa.pos = d_pos; sa.lp = d_lp; sa.nacc = d_nacc; sa.nanflag = d_nan;
                    sa.chain_pos = chain ? d_chain + (size_t)it * W * D : (double*)nullptr;
                    sa.chain_lp = chain ? d_chain + chunk * (size_t)W * D + (size_t)it * W : (double*)nullptr;
                    sa.a = a; sa.seed = seed; sa.step = step; sa.s0 = s0; sa.c0 = c0; sa.nC = half; sa.half = h;
                    if (ovl) {
                        // every walker has been updated k = done + it times when this step begins (the first half once more when
                        // h = 1): rows live in buffer (update count) & 1
                        const int k = done + it, t = 2 * k + h;
                        double* pb[2] = {mail ? d_mail : d_pos, mail ? d_mail + (size_t)W * 8 : d_pos1};
                        double* lb2[2] = {d_lp, d_lp1};
                        sa.ovl = mail ? 2 : 1; sa.need = t; sa.mine = t + 1; sa.ver = d_ver; sa.timeout = d_nan + 1;
                        sa.pos_x = pb[k & 1]; sa.lp_x = lb2[k & 1];
                        sa.pos_w = pb[(k + 1) & 1]; sa.lp_w = lb2[(k + 1) & 1];
                        sa.pos_c = pb[(h ? k + 1 : k) & 1];
                        launch_walker_stretch(c, half, sa, h ? s2 : s, h ? half : 0, split);
                    } else {
                        launch_walker_stretch(c, half, sa, s, 0, split);
                    }
                    continue;
                }
                if (!have_prop)
                    hipLaunchKernelGGL(vp::stretch_propose_kernel, dim3((half + thr - 1) / thr), dim3(thr), 0, s, d_pos, D, s0,
                                       half, c0, half, a, seed, step, h, d_prop, d_zz);
                if ((rc = enqueue_lnprob(c, half, d_prop, d_lpnew, s))) return rc;
                const bool store = chain && h == 1;
                double* cp = store ? d_chain + (size_t)it * W * D : (double*)nullptr;
                double* cl = store ? d_chain + chunk * (size_t)W * D + (size_t)it * W : (double*)nullptr;
                const bool last = (h == 1) && (done + it + 1 == nsteps);
                if (fuse && !last) {
                    vp::NextProposal nx;
                    nx.half = 1 - h; nx.s0 = nx.half ? half : 0; nx.c0 = nx.half ? 0 : half; nx.nS = half; nx.nC = half;
                    nx.step = h ? step + 1 : step;
                    hipLaunchKernelGGL(vp::stretch_accept_propose_kernel, dim3(1), dim3(wthr), 0, s, d_pos, d_lp, d_prop,
                                       d_lpnew, d_zz, W, D, s0, half, seed, step, h, d_nacc, d_nan, cp, cl, a, nx);
                    have_prop = true;
                } else {
                    hipLaunchKernelGGL(vp::stretch_accept_kernel, dim3((W + thr - 1) / thr), dim3(thr), 0, s, d_pos, d_lp, d_prop,
                                       d_lpnew, d_zz, W, D, s0, half, seed, step, h, d_nacc, d_nan, cp, cl);
                    have_prop = false;
                }
            }
        }
        HIP_TRY(c, hipGetLastError());
        if (ovl) {      // the second stream's half-steps of this chunk are done before the first stream copies anything out
            HIP_TRY(c, hipEventRecord(c->ev_join, s2));
            HIP_TRY(c, hipStreamWaitEvent(s, c->ev_join, 0));
        }
        if (chain) {
            HIP_TRY(c, hipMemcpyAsync(chain + (size_t)done * W * D, d_chain, (size_t)n * W * D * sizeof(double), hipMemcpyDeviceToHost, s));
            HIP_TRY(c, hipMemcpyAsync(chain_lnprob + (size_t)done * W, d_chain + chunk * (size_t)W * D, (size_t)n * W * sizeof(double),
                                      hipMemcpyDeviceToHost, s));
            HIP_TRY(c, hipStreamSynchronize(s));      // (the host waits: the next chunk's launches are enqueued behind the copies)
        }
        done += n;
    }
    std::vector<long long> h_nacc(W);
    int h_flags[2] = {0, 0};
    // (overlapped half-steps: after nsteps updates the rows are in buffer nsteps & 1)
    if (mail) {
        HIP_TRY(c, hipMemcpyAsync(h_mail.data(), d_mail + (size_t)(nsteps & 1) * W * 8, (size_t)W * 8 * sizeof(double), hipMemcpyDeviceToHost, s));
    } else {
        HIP_TRY(c, hipMemcpyAsync(pos, (ovl && (nsteps & 1)) ? d_pos1 : d_pos, (size_t)W * D * sizeof(double), hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipMemcpyAsync(lnprob, (ovl && (nsteps & 1)) ? d_lp1 : d_lp, (size_t)W * sizeof(double), hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(c, hipMemcpyAsync(h_nacc.data(), d_nacc, (size_t)W * sizeof(long long), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(h_flags, d_nan, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    if (mail)
        for (int w = 0; w < W; ++w) {
            std::memcpy(pos + (size_t)w * D, &h_mail[(size_t)w * 8], (size_t)D * sizeof(double));
            lnprob[w] = h_mail[(size_t)w * 8 + 6];
        }
    if (h_flags[1]) return fail(c, VP_EHIP, "vp_stretch_run: a workgroup's wait for its partner's half-step gave up (overlapped half-steps)");
    if (naccepted) for (int w = 0; w < W; ++w) naccepted[w] += (int64_t)h_nacc[w];
    if (h_flags[0]) return fail(c, VP_ENAN, "vp_stretch_run: Probability function returned NaN");
    return VP_OK;
}

// vp_slice_run on G >= 1 contexts (called with every context's mutex held).  G = 1 is the single-GPU sampler.  With G > 1
// (vp_multi_slice_run) every context holds the whole sampler state and runs the SAME control kernels on it -- begin,
// init, update, tune are deterministic functions of that state -- while each round's lnprob batch of B trial rows is cut
// into blocks of ceil(B / G) rows, one per context, evaluated with the launch structure the whole batch would get
// (policy_W), and written into EVERY replica's result vector (one double per row through peer-mapped pointers), followed
// by the event barrier.  The replicas therefore stay identical bit for bit, and equal to the single-context run.
static int multi_barrier(vp_multi* m);
namespace { __global__ void scatter_rows_kernel(const double* __restrict__ src, int n, int lo, vp::Replicas R) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const double v = src[k];
    for (int r = 0; r < R.n; ++r) R.lp[r][lo + k] = v;        // (R.lp: the replicas' row-result vectors)
} }

static int slice_run_impl(vp_multi* m, vp_ctx* const* cx, int G, int W, int D, double* pos, double* lnprob, int have_lnprob,
                          int nsteps, double* mu, int* tune, double tolerance, int patience, int maxsteps, uint64_t seed,
                          uint64_t step0, double* chain, double* chain_lnprob, double* mu_history, int64_t* n_evals, int* bad) {
    *bad = 0;
    vp_ctx* c = cx[0];
    int rc;
#define SFAIL(i, code, msg) do { *bad = (i); return fail(cx[i], (code), (msg)); } while (0)
#define STRY(i, expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) SFAIL(i, VP_EHIP, std::string(#expr) + ": " + hipGetErrorString(e__)); } while (0)
    if (W < 4 || (W & 1) || W > 2 * vp::SLICE_MAX_HALF)
        SFAIL(0, VP_EINVAL, "vp_slice_run: the number of walkers must be even, >= 4 and <= " + std::to_string(2 * vp::SLICE_MAX_HALF));
    if (nsteps < 0 || !mu || !(*mu > 0.0) || !tune || maxsteps < 1 || patience < 1 || !(tolerance >= 0.0))
        SFAIL(0, VP_EINVAL, "vp_slice_run: nsteps >= 0, mu > 0, maxsteps >= 1, patience >= 1, tolerance >= 0 and non-NULL mu/tune required");
    if ((chain == nullptr) != (chain_lnprob == nullptr)) SFAIL(0, VP_EINVAL, "vp_slice_run: chain and chain_lnprob go together");
    const int half = W / 2;
    // rows of every round's lnprob batch (the round kernel keeps one slice parameter per row in LDS: <= 4096 rows)
    const int B = std::min(2 * vp::SLICE_MAX_HALF, std::max(2, std::min(8, c->tune.slice_rows)) * half);
    const int per = (B + G - 1) / G;
    // A segment = the iterations the device works through on its own (slice_round_kernel) before the host collects the
    // chain: bounded by the chain chunk (256 MB) and by the table of random splits (64 MB).
    const size_t row = (size_t)W * (D + 1);
    size_t seg = std::min<size_t>((size_t)std::max(nsteps, 1), ((size_t)64 << 20) / ((size_t)W * sizeof(int)));
    if (chain) seg = std::min(seg, std::max<size_t>(1, ((size_t)256 << 20) / (row * sizeof(double))));
    if (c->tune.slice_seg > 0) seg = std::min(seg, (size_t)c->tune.slice_seg);
    // device state (doubles first): pos (W,D) | lp (W) | trial (2,B,D: the rounds alternate) | lnp_rows (2B) | X0, eta (half,D each) |
    // Z0, L, R (half each) | T (half, MAXC) | mu[4] | mu_hist (nsteps) | block results (per) | chain segment; then the integer state
    const size_t nd = (size_t)W * D + W + 2 * (size_t)B * D + 2 * (size_t)B + 2 * (size_t)half * D + (3 + vp::SLICE_MAXC) * (size_t)half + 4 +
                      (size_t)std::max(nsteps, 1) + (size_t)per;                // (lnp_rows twice: rounds alternate between the two when G > 1)
    const size_t ni = 7 * (size_t)half + 32 + seg * (size_t)W;                 // J K phase sides nshr row widx | counters, prog | perm table
    struct Dev { double *pos, *lp, *trial, *rows, *mu, *muhist, *blk, *chain; long long* ll; int *perm, *nact, *nan, *prog; vp::SliceState st; vp::SliceCounters cn; hipStream_t s; };
    std::vector<Dev> dv(G);
    const double h_mu[4] = {*mu, *tune > 0 ? (double)(*tune - 1) : 0.0, *tune ? 1.0 : 0.0, 0.0};   // `tune` carries the state across calls
    for (int i = 0; i < G; ++i) {
        vp_ctx* ci = cx[i];
        STRY(i, hipSetDevice(ci->device));
        if ((rc = ensure_workspace(ci, std::max(W, B)))) { *bad = i; return rc; }
        const size_t bytes = (nd + (i == 0 && chain ? seg * row : 0)) * sizeof(double) + 4 * sizeof(long long) + ni * sizeof(int) + 64;
        if ((rc = ensure_scratch(ci, bytes))) { *bad = i; return rc; }
        Dev& d = dv[i];
        d.s = ci->stream;
        d.pos = ci->d_scratch; d.lp = d.pos + (size_t)W * D; d.trial = d.lp + W; d.rows = d.trial + 2 * (size_t)B * D;
        d.st = vp::SliceState{};
        d.st.X0 = d.rows + 2 * (size_t)B; d.st.eta = d.st.X0 + (size_t)half * D; d.st.Z0 = d.st.eta + (size_t)half * D;
        d.st.L = d.st.Z0 + half; d.st.R = d.st.L + half; d.st.T = d.st.R + half;
        d.mu = d.st.T + (size_t)half * vp::SLICE_MAXC;                        // 4 doubles (3 used)
        d.muhist = d.mu + 4;
        d.blk = d.muhist + std::max(nsteps, 1);
        d.chain = d.blk + per;
        d.ll = reinterpret_cast<long long*>(d.chain + (i == 0 && chain ? seg * row : 0));     // n_evals, nexp, ncon, (pad)
        int* d_int = reinterpret_cast<int*>(d.ll + 4);
        d.st.J = d_int; d.st.K = d.st.J + half; d.st.phase = d.st.K + half; d.st.sides = d.st.phase + half; d.st.nshr = d.st.sides + half;
        d.st.row = d.st.nshr + half; d.st.widx = d.st.row + half;
        d.nact = d.st.widx + half;                                            // n_active, nanflag, ncand, (pad)
        d.nan = d.nact + 1;
        d.prog = d.nact + 8;                                                  // 8 ints (slice_kernels.h SliceRun::prog)
        d.perm = d.prog + 8;                                                  // (seg, W)
        d.cn = vp::SliceCounters{d.nact, d.ll, d.ll + 1, d.ll + 2, d.nan, d.nact + 2, d.mu};
        STRY(i, hipMemcpyAsync(d.pos, pos, (size_t)W * D * sizeof(double), hipMemcpyHostToDevice, d.s));
        STRY(i, hipMemcpyAsync(d.mu, h_mu, sizeof(h_mu), hipMemcpyHostToDevice, d.s));
        STRY(i, hipMemsetAsync(d.ll, 0, 4 * sizeof(long long), d.s));
        STRY(i, hipMemsetAsync(d.nact, 0, 16 * sizeof(int), d.s));
        ci->policy_W = 0;
        if (have_lnprob) STRY(i, hipMemcpyAsync(d.lp, lnprob, (size_t)W * sizeof(double), hipMemcpyHostToDevice, d.s));
        else if ((rc = enqueue_lnprob(ci, W, d.pos, d.lp, d.s))) { *bad = i; return rc; }     // (every replica: once per run)
    }
    {   // the start state must be finite everywhere (zeus: "Invalid walker initial positions")
        std::vector<double> h_lp(W);
        STRY(0, hipSetDevice(c->device));
        STRY(0, hipMemcpyAsync(h_lp.data(), dv[0].lp, (size_t)W * sizeof(double), hipMemcpyDeviceToHost, dv[0].s));
        STRY(0, hipStreamSynchronize(dv[0].s));
        for (int w = 0; w < W; ++w)
            if (!(std::fabs(h_lp[w]) <= 1.79e308))
                SFAIL(0, VP_ENAN, "vp_slice_run: the initial lnprob of walker " + std::to_string(w) + " is not finite");
    }
    // two words of mapped host memory that replica 0's round kernel keeps up to date: rounds consumed, done / error
    volatile int* h_words = nullptr;
    int* d_words = nullptr;
    {
        STRY(0, hipSetDevice(c->device));
        if (!c->h_done) {
            if (hipHostMalloc((void**)&c->h_done, 64, hipHostMallocMapped) != hipSuccess) { c->h_done = nullptr; (void)hipGetLastError(); }
            else std::memset(c->h_done, 0, 64);
        }
        void* dptr = nullptr;
        if (c->h_done && hipHostGetDevicePointer(&dptr, c->h_done, 0) == hipSuccess) {
            h_words = reinterpret_cast<volatile int*>(c->h_done + 8);
            d_words = reinterpret_cast<int*>(static_cast<uint32_t*>(dptr) + 8);
        } else {
            (void)hipGetLastError();
        }
    }
    int parity = 0;                                  // which of the two row-result vectors the coming round fills (G > 1)
    int tb = 0;                                      // which of the two trial batches the coming round evaluates
    auto done_ = [&](int code) { for (int i = 0; i < G; ++i) cx[i]->policy_W = 0; return code; };
    if (G > 1 && (rc = multi_barrier(m))) return done_(rc);
    vp::SliceRun P{};
    P.W = W; P.half = half; P.D = D; P.batch_rows = B; P.maxsteps = maxsteps; P.patience = patience;
    P.round_limit = (int)std::min<long long>(4ll * maxsteps + 4096, 1ll << 30);
    P.gamma0 = 2.38 / std::sqrt(2.0 * (double)D);
    P.tolerance = tolerance; P.seed = seed;
    auto launch_round = [&](int i, int start) {
        const Dev& d = dv[i];
        vp::SliceRun Pi = P;
        // (start: the first batch goes into buffer tb; a round consumes buffer tb ^ 1 -- flipped by then -- and fills tb)
        Pi.pos = d.pos; Pi.lp = d.lp; Pi.perm_tab = d.perm; Pi.prog = d.prog;
        Pi.trial = d.trial + (size_t)tb * B * D;
        Pi.trial_in = d.trial + (size_t)(tb ^ 1) * B * D;
        Pi.host = i == 0 ? d_words : nullptr;
        Pi.chain = i == 0 && chain ? d.chain : nullptr;
        Pi.chain_lp = i == 0 && chain ? d.chain + seg * (size_t)W * D : nullptr;
        Pi.mu_hist = d.muhist + (P.step_base - step0);
        hipLaunchKernelGGL(vp::slice_round_kernel, dim3(1), dim3(1024), 0, d.s, Pi, d.st, d.cn, d.rows + (G > 1 ? (size_t)(parity ^ 1) * B : 0), start);
    };
    // one round: the lnprob of the B trial rows (sharded when G > 1), then the round kernel on every replica
    auto round = [&]() -> int {
        const size_t roff = G > 1 ? (size_t)parity * B : 0;
        vp::Replicas R{};
        R.n = G;
        for (int i = 0; i < G; ++i) { R.pos[i] = nullptr; R.lp[i] = dv[i].rows + roff; }
        for (int i = 0; i < G; ++i) {
            vp_ctx* ci = cx[i];
            const Dev& d = dv[i];
            STRY(i, hipSetDevice(ci->device));
            if (G == 1) {
                if ((rc = enqueue_lnprob(ci, B, d.trial + (size_t)tb * B * D, d.rows, d.s))) { *bad = i; return rc; }
            } else {
                const int lo = std::min(i * per, B), n = std::min(lo + per, B) - lo;
                if (n <= 0) continue;
                ci->policy_W = B;
                if ((rc = enqueue_lnprob(ci, n, d.trial + ((size_t)tb * B + lo) * D, d.blk, d.s))) { *bad = i; return rc; }
                hipLaunchKernelGGL(scatter_rows_kernel, dim3((n + 255) / 256), dim3(256), 0, d.s, d.blk, n, lo, R);
            }
        }
        // every block's results are in every replica before any replica consumes them.  (The other hazard -- a block of
        // the NEXT round landing in a vector a replica's round kernel is still reading -- cannot occur: the rounds alternate
        // between two vectors, and a replica's kernel of round r precedes, on its own stream, the event it records in
        // round r + 1, which every scatter of round r + 2 waits for.)
        if (G > 1 && (rc = multi_barrier(m))) return rc;
        parity ^= 1;
        tb ^= 1;
        for (int i = 0; i < G; ++i) {
            STRY(i, hipSetDevice(cx[i]->device));
            launch_round(i, 0);
        }
        return VP_OK;
    };
    const int ahead = 8;                             // rounds the host keeps enqueued beyond the last one it has seen consumed
    for (int done = 0; done < nsteps;) {
        const int n = (int)std::min<size_t>(seg, (size_t)(nsteps - done));
        P.n = n;
        P.step_base = step0 + (uint64_t)done;
        if (h_words) { h_words[0] = 0; h_words[1] = 0; }
        for (int i = 0; i < G; ++i) {
            STRY(i, hipSetDevice(cx[i]->device));
            hipLaunchKernelGGL(vp::slice_perm_kernel, dim3(n), dim3(1024), 0, dv[i].s, W, seed, P.step_base, dv[i].perm);
            launch_round(i, 1);
        }
        int enq = 0, h_prog[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        auto t_seen = std::chrono::steady_clock::now();
        int last_seen = -1, idle = 0;
        for (;;) {
            if (h_words) {
                if (__atomic_load_n(const_cast<int*>(h_words + 1), __ATOMIC_ACQUIRE)) break;
                const int seen = __atomic_load_n(const_cast<int*>(h_words), __ATOMIC_RELAXED);
                if (enq - seen >= ahead) {           // far enough ahead: wait for the device (and notice a stream that died)
                    __builtin_ia32_pause();
                    if (seen != last_seen) { last_seen = seen; t_seen = std::chrono::steady_clock::now(); idle = 0; }
                    else if (++idle > 20000) { std::this_thread::sleep_for(std::chrono::microseconds(20)); }   // (rounds of milliseconds: big models)
                    else if (std::chrono::steady_clock::now() - t_seen > std::chrono::seconds(5)) {
                        STRY(0, hipSetDevice(c->device));
                        const hipError_t q = hipStreamQuery(dv[0].s);
                        if (q != hipErrorNotReady) {     // the queue ran dry (or failed) without the word moving
                            if (q != hipSuccess) { (void)hipGetLastError(); *bad = 0; return done_(fail(c, VP_EHIP, std::string("vp_slice_run: ") + hipGetErrorString(q))); }
                            if (!__atomic_load_n(const_cast<int*>(h_words + 1), __ATOMIC_ACQUIRE) &&
                                __atomic_load_n(const_cast<int*>(h_words), __ATOMIC_RELAXED) == seen) h_words = nullptr;   // word not delivered: poll by copy
                        }
                        t_seen = std::chrono::steady_clock::now();
                    }
                    continue;
                }
                if ((rc = round())) return done_(rc);
                ++enq;
            } else {                                 // no mapped word: look at the device's state every few rounds
                for (int r = 0; r < ahead; ++r)
                    if ((rc = round())) return done_(rc);
                STRY(0, hipSetDevice(c->device));
                STRY(0, hipMemcpyAsync(h_prog, dv[0].prog, sizeof(h_prog), hipMemcpyDeviceToHost, dv[0].s));
                STRY(0, hipStreamSynchronize(dv[0].s));
                if (h_prog[2]) break;
            }
        }
        // the segment is over on the device: drain what is still enqueued (launches that return at once), look at the outcome
        for (int i = G - 1; i >= 0; --i) {
            STRY(i, hipSetDevice(cx[i]->device));
            if (i == 0) STRY(0, hipMemcpyAsync(h_prog, dv[0].prog, sizeof(h_prog), hipMemcpyDeviceToHost, dv[0].s));
            STRY(i, hipStreamSynchronize(dv[i].s));
        }
        STRY(0, hipGetLastError());
        if (h_prog[3] == 1) { *bad = 0; return done_(fail(c, VP_ENAN, "vp_slice_run: Log Probability returned NaN")); }
        if (h_prog[3]) { *bad = 0; return done_(fail(c, VP_ESTATE, "vp_slice_run: a slice did not terminate")); }
        if (chain) {
            STRY(0, hipMemcpyAsync(chain + (size_t)done * W * D, dv[0].chain, (size_t)n * W * D * sizeof(double), hipMemcpyDeviceToHost, dv[0].s));
            STRY(0, hipMemcpyAsync(chain_lnprob + (size_t)done * W, dv[0].chain + seg * (size_t)W * D, (size_t)n * W * sizeof(double),
                                   hipMemcpyDeviceToHost, dv[0].s));
            STRY(0, hipStreamSynchronize(dv[0].s));
        }
        if (G > 1 && (rc = multi_barrier(m))) return done_(rc);
        done += n;
    }
    STRY(0, hipSetDevice(c->device));
    double h_mu_out[4];
    long long h_ll[4];
    STRY(0, hipMemcpyAsync(pos, dv[0].pos, (size_t)W * D * sizeof(double), hipMemcpyDeviceToHost, dv[0].s));
    STRY(0, hipMemcpyAsync(lnprob, dv[0].lp, (size_t)W * sizeof(double), hipMemcpyDeviceToHost, dv[0].s));
    STRY(0, hipMemcpyAsync(h_mu_out, dv[0].mu, sizeof(h_mu_out), hipMemcpyDeviceToHost, dv[0].s));
    STRY(0, hipMemcpyAsync(h_ll, dv[0].ll, sizeof(h_ll), hipMemcpyDeviceToHost, dv[0].s));
    if (mu_history && nsteps > 0) STRY(0, hipMemcpyAsync(mu_history, dv[0].muhist, (size_t)nsteps * sizeof(double), hipMemcpyDeviceToHost, dv[0].s));
    STRY(0, hipStreamSynchronize(dv[0].s));
    for (int i = 1; i < G; ++i) {                                    // the other replicas have nothing left in flight either
        STRY(i, hipSetDevice(cx[i]->device));
        STRY(i, hipStreamSynchronize(dv[i].s));
    }
    *mu = h_mu_out[0];
    *tune = h_mu_out[2] != 0.0 ? 1 + (int)h_mu_out[1] : 0;
    if (n_evals) *n_evals += (int64_t)h_ll[0];
#undef STRY
#undef SFAIL
    return done_(VP_OK);
}

int vp_slice_run(vp_ctx* c, int W, int D, double* pos, double* lnprob, int have_lnprob, int nsteps, double* mu,
                 int* tune, double tolerance, int patience, int maxsteps, uint64_t seed, uint64_t step0,
                 double* chain, double* chain_lnprob, double* mu_history, int64_t* n_evals) {
    if (!c) return VP_EINVAL;
    CtxGuard g(c);
    int rc = check_batch_args(c, W, D, pos, lnprob);
    if (rc) return rc;
    int bad = 0;
    vp_ctx* cx[1] = {c};
    return slice_run_impl(nullptr, cx, 1, W, D, pos, lnprob, have_lnprob, nsteps, mu, tune, tolerance, patience, maxsteps, seed, step0,
                          chain, chain_lnprob, mu_history, n_evals, &bad);
}

#ifdef VP_STAMPS
// diagnostic build only: the far-field workspace of the last lnprob launch (doubles)
extern "C" int vp_debug_read_ff(vp_ctx* c, double* out, long n) {
    if (!c || !c->d_ff) return -1;
    hipDeviceSynchronize();
    return hipMemcpy(out, c->d_ff, (size_t)n * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;
}
// diagnostic build only: thread 0's clock at the phases of the slice sampler's round kernels (round, stage)
extern "C" int vp_debug_read_slice_stamps(long long* out, int n) {
    const size_t bytes = sizeof(long long) * (size_t)std::min(n, vp::SLICE_STAMP_ROUNDS * vp::SLICE_STAMP_STAGES);
    hipDeviceSynchronize();
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(vp::g_slice_stamps), bytes, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
extern "C" int vp_debug_host_stamps(double* out8) { std::memcpy(out8, g_host_stamps, sizeof g_host_stamps); return 0; }
// diagnostic build only: the walker kernel's phase stamps of the last launch (shader clock), (walker, wave, stage)
extern "C" int vp_debug_read_stamps(long long* out, int n) {
    const size_t bytes = sizeof(long long) * (size_t)std::min(n, vp::STAMP_W * vp::STAMP_WAVES * vp::STAMP_STAGES);
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(vp::g_stamps), bytes, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif

void vp_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    const vp::Philox4 r = vp::philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1]);
    for (int i = 0; i < 4; ++i) out[i] = r.v[i];
}

int vp_voigt_h(vp_ctx* c, int na, const double* a, int nx, const double* x, double* out) {
    if (!c) return VP_EINVAL;
    CtxGuard g(c);
    if (na <= 0 || nx <= 0 || !a || !x || !out) return fail(c, VP_EINVAL, "vp_voigt_h: empty or NULL input");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t rec = (size_t)na * vp::LC_STRIDE, need = (rec + na + nx + (size_t)na * nx) * sizeof(double);
    int rc;
    if ((rc = ensure_scratch(c, need))) return rc;
    double* d_rec = c->d_scratch;
    double* d_a = d_rec + rec;
    double* d_x = d_a + na;
    double* d_o = d_x + nx;
    HIP_TRY(c, hipMemcpyAsync(d_a, a, na * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(d_x, x, nx * sizeof(double), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(vp::prep_h_kernel, dim3((na + 63) / 64), dim3(64), 0, c->stream, d_a, (int)na, d_rec);
    hipLaunchKernelGGL(vp::voigt_h_kernel, dim3((nx + 255) / 256, na), dim3(256), 0, c->stream, d_rec, d_x, nx, d_o);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(out, d_o, (size_t)na * nx * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return VP_OK;
}

int vp_profile_enable(vp_ctx* c, int enable) {
    if (!c) return VP_EINVAL;
    CtxGuard g(c);
    c->profiling = enable != 0;
    c->ev_used = 0;
    c->spans.clear();
    return VP_OK;
}

int vp_profile_read(vp_ctx* c, double* prep_ms, double* tile_ms, double* finalize_ms, int* n_tile_launches) {
    if (!c) return VP_EINVAL;
    CtxGuard g(c);
    HIP_TRY(c, hipSetDevice(c->device));
    double acc[3] = {0, 0, 0};
    int ntile = 0;
    for (const auto& sp : c->spans) {
        if (sp.a == (size_t)-1 || sp.b == (size_t)-1) continue;
        HIP_TRY(c, hipEventSynchronize(c->ev_pool[sp.b]));
        float ms = 0.f;
        HIP_TRY(c, hipEventElapsedTime(&ms, c->ev_pool[sp.a], c->ev_pool[sp.b]));
        acc[sp.kind] += ms;
        if (sp.kind == 1) ++ntile;
    }
    if (prep_ms) *prep_ms = acc[0];
    if (tile_ms) *tile_ms = acc[1];
    if (finalize_ms) *finalize_ms = acc[2];
    if (n_tile_launches) *n_tile_launches = ntile;
    c->ev_used = 0;
    c->spans.clear();
    return VP_OK;
}


// ---- several GPUs from ONE process, no torch / RCCL needed (vp_ctx_create(n_devices, device_ids)) ------
// Walkers are independent, so a batch is cut into contiguous blocks of ceil(W / G) theta rows, one per context;
// every context holds the same static data; all blocks are enqueued before any is waited for, and each block's
// lnprob lands directly in its slice of the caller's output -- the "gather" of this path is G device-to-host
// copies into one host vector (the RCCL all-gather is for device-resident ensembles, rbvfit_amd/dist.py).
int vp_multi_create(vp_multi** out, int n_devices, const int* device_ids) {
    if (!out) return fail(nullptr, VP_EINVAL, "out is NULL");
    *out = nullptr;
    if (n_devices <= 0 || !device_ids) return fail(nullptr, VP_EINVAL, "vp_multi_create: n_devices must be positive and device_ids non-NULL");
    vp_multi* m = new (std::nothrow) vp_multi();
    if (!m) return fail(nullptr, VP_ENOMEM, "out of host memory");
    for (int i = 0; i < n_devices; ++i) {
        vp_ctx* c = nullptr;
        const int rc = vp_ctx_create(&c, device_ids[i]);
        if (rc) {                                   // (g_create_error holds the reason)
            for (vp_ctx* p : m->ctx) vp_ctx_destroy(p);
            delete m;
            return rc;
        }
        m->ctx.push_back(c);
    }
    // the contexts write into each other's ensembles (vp_multi_stretch_run): peer access between distinct devices
    for (vp_ctx* a : m->ctx)
        for (vp_ctx* b : m->ctx)
            if (a->device != b->device && hipSetDevice(a->device) == hipSuccess) {
                const hipError_t pe = hipDeviceEnablePeerAccess(b->device, 0);
                if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) m->no_peer = true;
                (void)hipGetLastError();
            }
    *out = m;
    return VP_OK;
}

int vp_multi_destroy(vp_multi* m) {
    if (!m) return VP_OK;
    for (size_t i = 0; i < m->ev.size(); ++i) {
        hipSetDevice(m->ctx[i]->device);
        hipEventDestroy(m->ev[i]);
    }
    for (vp_ctx* c : m->ctx) vp_ctx_destroy(c);
    delete m;
    return VP_OK;
}

int vp_multi_n_devices(const vp_multi* m) { return m ? (int)m->ctx.size() : 0; }
vp_ctx* vp_multi_ctx(vp_multi* m, int i) { return (m && i >= 0 && i < (int)m->ctx.size()) ? m->ctx[i] : nullptr; }
const char* vp_multi_last_error(const vp_multi* m) { return m ? m->err.c_str() : g_create_error.c_str(); }

static int multi_fail(vp_multi* m, int i, int rc) {
    m->err = "device slot " + std::to_string(i) + ": " + vp_last_error(m->ctx[i]);
    return rc;
}

static int multi_broken(vp_multi* m) {
    m->err = "this vp_multi is unusable: an earlier vp_multi_set_bounds / vp_multi_add_instrument failed on one device and could not be undone on the others";
    return VP_ESTATE;
}

// (internal) drop the instrument added last to one context: the undo step of vp_multi_add_instrument
static int ctx_pop_instrument(vp_ctx* c) {
    CtxGuard g(c);
    if (c->inst.empty()) return VP_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipDeviceSynchronize());
    for (void* p : c->inst.back().allocs) hipFree(p);
    c->inst.pop_back();
    c->meta_dirty = true;
    return VP_OK;
}

int vp_multi_set_bounds(vp_multi* m, int D, const double* lb, const double* ub) {
    if (!m) return VP_EINVAL;
    std::lock_guard<std::mutex> g(m->mu);
    if (m->broken) return multi_broken(m);
    // what the contexts hold now (they are kept identical), to put back should a later context fail
    std::vector<double> old_lb, old_ub;
    int oldD = 0;
    {
        vp_ctx* c0 = m->ctx[0];
        std::lock_guard<std::mutex> g0(c0->mu);
        oldD = c0->D; old_lb = c0->h_lb; old_ub = c0->h_ub;
    }
    for (size_t i = 0; i < m->ctx.size(); ++i) {
        const int rc = vp_set_bounds(m->ctx[i], D, lb, ub);
        if (!rc) continue;
        multi_fail(m, (int)i, rc);
        for (size_t k = 0; k < i; ++k) {                 // undo on the contexts already changed
            if (oldD <= 0 || vp_set_bounds(m->ctx[k], oldD, old_lb.data(), old_ub.data()) != VP_OK) m->broken = true;
        }
        return rc;
    }
    return VP_OK;
}

int vp_multi_add_instrument(vp_multi* m, int P, const double* wave, const double* flux, const double* inv_sigma2,
                            const double* log_inv_sigma2, int L, const double* lambda0, const double* gamma,
                            const double* f, const double* zfac, const int32_t* N_idx, const int32_t* b_idx,
                            const int32_t* v_idx, int K, const double* taps, int lsf_mode, int voigt_method,
                            int* inst_index) {
    if (!m) return VP_EINVAL;
    std::lock_guard<std::mutex> g(m->mu);
    if (m->broken) return multi_broken(m);
    for (size_t i = 0; i < m->ctx.size(); ++i) {
        const int rc = vp_add_instrument(m->ctx[i], P, wave, flux, inv_sigma2, log_inv_sigma2, L, lambda0, gamma, f, zfac, N_idx, b_idx,
                                         v_idx, K, taps, lsf_mode, voigt_method, inst_index);
        if (!rc) continue;
        multi_fail(m, (int)i, rc);                       // (argument errors fail on the first context: nothing to undo)
        for (size_t k = 0; k < i; ++k)
            if (ctx_pop_instrument(m->ctx[k]) != VP_OK) m->broken = true;
        return rc;
    }
    return VP_OK;
}

int vp_multi_lnprob_batch(vp_multi* m, int W, int D, const double* theta, double* out) {
    if (!m) return VP_EINVAL;
    std::lock_guard<std::mutex> g(m->mu);
    if (m->broken) return multi_broken(m);
    if (W < 0 || (W > 0 && (!theta || !out))) { m->err = "vp_multi_lnprob_batch: bad batch"; return VP_EINVAL; }
    const int G = (int)m->ctx.size();
    const int per = W > 0 ? (W + G - 1) / G : 0;
    std::vector<std::unique_lock<std::mutex>> locks;
    std::vector<char> begun(G, 0);                     // whose batch is in flight (its pinned buffer exists and will be written)
    int rc = VP_OK;
    for (int i = 0; i < G && !rc; ++i) {               // every block in flight before any wait
        const int lo = std::min(i * per, W), n = std::min(lo + per, W) - lo;
        vp_ctx* c = m->ctx[i];
        locks.emplace_back(c->mu);
        prearm_cancel(c);
        if ((rc = check_batch_args(c, n, D, theta, out))) { multi_fail(m, i, rc); break; }
        if (n <= 0) continue;
        if ((rc = lnprob_host_begin(c, n, D, theta + (size_t)lo * D))) multi_fail(m, i, rc);
        else begun[i] = 1;
    }
    // drain what was started (also when a later block failed); `out` is written only when every block succeeded
    for (int i = 0; i < G; ++i) {
        if (!begun[i]) continue;
        vp_ctx* c = m->ctx[i];
        if (host_wait(c) != VP_OK && !rc) rc = multi_fail(m, i, VP_EHIP);
    }
    if (rc) return rc;
    for (int i = 0; i < G; ++i) {
        if (!begun[i]) continue;
        const int lo = std::min(i * per, W), n = std::min(lo + per, W) - lo;
        std::memcpy(out + lo, m->ctx[i]->h_pinned + (size_t)n * D, (size_t)n * sizeof(double));
    }
    return VP_OK;
}

// All G streams wait for all G streams: G event records + G (G - 1) stream waits, no host synchronisation.
static int multi_barrier(vp_multi* m) {
    const int G = (int)m->ctx.size();
    for (int g = 0; g < G; ++g) {
        vp_ctx* c = m->ctx[g];
        HIP_TRY(c, hipSetDevice(c->device));
        HIP_TRY(c, hipEventRecord(m->ev[g], c->stream));
    }
    for (int g = 0; g < G; ++g) {
        vp_ctx* c = m->ctx[g];
        HIP_TRY(c, hipSetDevice(c->device));
        for (int h = 0; h < G; ++h)
            if (h != g) HIP_TRY(c, hipStreamWaitEvent(c->stream, m->ev[h], 0));
    }
    return VP_OK;
}

// One ensemble, G device contexts (BASELINE config 4: "2048 zeus walkers sharded over 8 GPUs"; vfit_mcmc.py:425-440,
// 536-540 fans ONE ensemble over its workers): see csrc/sampler_kernels.h.  Per half-step every context runs its block of
// the active half -- one walker_kernel launch where vp_stretch_run would use one for the whole half, else propose ->
// lnprob launches -> accept -- with the launch structure chosen as for the whole half (policy_W), so every row gets
// the bits a single context gives it; then the event barrier.  Chain: replica 0 keeps it.
int vp_multi_stretch_run(vp_multi* m, int W, int D, double* pos, double* lnprob, int have_lnprob, int nsteps, double a,
                         uint64_t seed, uint64_t step0, double* chain, double* chain_lnprob, int64_t* naccepted) {
    if (!m) return VP_EINVAL;
    std::lock_guard<std::mutex> g(m->mu);
    if (m->broken) return multi_broken(m);
    const int G = (int)m->ctx.size();
    if (G > vp::MAX_REPLICAS) { m->err = "vp_multi_stretch_run: at most " + std::to_string(vp::MAX_REPLICAS) + " device contexts"; return VP_EINVAL; }
    if (m->no_peer) { m->err = "vp_multi_stretch_run: the devices cannot map each other's memory (no peer access)"; return VP_ESTATE; }
    std::vector<std::unique_lock<std::mutex>> locks;
    for (int i = 0; i < G; ++i) { locks.emplace_back(m->ctx[i]->mu); prearm_cancel(m->ctx[i]); }
    int rc;
    for (int i = 0; i < G; ++i)
        if ((rc = check_batch_args(m->ctx[i], W, D, pos, lnprob))) return multi_fail(m, i, rc);
    vp_ctx* c0 = m->ctx[0];
    if (W < 2 || (W & 1)) { c0->err = "vp_multi_stretch_run: the number of walkers must be even and >= 2"; return multi_fail(m, 0, VP_EINVAL); }
    if (nsteps < 0 || !(a > 1.0)) { c0->err = "vp_multi_stretch_run: nsteps must be >= 0 and a > 1"; return multi_fail(m, 0, VP_EINVAL); }
    if ((chain == nullptr) != (chain_lnprob == nullptr)) { c0->err = "vp_multi_stretch_run: chain and chain_lnprob go together"; return multi_fail(m, 0, VP_EINVAL); }
    if (have_lnprob)
        for (int w = 0; w < W; ++w)
            if (lnprob[w] != lnprob[w]) { c0->err = "vp_multi_stretch_run: the initial lnprob holds NaN (Probability function returned NaN)"; return multi_fail(m, 0, VP_ENAN); }
    if ((int)m->ev.size() != G) {
        for (int i = (int)m->ev.size(); i < G; ++i) {
            hipEvent_t e;
            if (hipSetDevice(m->ctx[i]->device) != hipSuccess || hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
                m->ctx[i]->err = "hipEventCreate failed"; return multi_fail(m, i, VP_EHIP);
            }
            m->ev.push_back(e);
        }
    }
    const int half = W / 2, per = (half + G - 1) / G;
    bool one_device = true;
    for (int i = 1; i < G; ++i) one_device = one_device && m->ctx[i]->device == c0->device;
    const bool flags_mode = c0->tune.multi_sync > 0;
    // device state per context: pos (W,D) | lp (W) | prop (per,D) | lp_new (per) | zz (per) | chain chunk (flags mode: every
    // context keeps the rows of ITS walkers; events mode: replica 0 snapshots the ensemble) ; nacc (W) | nanflag | flags (G) | done | timeout
    const size_t row = (size_t)W * (D + 1);
    const size_t chunk = chain ? std::max<size_t>(1, std::min<size_t>((size_t)std::max(nsteps, 1), ((size_t)256 << 20) / (row * sizeof(double)))) : 0;
    struct Dev { double *pos, *lp, *prop, *lpnew, *zz, *chain; long long* nacc; int *nan, *flags, *timeout; unsigned int* done; };
    std::vector<Dev> dv(G);
    for (int i = 0; i < G; ++i) {
        vp_ctx* c = m->ctx[i];
#define MTRY(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { c->err = std::string(#expr) + ": " + hipGetErrorString(e__); return multi_fail(m, i, VP_EHIP); } } while (0)
        MTRY(hipSetDevice(c->device));
        if ((rc = ensure_workspace(c, std::max(W, per)))) return multi_fail(m, i, rc);
        const size_t nchain = (flags_mode || i == 0) ? chunk * row : 0;
        const size_t nd = (size_t)W * D + W + (size_t)per * D + 2 * (size_t)per + nchain;
        if ((rc = ensure_scratch(c, nd * sizeof(double) + (size_t)W * sizeof(long long) + (vp::MAX_REPLICAS + 4 + vp::PUB_GROUPS + 1) * sizeof(int) + 64))) return multi_fail(m, i, rc);
        Dev& d = dv[i];
        d.pos = c->d_scratch; d.lp = d.pos + (size_t)W * D; d.prop = d.lp + W; d.lpnew = d.prop + (size_t)per * D;
        d.zz = d.lpnew + per; d.chain = d.zz + per;
        d.nacc = reinterpret_cast<long long*>(d.chain + nchain);
        d.nan = reinterpret_cast<int*>(d.nacc + W);
        d.flags = d.nan + 1; d.done = reinterpret_cast<unsigned int*>(d.flags + vp::MAX_REPLICAS);      // (PUB_GROUPS + 1 counters)
        d.timeout = reinterpret_cast<int*>(d.done + vp::PUB_GROUPS + 1);
        MTRY(hipMemcpyAsync(d.pos, pos, (size_t)W * D * sizeof(double), hipMemcpyHostToDevice, c->stream));
        MTRY(hipMemsetAsync(d.nacc, 0, (size_t)W * sizeof(long long) + (vp::MAX_REPLICAS + 4 + vp::PUB_GROUPS + 1) * sizeof(int), c->stream));
        if (have_lnprob) MTRY(hipMemcpyAsync(d.lp, lnprob, (size_t)W * sizeof(double), hipMemcpyHostToDevice, c->stream));
        else {
            // every replica evaluates the whole start state itself (once per run; the same launches as vp_stretch_run's)
            c->policy_W = 0;
            if ((rc = enqueue_lnprob(c, W, d.pos, d.lp, c->stream))) return multi_fail(m, i, rc);
            hipLaunchKernelGGL(vp::nan_flag_kernel, dim3((W + 255) / 256), dim3(256), 0, c->stream, d.lp, W, d.nan);
        }
    }
    if (!have_lnprob) {
        int h_nan0 = 0;
        vp_ctx* c = c0; const int i = 0;
        MTRY(hipSetDevice(c->device));
        MTRY(hipMemcpyAsync(&h_nan0, dv[0].nan, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        MTRY(hipStreamSynchronize(c->stream));
        if (h_nan0) { c0->err = "vp_multi_stretch_run: the initial lnprob holds NaN (Probability function returned NaN)"; return multi_fail(m, 0, VP_ENAN); }
    }
    // every replica's start state (and its zeroed flags) is in place before any kernel of the run starts
    for (int i = 0; i < G; ++i) {
        vp_ctx* c = m->ctx[i];
        MTRY(hipSetDevice(c->device));
        MTRY(hipStreamSynchronize(c->stream));
    }
    const int thr = 64;
    auto finish = [&](int code) { for (int i = 0; i < G; ++i) m->ctx[i]->policy_W = 0; return code; };
    std::vector<double> h_chunk;                       // flags mode: one context's chain chunk on its way to the caller's arrays
    int seq = 0;                                       // half-steps of this call so far
    for (int done = 0; done < nsteps;) {
        const int n = chain ? (int)std::min<size_t>(chunk, (size_t)(nsteps - done)) : nsteps - done;
        for (int it = 0; it < n; ++it) {
            const uint64_t step = step0 + (uint64_t)(done + it);
            for (int h = 0; h < 2; ++h) {
                const int s0 = h ? half : 0, cc0 = h ? 0 : half;
                ++seq;
                for (int i = 0; i < G; ++i) {
                    vp_ctx* c = m->ctx[i];
                    const int k0 = std::min(i * per, half), nk = std::min(k0 + per, half) - k0;
                    MTRY(hipSetDevice(c->device));
                    c->policy_W = half;
                    const Dev& d = dv[i];
                    vp::Replicas R{};
                    R.n = G;
                    for (int j = 0; j < G; ++j) { R.pos[j] = dv[j].pos; R.lp[j] = dv[j].lp; R.flags[j] = dv[j].flags; }
                    R.done = d.done; R.timeout = d.timeout; R.me = i; R.seq = seq; R.sync = flags_mode ? (one_device ? 1 : 2) : 0;
                    double* cp = (chain && flags_mode) ? d.chain + (size_t)it * W * D : (double*)nullptr;
                    double* cl = (chain && flags_mode) ? d.chain + chunk * (size_t)W * D + (size_t)it * W : (double*)nullptr;
                    if (nk <= 0) {
                        // (a context without rows in this half still has to tell the others that it is through)
                        if (flags_mode)
                            hipLaunchKernelGGL(vp::stretch_accept_block_kernel, dim3(1), dim3(thr), 0, c->stream, d.pos, d.lp, R, d.prop, d.lpnew,
                                               d.zz, D, s0, k0, 0, seed, step, h, d.nacc, d.nan, cp, cl);
                        continue;
                    }
                    // (the choice vp_stretch_run makes for the whole half)
                    const bool one_launch = !c->tune.no_fused_accept && c->tune.walker != 0 && walker_applies(c, half) &&
                                            (c->inst[0].dev.NCm == 0 || !c->tune.walker_clusters);
                    if (one_launch) {
                        vp::StretchArgs sa{};
                        sa.pos = d.pos; sa.lp = d.lp; sa.nacc = d.nacc; sa.nanflag = d.nan;
                        sa.chain_pos = cp; sa.chain_lp = cl;
                        sa.a = a; sa.seed = seed; sa.step = step; sa.s0 = s0 + k0; sa.c0 = cc0; sa.nC = half; sa.half = h;
                        sa.rep = R;
                        launch_walker_stretch(c, nk, sa, c->stream, 0, walker_split_for(c, half));    // (the form the WHOLE half would get)
                    } else {
                        hipLaunchKernelGGL(vp::stretch_propose_block_kernel, dim3((nk + thr - 1) / thr), dim3(thr), 0, c->stream, d.pos, D, s0,
                                           half, cc0, half, a, seed, step, h, k0, nk, d.prop, d.zz, R);
                        if ((rc = enqueue_lnprob(c, nk, d.prop, d.lpnew, c->stream))) return finish(multi_fail(m, i, rc));
                        hipLaunchKernelGGL(vp::stretch_accept_block_kernel, dim3((nk + thr - 1) / thr), dim3(thr), 0, c->stream, d.pos, d.lp, R,
                                           d.prop, d.lpnew, d.zz, D, s0, k0, nk, seed, step, h, d.nacc, d.nan, cp, cl);
                    }
                    MTRY(hipGetLastError());
                }
                if (!flags_mode && (rc = multi_barrier(m))) return finish(multi_fail(m, 0, rc));
            }
            if (chain && !flags_mode) {
                vp_ctx* c = c0; const int i = 0;
                MTRY(hipSetDevice(c->device));
                MTRY(hipMemcpyAsync(dv[0].chain + (size_t)it * W * D, dv[0].pos, (size_t)W * D * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
                MTRY(hipMemcpyAsync(dv[0].chain + chunk * (size_t)W * D + (size_t)it * W, dv[0].lp, (size_t)W * sizeof(double),
                                    hipMemcpyDeviceToDevice, c->stream));
                // nobody writes into replica 0 (the next half-step's moved rows) before this snapshot is taken
                MTRY(hipEventRecord(m->ev[0], c->stream));
                for (int j = 1; j < G; ++j) {
                    vp_ctx* cj = m->ctx[j];
                    if (hipSetDevice(cj->device) != hipSuccess || hipStreamWaitEvent(cj->stream, m->ev[0], 0) != hipSuccess) {
                        cj->err = "hipStreamWaitEvent failed"; return finish(multi_fail(m, j, VP_EHIP));
                    }
                }
            }
        }
        if (chain && !flags_mode) {
            vp_ctx* c = c0; const int i = 0;
            MTRY(hipSetDevice(c->device));
            MTRY(hipMemcpyAsync(chain + (size_t)done * W * D, dv[0].chain, (size_t)n * W * D * sizeof(double), hipMemcpyDeviceToHost, c->stream));
            MTRY(hipMemcpyAsync(chain_lnprob + (size_t)done * W, dv[0].chain + chunk * (size_t)W * D, (size_t)n * W * sizeof(double),
                                hipMemcpyDeviceToHost, c->stream));
            MTRY(hipStreamSynchronize(c->stream));
        }
        if (chain && flags_mode) {
            // every context holds the chain rows of the walkers it moves (blocks k0 ... of both halves): merged here
            h_chunk.resize((size_t)n * row);
            for (int i = 0; i < G; ++i) {
                vp_ctx* c = m->ctx[i];
                const int k0 = std::min(i * per, half), nk = std::min(k0 + per, half) - k0;
                if (nk <= 0) continue;
                MTRY(hipSetDevice(c->device));
                MTRY(hipMemcpyAsync(h_chunk.data(), dv[i].chain, (size_t)n * W * D * sizeof(double), hipMemcpyDeviceToHost, c->stream));
                MTRY(hipMemcpyAsync(h_chunk.data() + (size_t)n * W * D, dv[i].chain + chunk * (size_t)W * D, (size_t)n * W * sizeof(double),
                                    hipMemcpyDeviceToHost, c->stream));
                MTRY(hipStreamSynchronize(c->stream));
                for (int t = 0; t < n; ++t)
                    for (int hh = 0; hh < 2; ++hh) {
                        const size_t w0 = (size_t)hh * half + k0;
                        std::memcpy(chain + ((size_t)(done + t) * W + w0) * D, h_chunk.data() + ((size_t)t * W + w0) * D, (size_t)nk * D * sizeof(double));
                        std::memcpy(chain_lnprob + (size_t)(done + t) * W + w0, h_chunk.data() + (size_t)n * W * D + (size_t)t * W + w0,
                                    (size_t)nk * sizeof(double));
                    }
            }
        }
        done += n;
    }
    // results: the ensemble from replica 0, every walker's acceptance count from the context that moved it, NaN flags from all
    int any_nan = 0, any_timeout = 0;
    std::vector<long long> h_nacc(W);
    for (int i = 0; i < G; ++i) {                      // (flags mode: replica 0 is complete only when every context is through)
        vp_ctx* c = m->ctx[i];
        MTRY(hipSetDevice(c->device));
        MTRY(hipStreamSynchronize(c->stream));
    }
    for (int i = 0; i < G; ++i) {
        vp_ctx* c = m->ctx[i];
        MTRY(hipSetDevice(c->device));
        int h_nan = 0, h_to = 0;
        MTRY(hipMemcpyAsync(&h_nan, dv[i].nan, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        MTRY(hipMemcpyAsync(&h_to, dv[i].timeout, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        MTRY(hipMemcpyAsync(h_nacc.data(), dv[i].nacc, (size_t)W * sizeof(long long), hipMemcpyDeviceToHost, c->stream));
        if (i == 0) {
            MTRY(hipMemcpyAsync(pos, dv[0].pos, (size_t)W * D * sizeof(double), hipMemcpyDeviceToHost, c->stream));
            MTRY(hipMemcpyAsync(lnprob, dv[0].lp, (size_t)W * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        }
        MTRY(hipStreamSynchronize(c->stream));
        any_nan |= h_nan; any_timeout |= h_to;
        if (naccepted) {
            const int k0 = std::min(i * per, half), nk = std::min(k0 + per, half) - k0;
            for (int hh = 0; hh < 2; ++hh)
                for (int k = k0; k < k0 + nk; ++k) naccepted[hh * half + k] += (int64_t)h_nacc[hh * half + k];
        }
    }
#undef MTRY
    if (any_timeout) { c0->err = "vp_multi_stretch_run: a context gave up waiting for its peers' half-step (in-kernel flags; try the multi_sync = 0 option)"; return finish(multi_fail(m, 0, VP_ESTATE)); }
    if (any_nan) { c0->err = "vp_multi_stretch_run: Probability function returned NaN"; return finish(multi_fail(m, 0, VP_ENAN)); }
    return finish(VP_OK);
}

int vp_multi_slice_run(vp_multi* m, int W, int D, double* pos, double* lnprob, int have_lnprob, int nsteps, double* mu,
                       int* tune, double tolerance, int patience, int maxsteps, uint64_t seed, uint64_t step0,
                       double* chain, double* chain_lnprob, double* mu_history, int64_t* n_evals) {
    if (!m) return VP_EINVAL;
    std::lock_guard<std::mutex> g(m->mu);
    if (m->broken) return multi_broken(m);
    const int G = (int)m->ctx.size();
    if (G > vp::MAX_REPLICAS) { m->err = "vp_multi_slice_run: at most " + std::to_string(vp::MAX_REPLICAS) + " device contexts"; return VP_EINVAL; }
    if (m->no_peer) { m->err = "vp_multi_slice_run: the devices cannot map each other's memory (no peer access)"; return VP_ESTATE; }
    std::vector<std::unique_lock<std::mutex>> locks;
    for (int i = 0; i < G; ++i) { locks.emplace_back(m->ctx[i]->mu); prearm_cancel(m->ctx[i]); }
    for (int i = 0; i < G; ++i)
        if (int rc = check_batch_args(m->ctx[i], W, D, pos, lnprob)) return multi_fail(m, i, rc);
    for (int i = (int)m->ev.size(); i < G; ++i) {
        hipEvent_t e;
        if (hipSetDevice(m->ctx[i]->device) != hipSuccess || hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
            m->ctx[i]->err = "hipEventCreate failed"; return multi_fail(m, i, VP_EHIP);
        }
        m->ev.push_back(e);
    }
    int bad = 0;
    const int rc = slice_run_impl(m, m->ctx.data(), G, W, D, pos, lnprob, have_lnprob, nsteps, mu, tune, tolerance, patience, maxsteps,
                                  seed, step0, chain, chain_lnprob, mu_history, n_evals, &bad);
    return rc ? multi_fail(m, bad, rc) : VP_OK;
}

void* vp_ctx_stream(const vp_ctx* c) { return c ? (void*)c->stream : nullptr; }

int vp_num_instruments(const vp_ctx* c) {
    if (!c) return 0;
    std::lock_guard<std::mutex> g(c->mu);
    return (int)c->inst.size();
}
int vp_ndim(const vp_ctx* c) {
    if (!c) return 0;
    std::lock_guard<std::mutex> g(c->mu);
    return c->D;
}
int vp_instrument_pixels(const vp_ctx* c, int inst) {
    if (!c) return -1;
    std::lock_guard<std::mutex> g(c->mu);
    if (inst < 0 || inst >= (int)c->inst.size()) return -1;
    return c->inst[inst].dev.P;
}
int vp_device_id(const vp_ctx* c) { return c ? c->device : -1; }
int vp_last_farfield_info(vp_ctx* c, int* variant, int64_t* covered, int64_t* covered_members, int64_t* pairs) {
    if (!c) return VP_EINVAL;
    CtxGuard g(c);
    if (variant) *variant = 0;
    if (covered) *covered = 0;
    if (covered_members) *covered_members = 0;
    if (pairs) *pairs = 0;
    const vp_ctx::LastFF lf = c->last_ff;
    if (lf.inst < 0 || !lf.ff || lf.W <= 0 || lf.nbk <= 0) return VP_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipDeviceSynchronize());
    const size_t n = (size_t)lf.W * lf.nbk;
    std::vector<double> h(n * vp::FF_STRIDE);
    HIP_TRY(c, hipMemcpy(h.data(), lf.ff, h.size() * sizeof(double), hipMemcpyDeviceToHost));
    const Instrument& in = c->inst[lf.inst];
    int64_t cov = 0, mem = 0;
    for (size_t i = 0; i < n; ++i) {
        unsigned long long m[2];
        memcpy(m, h.data() + i * vp::FF_STRIDE + vp::FF_MASK0, sizeof m);
        cov += __builtin_popcountll(m[0]) + __builtin_popcountll(m[1]);
        mem += __builtin_popcountll(m[0] & in.member_mask[0]) + __builtin_popcountll(m[1] & in.member_mask[1]);
    }
    if (variant) *variant = lf.members ? 2 : 1;
    if (covered) *covered = cov;
    if (covered_members) *covered_members = mem;
    if (pairs) *pairs = (int64_t)n * in.lines.L;
    return VP_OK;
}

int vp_last_launch_kind(const vp_ctx* c) {
    if (!c) return -1;
    std::lock_guard<std::mutex> g(c->mu);
    return c->last_kind;
}
int vp_last_walker_split(const vp_ctx* c) {
    if (!c) return -1;
    std::lock_guard<std::mutex> g(c->mu);
    return c->last_kind == 1 ? c->last_split : 0;
}

}  // extern "C"
