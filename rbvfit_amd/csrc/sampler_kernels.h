// Device-resident stretch-move ensemble sampler: the walker loop the reference
// delegates to emcee (vfit_mcmc.py:408-423, 536-540), with positions, lnprob, proposals and
// accept/reject kept in HBM.  Goodman & Weare (2010) stretch move in emcee's red-blue form: the
// ensemble is split in two halves and each half is updated against the other with ONE lnprob batch.
//
// Randomness: Philox4x32-10 (Salmon et al. 2011), counter-based -- the draw for (step, half, walker,
// purpose) is a pure function of the seed, so a run is reproducible and can be split into calls.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vp {

struct Philox4 { uint32_t v[4]; };

__host__ __device__ inline Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                 uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    Philox4 o;
    o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
    return o;
}

// 53-bit uniform in [0, 1) from two 32-bit words
__host__ __device__ inline double u01(uint32_t hi, uint32_t lo) {
    const uint64_t x = (((uint64_t)hi << 32) | lo) >> 11;
    return (double)x * (1.0 / 9007199254740992.0);
}

// counter layout: (walker, step low, step high | half << 31, purpose)
__device__ inline Philox4 draw(uint64_t seed, uint64_t step, int half, int walker, uint32_t purpose) {
    return philox4x32_10((uint32_t)walker, (uint32_t)step, (uint32_t)(step >> 32) | ((uint32_t)half << 31), purpose,
                         (uint32_t)seed, (uint32_t)(seed >> 32));
}

// Proposal for the nS walkers of half S (rows [s0, s0+nS)) against the complementary rows [c0, c0+nC):
//   z ~ g(z) ∝ 1/sqrt(z) on [1/a, a]:  z = ((a-1) u + 1)^2 / a;   Y = X_j + z (X_k - X_j)
__device__ inline void stretch_propose(int k, const double* __restrict__ pos, int D, int s0, int nS, int c0, int nC,
                                       double a, uint64_t seed, uint64_t step, int half, double* __restrict__ prop,
                                       double* __restrict__ zz) {
    if (k >= nS) return;
    const Philox4 r = draw(seed, step, half, s0 + k, 0u);
    const double u1 = u01(r.v[0], r.v[1]), u2 = u01(r.v[2], r.v[3]);
    const double t = (a - 1.0) * u1 + 1.0;
    const double z = t * t / a;
    int j = (int)(u2 * (double)nC);
    j = j < nC - 1 ? j : nC - 1;
    const double* __restrict__ x = pos + (size_t)(s0 + k) * D;
    const double* __restrict__ c = pos + (size_t)(c0 + j) * D;
    double* __restrict__ y = prop + (size_t)k * D;
    for (int d = 0; d < D; ++d) y[d] = c[d] - (c[d] - x[d]) * z;
    zz[k] = z;
}
__global__ void stretch_propose_kernel(const double* __restrict__ pos, int D, int s0, int nS, int c0, int nC, double a,
                                       uint64_t seed, uint64_t step, int half, double* __restrict__ prop,
                                       double* __restrict__ zz) {
    stretch_propose(blockIdx.x * blockDim.x + threadIdx.x, pos, D, s0, nS, c0, nC, a, seed, step, half, prop, zz);
}

// Accept/reject for half S and, when chain_pos != nullptr, storage of the whole ensemble's state
// (row `w` is handled by thread w; rows outside S are only stored).
//   ln q = (D-1) ln z + lnprob(Y) - lnprob(X);  accept when ln u < ln q.   NaN lnprob(Y) -> flag (emcee raises).
__device__ inline void stretch_accept(int w, double* __restrict__ pos, double* __restrict__ lp,
                                      const double* __restrict__ prop, const double* __restrict__ lp_new,
                                      const double* __restrict__ zz, int W, int D, int s0, int nS, uint64_t seed,
                                      uint64_t step, int half, long long* __restrict__ nacc, int* __restrict__ nanflag,
                                      double* __restrict__ chain_pos, double* __restrict__ chain_lp) {
    if (w >= W) return;
    double* __restrict__ x = pos + (size_t)w * D;
    const int k = w - s0;
    if (k >= 0 && k < nS) {
        const double ln = lp_new[k];
        if (ln != ln) {
            atomicExch(nanflag, 1);
        } else {
            const Philox4 r = draw(seed, step, half, w, 1u);
            const double u = u01(r.v[0], r.v[1]);
            const double lnq = (double)(D - 1) * log(zz[k]) + ln - lp[w];
            if (log(u) < lnq) {
                const double* __restrict__ y = prop + (size_t)k * D;
                for (int d = 0; d < D; ++d) x[d] = y[d];
                lp[w] = ln;
                nacc[w] += 1;
            }
        }
    }
    if (chain_pos) {
        double* __restrict__ o = chain_pos + (size_t)w * D;
        for (int d = 0; d < D; ++d) o[d] = x[d];
        chain_lp[w] = lp[w];
    }
}
__global__ void stretch_accept_kernel(double* __restrict__ pos, double* __restrict__ lp, const double* __restrict__ prop,
                                      const double* __restrict__ lp_new, const double* __restrict__ zz, int W, int D,
                                      int s0, int nS, uint64_t seed, uint64_t step, int half,
                                      long long* __restrict__ nacc, int* __restrict__ nanflag,
                                      double* __restrict__ chain_pos, double* __restrict__ chain_lp) {
    stretch_accept(blockIdx.x * blockDim.x + threadIdx.x, pos, lp, prop, lp_new, zz, W, D, s0, nS, seed, step, half, nacc,
                   nanflag, chain_pos, chain_lp);
}

// Accept/reject of one half followed by the proposal for the other half, in ONE single-workgroup launch
// (W <= 1024): the barrier orders the accepted positions before the proposals that read them.  Saves
// a launch per half-ensemble pass (each of these tiny kernels is launch latency and nothing else).
struct NextProposal { int s0, nS, c0, nC, half; uint64_t step; };
__global__ __launch_bounds__(1024) void stretch_accept_propose_kernel(
    double* __restrict__ pos, double* __restrict__ lp, double* __restrict__ prop, double* __restrict__ lp_new,
    double* __restrict__ zz, int W, int D, int s0, int nS, uint64_t seed, uint64_t step, int half,
    long long* __restrict__ nacc, int* __restrict__ nanflag, double* __restrict__ chain_pos,
    double* __restrict__ chain_lp, double a, NextProposal nx) {
    stretch_accept(threadIdx.x, pos, lp, prop, lp_new, zz, W, D, s0, nS, seed, step, half, nacc, nanflag, chain_pos,
                   chain_lp);
    __syncthreads();             // all reads of prop/zz and all writes of pos by this launch's accept are done
    stretch_propose(threadIdx.x, pos, D, nx.s0, nx.nS, nx.c0, nx.nC, a, seed, nx.step, nx.half, prop, zz);
}

// ---- one ensemble sharded over G device contexts (vp_multi_stretch_run) -------------------------------------------
// Every context holds the WHOLE ensemble (positions, lnprob); of the active half it proposes, evaluates and accepts
// only its block of rows [k0, k0 + nk), and writes the rows it moved into EVERY replica (its own included) -- plain
// stores through peer-mapped pointers, (D + 1) doubles per moved walker; the host orders the half-steps with events.
// The draws are keyed by the walker's index in the whole ensemble, so the chain does not depend on G.
constexpr int MAX_REPLICAS = 8;
struct Replicas {
    double* pos[MAX_REPLICAS]; double* lp[MAX_REPLICAS]; int n;
    // In-kernel ordering of the half-steps (sync != 0) instead of events between the contexts' streams: every replica has
    // a flag per context, flags[p][g] = the last half-step (1, 2, ... within the call) context g has finished, written by g
    // into every replica p behind all the rows it moved; the kernels of half-step s start by waiting for every peer's
    // flag >= s - 1 in their OWN replica.  No kernel waits for a kernel that is queued behind it, so the scheme cannot lock
    // up as long as the contexts' queues all run.  `done` counts the finished workgroups of the running kernel (reset by the
    // last one), `timeout` is set when a wait gives up (the host then fails the call instead of hanging the GPU).
    int* flags[MAX_REPLICAS];
    unsigned int* done;
    int* timeout;
    int me, seq, sync;
};
constexpr int SYNC_SPIN_LIMIT = 1 << 22;       // polls (each behind an s_sleep) before a wait gives up: ~ a second

// Scopes: sync == 1 -- all contexts on ONE device -- orders through that device's L2 (agent scope); sync == 2 -- contexts on
// different devices -- uses system scope, and only the LAST workgroup of a kernel pays the system-scope release (an L2
// write-back), the others release to the agent ahead of their count (measured on one device with system scope in every
// workgroup: 48 us per half-step instead of 22).
// all lanes: wait until every peer has finished half-step R.seq - 1 (their rows in this replica have landed)
__device__ __forceinline__ void replicas_wait(const Replicas& R) {
    if (!R.sync || R.seq <= 1 || R.n <= 1) return;
    const int lane = threadIdx.x & 63;
    const int need = R.seq - 1;
    int spins = 0;
    for (;;) {
        int f = need;
        if (lane < R.n && lane != R.me)
            f = R.sync == 2 ? __hip_atomic_load(R.flags[R.me] + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                            : __hip_atomic_load(R.flags[R.me] + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__ballot(f < need) == 0ull) break;
        if (++spins > SYNC_SPIN_LIMIT) {
            if (lane == 0) __hip_atomic_store(R.timeout, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        __builtin_amdgcn_s_sleep(8);
    }
    // nothing below is read before the flags were seen
    if (R.sync == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// one thread per workgroup, behind the workgroup's last store into the replicas: the last workgroup of the grid
// publishes the half-step to every replica.  The workgroups are counted in PUB_GROUPS groups (by index) whose last members
// count once more on a top word: 512 returning atomics on ONE word took the tail of a launch ~10 us (a word serves ~90 of them
// per microsecond), 16 words of 32 do not queue.  R.done: PUB_GROUPS group counters, then the top one.
constexpr unsigned int PUB_GROUPS = 16;
// (index: this caller's number among the nworkgroups callers -- the workgroup's index, or the walker's where a walker is several
//  workgroups of which one, the last to finish, calls: walker_kernel's split form)
__device__ __forceinline__ void replicas_publish(const Replicas& R, unsigned int nworkgroups, unsigned int index) {
    if (!R.sync) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");         // this workgroup's rows before its count
    const unsigned int g = index % PUB_GROUPS, ngroups = nworkgroups < PUB_GROUPS ? nworkgroups : PUB_GROUPS;
    const unsigned int members = (nworkgroups - g + PUB_GROUPS - 1u) / PUB_GROUPS;          // workgroups w < nworkgroups with w % 16 == g
    if (__hip_atomic_fetch_add(R.done + g, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) != members - 1u) return;
    __hip_atomic_store(R.done + g, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__hip_atomic_fetch_add(R.done + PUB_GROUPS, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) != ngroups - 1u) return;
    __hip_atomic_store(R.done + PUB_GROUPS, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (R.sync == 2) {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "");           // every workgroup's rows, out of this device's L2, before the flags
        for (int p = 0; p < R.n; ++p)
            __hip_atomic_store(R.flags[p] + R.me, R.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    } else {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
        for (int p = 0; p < R.n; ++p)
            __hip_atomic_store(R.flags[p] + R.me, R.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Ordering for the direct-write gather of lnprob batches (vp_gather_*), whose passes are whole launches on one stream: a
// launch's stores into the peers' vectors (system-scope atomic stores: no cache holds them) are complete when the launch is, so
// the NEXT launch on the stream can vouch for them -- its first workgroup raises this rank's flag in every peer to the pass
// before, then all its workgroups wait for the peers' flags of that pass before they compute.  No count of finished
// workgroups, no fence, nothing at the end of a launch: a count (a returning device-scope atomic or two per workgroup behind its
// last store) cost the tail of a 512-workgroup launch 5.5 us, release fences in front of it -- L2 write-backs on a GPU of eight
// L2s -- 11 us; a blocking RCCL all_gather 9 us.
// A launch that waits inside its workgroups needs its peers' launches to START while it holds its wave slots: true for one
// rank per GPU; ranks that share a GPU (tests, small jobs) could fill it with waiting workgroups, so for them the handshake is
// a one-wave launch of its own in front of the pass (R.sync = 0 in the pass's launches; vp_gather_connect's `shared_device`).
// all lanes of every wave; R.seq = the pass this launch computes (the flags of pass R.seq - 1 are raised and awaited)
__device__ __forceinline__ void replicas_handshake(const Replicas& R) {
    if (R.seq <= 1 || R.n <= 1 || !R.sync) return;
    const int lane = threadIdx.x & 63;
    const int need = R.seq - 1;
    if (blockIdx.x == 0 && threadIdx.x < 64 && lane < R.n && lane != R.me)
        __hip_atomic_store(R.flags[lane] + R.me, need, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    int spins = 0;
    for (;;) {
        int f = need;
        if (lane < R.n && lane != R.me) f = __hip_atomic_load(R.flags[R.me] + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (__ballot(f < need) == 0ull) break;
        if (++spins > SYNC_SPIN_LIMIT) {
            if (lane == 0) __hip_atomic_store(R.timeout, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        __builtin_amdgcn_s_sleep(8);
    }
}

__global__ void stretch_propose_block_kernel(const double* __restrict__ pos, int D, int s0, int nS, int c0, int nC, double a,
                                             uint64_t seed, uint64_t step, int half, int k0, int nk,
                                             double* __restrict__ prop, double* __restrict__ zz, Replicas R) {
    replicas_wait(R);                                   // the complementary half as the peers left it
    const int kk = blockIdx.x * blockDim.x + threadIdx.x;
    if (kk >= nk) return;
    (void)nS;
    const int k = k0 + kk;
    const Philox4 r = draw(seed, step, half, s0 + k, 0u);
    const double u1 = u01(r.v[0], r.v[1]), u2 = u01(r.v[2], r.v[3]);
    const double t = (a - 1.0) * u1 + 1.0;
    const double z = t * t / a;
    int j = (int)(u2 * (double)nC);
    j = j < nC - 1 ? j : nC - 1;
    const double* __restrict__ x = pos + (size_t)(s0 + k) * D;
    const double* __restrict__ c = pos + (size_t)(c0 + j) * D;
    double* __restrict__ y = prop + (size_t)kk * D;
    for (int d = 0; d < D; ++d) y[d] = c[d] - (c[d] - x[d]) * z;       // (stretch_propose's arithmetic)
    zz[kk] = z;
}

// accept / reject of the block (stretch_accept's arithmetic); moved rows go to every replica, the walker's chain entry
// (state after this step: every walker is in the active half exactly once per step) to this context's chain buffer
__global__ void stretch_accept_block_kernel(const double* __restrict__ pos_own, const double* __restrict__ lp_own, Replicas R,
                                            const double* __restrict__ prop, const double* __restrict__ lp_new,
                                            const double* __restrict__ zz, int D, int s0, int k0, int nk, uint64_t seed,
                                            uint64_t step, int half, long long* __restrict__ nacc, int* __restrict__ nanflag,
                                            double* __restrict__ chain_pos, double* __restrict__ chain_lp) {
    const int kk = blockIdx.x * blockDim.x + threadIdx.x;
    if (kk < nk) {
        const int w = s0 + k0 + kk;
        const double ln = lp_new[kk];
        bool accept = false;
        if (ln != ln) {
            atomicExch(nanflag, 1);
        } else {
            const Philox4 r = draw(seed, step, half, w, 1u);
            const double u = u01(r.v[0], r.v[1]);
            const double lnq = (double)(D - 1) * log(zz[kk]) + ln - lp_own[w];
            accept = log(u) < lnq;
        }
        const double* __restrict__ y = prop + (size_t)kk * D;
        if (chain_pos) {                                  // (before the row is overwritten below)
            for (int d = 0; d < D; ++d) chain_pos[(size_t)w * D + d] = accept ? y[d] : pos_own[(size_t)w * D + d];
            chain_lp[w] = accept ? ln : lp_own[w];
        }
        if (accept) {
            for (int rr = 0; rr < R.n; ++rr) {
                double* __restrict__ x = R.pos[rr] + (size_t)w * D;
                for (int d = 0; d < D; ++d) x[d] = y[d];
                R.lp[rr][w] = ln;
            }
            nacc[w] += 1;
        }
    }
    if (R.sync) {
        __syncthreads();
        if (threadIdx.x == 0) replicas_publish(R, gridDim.x, blockIdx.x);
    }
}

// Flags a NaN in a lnprob vector (the initial state of a run: emcee and the host sampler raise on it).
__global__ void nan_flag_kernel(const double* __restrict__ lp, int W, int* __restrict__ nanflag) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w < W && lp[w] != lp[w]) atomicExch(nanflag, 1);
}

}  // namespace vp
