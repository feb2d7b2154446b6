// Device-resident ensemble slice sampler (second move): the walker loop the reference
// delegates to zeus when built with sampler='zeus' (vfit_mcmc.py:425-440, 536-540) -- ensemble slice
// sampling with the differential move (Karamanis, Beutler & Peacock 2021).  Positions, lnprob, slice
// brackets and the ragged sets of still-active walkers stay in HBM: the host never sees a
// stepping-out or shrinking round, it only learns (once per group of rounds) whether any walker of the
// half-ensemble is still active.
//
// Per iteration: a random split of the ensemble into two halves; each walker k of the active half
// slices along eta_k = mu * 2.38/sqrt(2D) * (X_l - X_m), l != m drawn from the other half, with
// height Z0 = lnprob(X_k) + ln u.  Every walker runs its own little state machine
//     OUT_L (expand the left edge while lnprob(edge) > Z0 and budget J > 0)
//  -> OUT_R (same on the right, budget K)
//  -> SHRINK (draw inside [L, R]; accept when lnprob > Z0, else pull the bracket in)
//  -> DONE
// and ONE lnprob batch per round evaluates the pending trial point of every walker that is not
// DONE, whatever its phase.  The batch always has `half` rows: the active walkers' trial points are
// compacted to the front, the rest is filled with +inf, which the box prior turns into -inf without
// evaluating the model (so the tile workgroups of filler rows exit at once and the tile geometry --
// hence the last bit of every lnprob -- does not depend on how many walkers are still active).
//
// Randomness: Philox4x32-10 keyed by (seed; walker, step, half, purpose) like the stretch move, so a
// run is reproducible, splittable into calls, and can be replayed on the host draw by draw
// (tests/test_gpu_sampler.py).  Purposes: 16 permutation key, 17 partners, 18 height and bracket
// position, 19 expansion budgets, 32 + c the c-th shrink draw of the walker in this half-step.
#pragma once
#include "sampler_kernels.h"

namespace vp {

constexpr int SL_OUT_L = 0, SL_OUT_R = 1, SL_SHRINK = 2, SL_DONE = 3;
constexpr int SLICE_MAX_HALF = 1024;    // one workgroup handles a half-ensemble (W <= 2048)

struct SliceState {          // per walker-slot k of the active half (device arrays of length half)
    double* X0;              // (half, D) position at the start of the half-step
    double* eta;             // (half, D) direction
    double* Z0;              // slice height
    double* L;               // bracket
    double* R;
    double* Wd;              // pending shrink draw
    int* J;                  // expansion budgets left / right
    int* K;
    int* phase;
    int* nshr;               // shrink draws used so far
    int* row;                // row of the pending trial point in the batch, -1 when none
    int* widx;               // walker index w = perm[h * half + k]
};

struct SliceCounters {       // device scalars
    int* n_active;           // walkers with a pending trial point (after the last advance)
    long long* n_evals;      // trial points evaluated so far
    long long* nexp;         // expansions / contractions of the current iteration (mu tuning)
    long long* ncon;
    int* nanflag;
    double* mu;              // [0] mu, [1] number of consecutive in-tolerance iterations, [2] tuning on (1.0) / off (0.0)
};

// Exclusive prefix sum of a 0/1 flag over the (<= 1024) threads of the workgroup; `total` gets the sum.
__device__ inline int block_scan01(bool flag, int* lds_counts, int* total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    const unsigned long long m = __ballot(flag);
    const int within = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) lds_counts[wid] = __popcll(m);
    __syncthreads();
    int base = 0, tot = 0;
    for (int i = 0; i < nw; ++i) {
        const int c = lds_counts[i];
        if (i < wid) base += c;
        tot += c;
    }
    __syncthreads();
    *total = tot;
    return base + within;
}

// Pending trial point X0 + t eta of slot k in its current phase (advancing through exhausted phases): sets t;
// false when DONE.
__device__ inline bool slice_trial(int k, const SliceState& st, uint64_t seed, uint64_t step, int h, double* tout) {
    int ph = st.phase[k];
    if (ph == SL_OUT_L && st.J[k] <= 0) ph = SL_OUT_R;
    if (ph == SL_OUT_R && st.K[k] <= 0) ph = SL_SHRINK;
    st.phase[k] = ph;
    if (ph == SL_DONE) return false;
    double t;
    if (ph == SL_OUT_L) t = st.L[k];
    else if (ph == SL_OUT_R) t = st.R[k];
    else {
        const Philox4 r = draw(seed, step, h, st.widx[k], 32u + (uint32_t)st.nshr[k]);
        const double u = u01(r.v[0], r.v[1]);
        t = st.L[k] + u * (st.R[k] - st.L[k]);
        st.Wd[k] = t;
    }
    *tout = t;
    return true;
}

// Writes the batch for the next round: active slots' trial points compacted to the front, +inf filler behind.
__device__ inline void slice_emit(int k, int half, bool active, double t, const SliceState& st, int D,
                                  double* __restrict__ trial, const SliceCounters& cn, int* lds_counts) {
    if (k < half) {
        double* row = trial + (size_t)k * D;
        for (int d = 0; d < D; ++d) row[d] = __builtin_inf();
    }
    int total;
    const int r = block_scan01(active, lds_counts, &total);       // (contains barriers: the filler is complete behind it)
    if (k < half) st.row[k] = active ? r : -1;
    if (active) {
        double* row = trial + (size_t)r * D;
        const double* x0 = st.X0 + (size_t)k * D;
        const double* e = st.eta + (size_t)k * D;
        for (int d = 0; d < D; ++d) row[d] = x0[d] + t * e[d];
    }
    if (threadIdx.x == 0) {
        *cn.n_active = total;
        *cn.n_evals += total;
    }
}

// mu tuning from one iteration's counts (zeus' rule: mu *= 2 nexp / (nexp + ncon), switched off after `patience`
// consecutive iterations within `tolerance` of 1); the counts are cleared for the next iteration.
__device__ inline void slice_tune(const SliceCounters& cn, double tolerance, int patience, double* mu_hist) {
    if (cn.mu[2] != 0.0) {
        const double ne = (double)(*cn.nexp > 0 ? *cn.nexp : 1), nc = (double)*cn.ncon;
        const double ratio = 2.0 * ne / (ne + nc);
        cn.mu[0] *= ratio;
        cn.mu[1] = (fabs(ratio - 1.0) < tolerance) ? cn.mu[1] + 1.0 : 0.0;
        if (cn.mu[1] >= (double)patience) cn.mu[2] = 0.0;
    }
    if (mu_hist) *mu_hist = cn.mu[0];
    *cn.nexp = 0;
    *cn.ncon = 0;
}
__global__ void slice_tune_kernel(SliceCounters cn, double tolerance, int patience, double* mu_hist) {
    if (threadIdx.x == 0 && blockIdx.x == 0) slice_tune(cn, tolerance, patience, mu_hist);
}

// Start of an iteration: the tuning step of the previous iteration, then a random permutation of the walkers by
// ranking one 64-bit Philox key per walker (ties by index).
__global__ __launch_bounds__(1024) void slice_begin_kernel(int W, uint64_t seed, uint64_t step, int* __restrict__ perm,
                                                           SliceCounters cn, int tune_now, double tolerance, int patience,
                                                           double* __restrict__ mu_hist) {
    __shared__ unsigned long long keys[2 * SLICE_MAX_HALF];
    if (threadIdx.x == 0 && tune_now) slice_tune(cn, tolerance, patience, mu_hist);
    for (int w = threadIdx.x; w < W; w += blockDim.x) {
        const Philox4 r = draw(seed, step, 0, w, 16u);
        keys[w] = ((unsigned long long)r.v[0] << 32) | r.v[1];
    }
    __syncthreads();
    for (int w = threadIdx.x; w < W; w += blockDim.x) {
        const unsigned long long kw = keys[w];
        int rank = 0;
        for (int v = 0; v < W; ++v) {
            const unsigned long long kv = keys[v];
            rank += (kv < kw) || (kv == kw && v < w);
        }
        perm[rank] = w;
    }
}

// Start of a half-step (half h: active walkers perm[h*half ...], complementary the other half): directions,
// heights, brackets, budgets, and the first batch.
__global__ __launch_bounds__(1024) void slice_init_kernel(const double* __restrict__ pos, const double* __restrict__ lp,
                                                          const int* __restrict__ perm, int half, int D, int h,
                                                          uint64_t seed, uint64_t step, double gamma0, int maxsteps,
                                                          SliceState st, SliceCounters cn, double* __restrict__ trial) {
    __shared__ int lds_counts[16];
    const int k = threadIdx.x;
    double t = 0.0;
    bool active = false;
    if (k < half) {
        const int* S = perm + h * half;
        const int* C = perm + (1 - h) * half;
        const int w = S[k];
        st.widx[k] = w;
        const Philox4 ra = draw(seed, step, h, w, 17u), rb = draw(seed, step, h, w, 18u), rc = draw(seed, step, h, w, 19u);
        int l = (int)(u01(ra.v[0], ra.v[1]) * (double)half);
        l = l < half - 1 ? l : half - 1;
        int mo = (int)(u01(ra.v[2], ra.v[3]) * (double)(half - 1));
        mo = mo < half - 2 ? mo : half - 2;
        const int m = (l + 1 + mo) % half;                         // a second, different partner
        const double* xl = pos + (size_t)C[l] * D;
        const double* xm = pos + (size_t)C[m] * D;
        const double* x = pos + (size_t)w * D;
        const double s = cn.mu[0] * gamma0;
        for (int d = 0; d < D; ++d) {
            st.X0[(size_t)k * D + d] = x[d];
            st.eta[(size_t)k * D + d] = s * (xl[d] - xm[d]);
        }
        st.Z0[k] = lp[w] + log(u01(rb.v[0], rb.v[1]));             // lnprob - Exp(1)
        const double L = -u01(rb.v[2], rb.v[3]);
        st.L[k] = L;
        st.R[k] = L + 1.0;
        int J = (int)((double)maxsteps * u01(rc.v[0], rc.v[1]));
        J = J < maxsteps - 1 ? J : maxsteps - 1;
        st.J[k] = J;
        st.K[k] = (maxsteps - 1) - J;
        st.phase[k] = SL_OUT_L;
        st.nshr[k] = 0;
        active = slice_trial(k, st, seed, step, h, &t);
    }
    slice_emit(k, half, active, t, st, D, trial, cn, lds_counts);
}

// After a round's lnprob batch: every slot with a pending trial point takes its result, moves its state machine
// (an accepted shrink draw updates the walker's position and lnprob in place), and the next batch is written.
__global__ __launch_bounds__(1024) void slice_update_kernel(double* __restrict__ pos, double* __restrict__ lp,
                                                            const double* __restrict__ lnp_rows, int half, int D, int h,
                                                            uint64_t seed, uint64_t step, SliceState st, SliceCounters cn,
                                                            double* __restrict__ trial) {
    __shared__ int lds_counts[16];
    __shared__ int s_exp, s_con;
    const int k = threadIdx.x;
    if (k == 0) { s_exp = 0; s_con = 0; }
    __syncthreads();
    double t = 0.0;
    bool active = false;
    if (k < half && st.row[k] >= 0) {
        const int r = st.row[k];
        const double v = lnp_rows[r];
        const double z0 = st.Z0[k];
        const int ph = st.phase[k];
        if (v != v) {
            atomicExch(cn.nanflag, 1);
            st.phase[k] = SL_DONE;
        } else if (ph == SL_OUT_L) {
            if (v > z0) { st.L[k] -= 1.0; st.J[k] -= 1; atomicAdd(&s_exp, 1); }
            else st.phase[k] = SL_OUT_R;
        } else if (ph == SL_OUT_R) {
            if (v > z0) { st.R[k] += 1.0; st.K[k] -= 1; atomicAdd(&s_exp, 1); }
            else st.phase[k] = SL_SHRINK;
        } else {
            if (v > z0) {                                          // accept: the trial point (re-formed: same bits) is the new position
                const int w = st.widx[k];
                const double wd = st.Wd[k];
                const double* x0 = st.X0 + (size_t)k * D;
                const double* e = st.eta + (size_t)k * D;
                for (int d = 0; d < D; ++d) pos[(size_t)w * D + d] = x0[d] + wd * e[d];
                lp[w] = v;
                st.phase[k] = SL_DONE;
            } else {
                const double wd = st.Wd[k];
                if (wd < 0.0) st.L[k] = wd; else st.R[k] = wd;
                st.nshr[k] += 1;
                atomicAdd(&s_con, 1);
            }
        }
        active = slice_trial(k, st, seed, step, h, &t);
    }
    slice_emit(k, half, active, t, st, D, trial, cn, lds_counts);
    if (k == 0) { *cn.nexp += s_exp; *cn.ncon += s_con; }
}

}  // namespace vp
