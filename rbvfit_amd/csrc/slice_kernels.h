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
//     OUT    (expand the left edge while lnprob(edge) > Z0 and budget J > 0; same on the right, budget K)
//  -> SHRINK (draw inside [L, R]; accept when lnprob > Z0, else pull the bracket in)
//  -> DONE
// (the two edges of OUT are independent and are worked on side by side) and ONE lnprob batch per round
// evaluates the next trial points of every walker that is not DONE.  The batch always has W rows
// (twice the half-ensemble): the active walkers are compacted to the front and share the rows evenly
// -- 2 candidates each while all are active, up to 8 for the last stragglers (slice_emit) --, the rest
// is filled with +inf, which the box prior turns into -inf without evaluating the model (so the tile
// workgroups of filler rows exit at once and the tile geometry -- hence the last bit of every lnprob --
// does not depend on how many walkers are still active).
//
// Randomness: Philox4x32-10 keyed by (seed; walker, step, half, purpose) like the stretch move, so a
// run is reproducible, splittable into calls, and can be replayed on the host draw by draw
// (tests/test_gpu_sampler.py).  Purposes: 16 permutation key, 17 partners, 18 height and bracket
// position, 19 expansion budgets, 32 + c the c-th shrink draw of the walker in this half-step.
#pragma once
#include "sampler_kernels.h"

namespace vp {

constexpr int SL_OUT = 0, SL_SHRINK = 2, SL_DONE = 3;
constexpr int SLICE_MAX_HALF = 2048;    // one workgroup of <= 1024 threads handles a half-ensemble, up to SLICE_KPT walkers per thread (W <= 4096)
constexpr int SLICE_KPT = 2;
#ifndef VP_SLICE_MAXC
#define VP_SLICE_MAXC 8
#endif
constexpr int SLICE_MAXC = VP_SLICE_MAXC;   // candidates a walker may have in one round

struct SliceState {          // per walker-slot k of the active half (device arrays of length half)
    double* X0;              // (half, D) position at the start of the half-step
    double* eta;             // (half, D) direction
    double* Z0;              // slice height
    double* L;               // bracket
    double* R;
    double* T;               // (half, SLICE_MAXC) slice parameters of the candidates of the current round
    int* J;                  // expansion budgets left / right
    int* K;
    int* phase;              // SL_OUT | SL_SHRINK | SL_DONE
    int* sides;              // SL_OUT: bit 0 = left edge settled, bit 1 = right edge settled
    int* nshr;               // shrink draws used so far
    int* row;                // rank of the walker among the active ones (its candidates are rows row*nc ... of the batch), -1 when DONE
    int* widx;               // walker index w = perm[h * half + k]
};

struct SliceCounters {       // device scalars
    int* n_active;           // walkers that are not DONE (after the last update); [1] = nanflag, [2] = candidates per walker of the batch
    long long* n_evals;      // trial points the sequential algorithm would have evaluated so far
    long long* nexp;         // expansions / contractions of the current iteration (mu tuning)
    long long* ncon;
    int* nanflag;
    int* ncand;
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

// The next round's batch.  Every walker that is not DONE gets `nc` rows (nc even, 2 <= nc <= SLICE_MAXC, the more
// the fewer walkers are left: nc = batch rows / active walkers) holding the NEXT nc trial points the sequential
// procedure could ask for, in its order, under the assumption that each one before them comes out the way that keeps
// the procedure going:
//   SL_OUT     nc/2 per side: the left edge at L, L-1, ... (as long as the budget J lasts) and the right edge at
//              R, R+1, ...: an edge expands while lnprob(edge) > Z0, so candidate s is only looked at if 0..s-1 expanded;
//   SL_SHRINK  the shrink draws c, c+1, ...: Wd = L' + u (R' - L') with the bracket pulled in behind every earlier
//              draw -- which side it is pulled in from depends on the sign of the draw alone, not on its lnprob, so the
//              whole rejected-so-far sequence is known in advance; the first draw with lnprob > Z0 is accepted.
// The update kernel consumes the results in order and stops at the first one that ends the phase, so the accepted
// point, the brackets, the expansion / contraction counts (hence mu) and the count of evaluations are exactly those
// of the one-evaluation-at-a-time procedure; speculation only buys rounds (latency) with throughput.
// Rows of invalid candidates and of DONE walkers hold +inf, which the box prior rejects without evaluating the model.
__device__ inline void slice_emit(int half, const SliceState& st, int D, int batch_rows, uint64_t seed, uint64_t step,
                                  int h, double* __restrict__ trial, const SliceCounters& cn, int* lds_counts) {
    // thread t looks after the walkers k = t, t + blockDim, ...: their ranks among the active ones run through the
    // passes in that order (rank = active walkers with a smaller k)
    bool active[SLICE_KPT];
    int a[SLICE_KPT], total = 0;
    for (int r = threadIdx.x; r < batch_rows; r += blockDim.x) {
        double* row = trial + (size_t)r * D;
        for (int d = 0; d < D; ++d) row[d] = __builtin_inf();
    }
#pragma unroll
    for (int p = 0; p < SLICE_KPT; ++p) {
        const int k = threadIdx.x + p * blockDim.x;
        active[p] = k < half && st.phase[k] != SL_DONE;
        int tp;
        a[p] = total + block_scan01(active[p], lds_counts, &tp);   // (contains barriers: the filler is complete behind it)
        total += tp;
    }
    int nc = total > 0 ? (batch_rows / total) & ~1 : 2;
    nc = nc < 2 ? 2 : (nc > SLICE_MAXC ? SLICE_MAXC : nc);
#pragma unroll
    for (int p = 0; p < SLICE_KPT; ++p) {
        const int k = threadIdx.x + p * blockDim.x;
        if (k < half) st.row[k] = active[p] ? a[p] : -1;
        if (!active[p]) continue;
        double* T = st.T + (size_t)k * SLICE_MAXC;
        bool valid[SLICE_MAXC];
        if (st.phase[k] == SL_OUT) {
            const int hs = nc >> 1, sd = st.sides[k], J = st.J[k], K = st.K[k];
            const double L = st.L[k], R = st.R[k];
#pragma unroll
            for (int s = 0; s < SLICE_MAXC; ++s) {
                const bool left = s < hs;
                const int i = left ? s : s - hs;
                T[s] = left ? L - (double)i : R + (double)i;
                valid[s] = s < nc && (left ? (!(sd & 1) && J > i) : (!(sd & 2) && K > i));
            }
        } else {
            double Lc = st.L[k], Rc = st.R[k];
            const int c0 = st.nshr[k];
#pragma unroll
            for (int s = 0; s < SLICE_MAXC; ++s) {
                valid[s] = s < nc;
                if (s < nc) {
                    const Philox4 r = draw(seed, step, h, st.widx[k], 32u + (uint32_t)(c0 + s));
                    const double wd = Lc + u01(r.v[0], r.v[1]) * (Rc - Lc);
                    T[s] = wd;
                    if (wd < 0.0) Lc = wd; else Rc = wd;
                }
            }
        }
        const double* x0 = st.X0 + (size_t)k * D;
        const double* e = st.eta + (size_t)k * D;
#pragma unroll
        for (int s = 0; s < SLICE_MAXC; ++s) {
            if (s < nc && valid[s]) {
                double* row = trial + ((size_t)a[p] * nc + s) * D;
                const double t = T[s];
                for (int d = 0; d < D; ++d) row[d] = x0[d] + t * e[d];
            }
        }
    }
    if (threadIdx.x == 0) {
        *cn.n_active = total;
        *cn.ncand = nc;
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
                                                          int batch_rows, SliceState st, SliceCounters cn,
                                                          double* __restrict__ trial) {
    __shared__ int lds_counts[16];
    for (int k = threadIdx.x; k < half; k += blockDim.x) {
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
        const int K = (maxsteps - 1) - J;
        st.J[k] = J;
        st.K[k] = K;
        const int sd = (J <= 0 ? 1 : 0) | (K <= 0 ? 2 : 0);
        st.sides[k] = sd;
        st.phase[k] = sd == 3 ? SL_SHRINK : SL_OUT;
        st.nshr[k] = 0;
    }
    slice_emit(half, st, D, batch_rows, seed, step, h, trial, cn, lds_counts);
}

// After a round's lnprob batch: every walker that is not DONE consumes the results of its candidates in order, up to
// the first one that ends its phase (an accepted shrink draw updates the walker's position and lnprob in place),
// then the next batch is written.
__global__ __launch_bounds__(1024) void slice_update_kernel(double* __restrict__ pos, double* __restrict__ lp,
                                                            const double* __restrict__ lnp_rows, int half, int D, int h,
                                                            uint64_t seed, uint64_t step, int batch_rows, SliceState st,
                                                            SliceCounters cn, double* __restrict__ trial) {
    __shared__ int lds_counts[16];
    __shared__ int s_exp, s_con, s_ev;
    if (threadIdx.x == 0) { s_exp = 0; s_con = 0; s_ev = 0; }
    __syncthreads();
    const int nc = *cn.ncand;
    for (int k = threadIdx.x; k < half; k += blockDim.x) {
        if (st.row[k] < 0) continue;
        const double* res = lnp_rows + (size_t)st.row[k] * nc;
        const double* T = st.T + (size_t)k * SLICE_MAXC;
        const double z0 = st.Z0[k];
        int nev = 0, nex = 0, nco = 0;
        bool nan = false;
        if (st.phase[k] == SL_OUT) {
            const int hs = nc >> 1;
            int sd = st.sides[k], J = st.J[k], K = st.K[k];
            double L = st.L[k], R = st.R[k];
            for (int s = 0; s < hs && !(sd & 1); ++s) {            // left edge: expands while lnprob(edge) > Z0 and budget lasts
                const double v = res[s];
                ++nev;
                if (v != v) { nan = true; break; }
                if (v > z0) { L -= 1.0; J -= 1; ++nex; if (J <= 0) sd |= 1; }
                else sd |= 1;
            }
            for (int s = 0; s < hs && !(sd & 2) && !nan; ++s) {    // right edge
                const double v = res[hs + s];
                ++nev;
                if (v != v) { nan = true; break; }
                if (v > z0) { R += 1.0; K -= 1; ++nex; if (K <= 0) sd |= 2; }
                else sd |= 2;
            }
            st.L[k] = L; st.R[k] = R; st.J[k] = J; st.K[k] = K; st.sides[k] = sd;
            if (sd == 3) st.phase[k] = SL_SHRINK;
        } else {
            double L = st.L[k], R = st.R[k];
            int c = st.nshr[k];
            for (int s = 0; s < nc; ++s) {
                const double v = res[s], wd = T[s];
                ++nev;
                if (v != v) { nan = true; break; }
                if (v > z0) {                                      // accept: the trial point (re-formed: same bits) is the new position
                    const int w = st.widx[k];
                    const double* x0 = st.X0 + (size_t)k * D;
                    const double* e = st.eta + (size_t)k * D;
                    for (int d = 0; d < D; ++d) pos[(size_t)w * D + d] = x0[d] + wd * e[d];
                    lp[w] = v;
                    st.phase[k] = SL_DONE;
                    break;
                }
                if (wd < 0.0) L = wd; else R = wd;
                ++c; ++nco;
            }
            st.L[k] = L; st.R[k] = R; st.nshr[k] = c;
        }
        if (nan) {
            atomicExch(cn.nanflag, 1);
            st.phase[k] = SL_DONE;
        }
        atomicAdd(&s_ev, nev); atomicAdd(&s_exp, nex); atomicAdd(&s_con, nco);
    }
    __syncthreads();
    if (threadIdx.x == 0) { *cn.nexp += s_exp; *cn.ncon += s_con; *cn.n_evals += s_ev; }
    slice_emit(half, st, D, batch_rows, seed, step, h, trial, cn, lds_counts);
}

}  // namespace vp
