// Device-resident ensemble slice sampler (second move): the walker loop the reference
// delegates to zeus when built with sampler='zeus' (vfit_mcmc.py:425-440, 536-540) -- ensemble slice
// sampling with the differential move (Karamanis, Beutler & Peacock 2021).  Positions, lnprob, slice
// brackets, the ragged sets of still-active walkers AND the loop itself live on the GPU: the host enqueues
// (lnprob batch, slice_round_kernel) pairs and the round kernel decides what the next batch is -- more
// stepping-out / shrinking candidates, the other half-ensemble's start, the next iteration (chain row, mu
// tuning) or nothing more.  The host never waits inside a segment of iterations; it reads two words the
// kernel keeps up to date in mapped host memory (rounds consumed, done) to stay a few rounds ahead.
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
// -- 2 candidates each while all are active, up to 8 for the last stragglers --, the rest
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

// Everything the walker loop needs between two lnprob batches (slice_round_kernel): the host only enqueues
// (lnprob batch, slice_round_kernel) pairs and watches two words the kernel writes into mapped host memory.
struct SliceRun {
    double* pos;             // (W, D) ensemble, updated in place
    double* lp;              // (W)
    double* trial;           // (batch_rows, D) the next lnprob batch
    const double* trial_in;  // (batch_rows, D) the batch whose results this launch consumes (the two alternate)
    const int* perm_tab;     // (n, W) the random split of every iteration of this segment (slice_perm_kernel)
    int* prog;               // device: [0] iterations finished, [1] half, [2] done, [3] error (1 NaN, 2 no termination),
                             //         [4] rounds of this half-step, [5] rounds consumed in this segment
    int* host;               // mapped host memory (or NULL): [0] rounds consumed, [1] 0 running / 1 done / 2 error
    double* chain;           // (n, W, D) / (n, W) positions and lnprob behind every iteration, or NULL
    double* chain_lp;
    double* mu_hist;         // (n) mu behind every iteration's tuning step
    int W, half, D, batch_rows, n, maxsteps, patience, round_limit;
    double gamma0, tolerance;
    uint64_t seed, step_base;   // iteration `it` of the segment is step step_base + it of the run
};

// Workgroup barrier behind which the waves' LDS traffic (only) is complete: what is in flight to or from global memory
// stays in flight.  The round kernel is a chain of memory round trips; __syncthreads() would end each of them early.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Ranks of the flagged walker slots (slot k = thread + p * blockDim, p < SLICE_KPT; rank = flagged slots with a smaller k)
// and their number, over the (<= 1024) threads of the workgroup.
__device__ inline int block_rank(const bool (&flag)[SLICE_KPT], int (&rank)[SLICE_KPT], int (*lds_counts)[16], int half) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    int within[SLICE_KPT];
#pragma unroll
    for (int p = 0; p < SLICE_KPT; ++p) {
        within[p] = 0;
        if (p * (int)blockDim.x >= half) continue;   // (uniform) no walker slot in this pass
        const unsigned long long m = __ballot(flag[p]);
        within[p] = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) lds_counts[p][wid] = __popcll(m);
    }
    lds_barrier();
    int total = 0;
#pragma unroll
    for (int p = 0; p < SLICE_KPT; ++p) {
        int base = 0, tot = 0;
        rank[p] = total;
        if (p * (int)blockDim.x >= half) continue;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = i < nw ? lds_counts[p][i] : 0;
            if (i < wid) base += c;
            tot += c;
        }
        rank[p] = total + base + within[p];
        total += tot;
    }
    return total;
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

// The random splits of a segment's iterations, one workgroup per iteration: a permutation of the walkers by ranking
// one 64-bit Philox key per walker (ties by index); the first half of it moves first.
__global__ __launch_bounds__(1024) void slice_perm_kernel(int W, uint64_t seed, uint64_t step_base, int* __restrict__ perm_tab) {
    __shared__ unsigned long long keys[2 * SLICE_MAX_HALF];
    const uint64_t step = step_base + blockIdx.x;
    int* perm = perm_tab + (size_t)blockIdx.x * W;
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

#ifdef VP_STAMPS
// diagnostic build only: thread 0's clock at the phases of the round kernels of a segment (round, stage)
constexpr int SLICE_STAMP_ROUNDS = 256, SLICE_STAMP_STAGES = 16;
__device__ long long g_slice_stamps[SLICE_STAMP_ROUNDS * SLICE_STAMP_STAGES];
#define SL_STAMP(stage) do { if (threadIdx.x == 0 && sl_round < SLICE_STAMP_ROUNDS) \
    g_slice_stamps[sl_round * SLICE_STAMP_STAGES + (stage)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define SL_STAMP(stage) do { } while (0)
#endif

// A walker slot's state in registers for the length of a round kernel (loaded once, stored once).
struct SliceW { int row, phase, sides, J, K, nshr, widx; double Z0, L, R; };
__device__ inline SliceW slice_load(const SliceState& st, int k) {
    SliceW w;
    w.row = st.row[k]; w.phase = st.phase[k]; w.sides = st.sides[k]; w.J = st.J[k]; w.K = st.K[k]; w.nshr = st.nshr[k];
    w.widx = st.widx[k]; w.Z0 = st.Z0[k]; w.L = st.L[k]; w.R = st.R[k];
    return w;
}
__device__ inline void slice_store(const SliceState& st, int k, const SliceW& w) {
    st.row[k] = w.row; st.phase[k] = w.phase; st.sides[k] = w.sides; st.J[k] = w.J; st.K[k] = w.K; st.nshr[k] = w.nshr;
    st.widx[k] = w.widx; st.Z0[k] = w.Z0; st.L[k] = w.L; st.R[k] = w.R;
}

// Start of a half-step (half h: active walkers perm[h*half ...], complementary the other half): directions,
// heights, brackets, budgets of walker slot k.
__device__ inline SliceW slice_init_walker(const SliceRun& P, const SliceState& st, const SliceCounters& cn, const int* __restrict__ perm,
                                           int h, uint64_t step, int k) {
    const int half = P.half, D = P.D;
    const int* __restrict__ S = perm + h * half;
    const int* __restrict__ C = perm + (1 - h) * half;
    SliceW ws;
    const int w = S[k];
    ws.widx = w;
    const Philox4 ra = draw(P.seed, step, h, w, 17u);
    int l = (int)(u01(ra.v[0], ra.v[1]) * (double)half);
    l = l < half - 1 ? l : half - 1;
    int mo = (int)(u01(ra.v[2], ra.v[3]) * (double)(half - 1));
    mo = mo < half - 2 ? mo : half - 2;
    const int m = (l + 1 + mo) % half;                         // a second, different partner
    const int wl = C[l], wm = C[m];                            // (on their way while the other two blocks are drawn)
    const Philox4 rb = draw(P.seed, step, h, w, 18u), rc = draw(P.seed, step, h, w, 19u);
    const double* __restrict__ xl = P.pos + (size_t)wl * D;
    const double* __restrict__ xm = P.pos + (size_t)wm * D;
    const double* __restrict__ x = P.pos + (size_t)w * D;
    double* __restrict__ X0 = st.X0 + (size_t)k * D;
    double* __restrict__ eta = st.eta + (size_t)k * D;
    const double s = cn.mu[0] * P.gamma0;
    for (int d = 0; d < D; ++d) {
        X0[d] = x[d];
        eta[d] = s * (xl[d] - xm[d]);
    }
    ws.Z0 = P.lp[w] + log(u01(rb.v[0], rb.v[1]));              // lnprob - Exp(1)
    ws.L = -u01(rb.v[2], rb.v[3]);
    ws.R = ws.L + 1.0;
    int J = (int)((double)P.maxsteps * u01(rc.v[0], rc.v[1]));
    J = J < P.maxsteps - 1 ? J : P.maxsteps - 1;
    ws.J = J;
    ws.K = (P.maxsteps - 1) - J;
    ws.sides = (ws.J <= 0 ? 1 : 0) | (ws.K <= 0 ? 2 : 0);
    ws.phase = ws.sides == 3 ? SL_SHRINK : SL_OUT;
    ws.nshr = 0;
    ws.row = -1;
    return ws;
}

// One walker's share of a round's results (v: its nc candidates in order; vr: the right edge's, v[nc/2 ...]; T: their slice
// parameters): the candidates in order, up to the first one that ends its phase.  Returns the index of an accepted shrink draw
// (its lnprob in `vacc`) or -1.
__device__ inline int slice_consume(SliceW& w, int nc, const double (&v)[SLICE_MAXC], const double (&vr)[SLICE_MAXC / 2],
                                    const double (&T)[SLICE_MAXC], int& nev, int& nex, int& nco, bool& nan, double& vacc) {
    const int hs = nc >> 1;
    const double z0 = w.Z0;
    int acc = -1;
    if (w.phase == SL_OUT) {
#pragma unroll
        for (int s = 0; s < SLICE_MAXC / 2; ++s) {                 // left edge: expands while lnprob(edge) > Z0 and budget lasts
            if (s < hs && !(w.sides & 1) && !nan) {
                ++nev;
                if (v[s] != v[s]) nan = true;
                else if (v[s] > z0) { w.L -= 1.0; w.J -= 1; ++nex; if (w.J <= 0) w.sides |= 1; }
                else w.sides |= 1;
            }
        }
#pragma unroll
        for (int s = 0; s < SLICE_MAXC / 2; ++s) {                 // right edge
            if (s < hs && !(w.sides & 2) && !nan) {
                ++nev;
                if (vr[s] != vr[s]) nan = true;
                else if (vr[s] > z0) { w.R += 1.0; w.K -= 1; ++nex; if (w.K <= 0) w.sides |= 2; }
                else w.sides |= 2;
            }
        }
        if (w.sides == 3) w.phase = SL_SHRINK;
    } else {
        bool open = true;
#pragma unroll
        for (int s = 0; s < SLICE_MAXC; ++s) {
            if (s < nc && open) {
                ++nev;
                if (v[s] != v[s]) { nan = true; open = false; }
                else if (v[s] > z0) { acc = s; vacc = v[s]; open = false; w.phase = SL_DONE; }
                else { if (T[s] < 0.0) w.L = T[s]; else w.R = T[s]; ++w.nshr; ++nco; }
            }
        }
    }
    if (nan) w.phase = SL_DONE;
    return acc;
}

// Between two lnprob batches, ONE workgroup of 1024 threads (thread t looks after walker slots t, t + 1024):
//   consume   every walker that is not DONE takes the results of its candidates (slice_consume); an accepted shrink draw
//             becomes the walker's position: its row of the batch, copied;
//   advance   when no walker of the half is left: the other half's start (slice_init_walker), and behind the second half
//             the iteration's end -- chain row, mu tuning -- and the next iteration's first half; behind the segment's last
//             iteration the kernel marks the run done and every later launch of it returns at once;
//   emit      the next batch.  Every walker that is not DONE gets `nc` rows (nc even, 2 <= nc <= SLICE_MAXC, the more the
//             fewer walkers are left: nc = batch rows / active walkers) holding the NEXT nc trial points the sequential
//             procedure could ask for, in its order, under the assumption that each one before them comes out the way
//             that keeps the procedure going:
//               SL_OUT     nc/2 per side: the left edge at L, L-1, ... (as long as the budget J lasts) and the right edge
//                          at R, R+1, ...: an edge expands while lnprob(edge) > Z0, so candidate s is only looked at if
//                          0..s-1 expanded;
//               SL_SHRINK  the shrink draws c, c+1, ...: Wd = L' + u (R' - L') with the bracket pulled in behind every
//                          earlier draw -- which side it is pulled in from depends on the sign of the draw alone, not on
//                          its lnprob, so the whole rejected-so-far sequence is known in advance; the first draw with
//                          lnprob > Z0 is accepted.
//             Consumption stops at the first result that ends the phase, so the accepted point, the brackets, the
//             expansion / contraction counts (hence mu) and the count of evaluations are exactly those of the
//             one-evaluation-at-a-time procedure; speculation only buys rounds (latency) with throughput.  Rows of
//             invalid candidates and of DONE walkers hold +inf, which the box prior rejects without evaluating the model.
// The kernel is a chain of memory round trips on one CU, so it is written to have few of them: a walker's state is
// loaded once (with everything else that does not depend on it) and lives in registers, the results follow in a
// second trip, the uniform draws of the shrink candidates (a Philox block each) and the rows themselves are spread
// over all threads (by batch row, through LDS), and only the rows' X0 / eta are a third.
// start = 1: the segment's first launch (no results to consume yet).
__global__ __launch_bounds__(1024) void slice_round_kernel(SliceRun P, SliceState st, SliceCounters cn,
                                                           const double* __restrict__ lnp_rows, int start) {
    __shared__ int lds_counts[SLICE_KPT][16];
    __shared__ int s_exp, s_con, s_ev, s_nan;
    __shared__ double t_lds[2 * SLICE_MAX_HALF];     // per batch row: the uniform draw, then the slice parameter (NaN: no candidate)
    __shared__ int o_k[SLICE_MAX_HALF];              // rank among the active walkers -> walker slot,
    __shared__ int o_w[SLICE_MAX_HALF];              //   its walker index,
    __shared__ int o_c[SLICE_MAX_HALF];              //   its count of shrink draws so far (-1: stepping out)
    const int tid = threadIdx.x, nt = blockDim.x;
    const int half = P.half, D = P.D;
    int it = 0, h = 0, rounds_half = 0, rounds_seg = 0;
    bool active[SLICE_KPT];
    int a[SLICE_KPT], total = 0, err = 0;
    bool fresh = start != 0, finished = false;
    SliceW ws[SLICE_KPT];
#ifdef VP_STAMPS
    int sl_round = SLICE_STAMP_ROUNDS;
    if (!start) sl_round = P.prog[5];
#endif
    SL_STAMP(0);
    long long c_exp = 0, c_con = 0, c_ev = 0;
    int acc_row[SLICE_KPT];                          // an accepted shrink draw: its row of the batch just evaluated (-1: none)
    double vacc[SLICE_KPT];
#pragma unroll
    for (int p = 0; p < SLICE_KPT; ++p) { acc_row[p] = -1; vacc[p] = 0.0; }
    bool late_accept = false;
    if (!start) {
        // first trip: the program state, the counters, the candidates per walker, the walkers' state and slice parameters,
        // and the batch's results (by batch row, into LDS)
        const int p_it = P.prog[0], p_h = P.prog[1], p_done = P.prog[2], p_rh = P.prog[4], p_rs = P.prog[5];
        const int nc = *cn.ncand;
        c_exp = *cn.nexp; c_con = *cn.ncon; c_ev = *cn.n_evals;
        // (one CU's L1 serves all of this: only the threads that own a row / a walker slot load)
        double T[SLICE_KPT][SLICE_MAXC], lr[(2 * SLICE_MAX_HALF) / 1024];
#pragma unroll
        for (int j = 0; j < (2 * SLICE_MAX_HALF) / 1024; ++j) {
            lr[j] = 0.0;
            if (tid + j * 1024 < P.batch_rows) lr[j] = lnp_rows[tid + j * 1024];
        }
#pragma unroll
        for (int p = 0; p < SLICE_KPT; ++p) {
            const int k = tid + p * nt;
            ws[p] = SliceW{-1, SL_DONE, 3, 0, 0, 0, 0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < SLICE_MAXC; ++s) T[p][s] = 0.0;
            if (k < half) {
                ws[p] = slice_load(st, k);
                const double* __restrict__ Tk = st.T + (size_t)k * SLICE_MAXC;
#pragma unroll
                for (int s = 0; s < SLICE_MAXC; ++s) T[p][s] = Tk[s];
            }
        }
        it = p_it; h = p_h; rounds_half = p_rh; rounds_seg = p_rs;
        if (tid == 0) { s_exp = 0; s_con = 0; s_ev = 0; s_nan = 0; }
        SL_STAMP(1);
#pragma unroll
        for (int j = 0; j < (2 * SLICE_MAX_HALF) / 1024; ++j)
            if (tid + j * 1024 < P.batch_rows) t_lds[tid + j * 1024] = lr[j];
        lds_barrier();
        if (p_done) return;                          // (uniform) the segment is over: left-over launches do nothing
        SL_STAMP(2);
        const int hs = nc >> 1;
        int nev = 0, nex = 0, nco = 0;
        bool nan = false;
#pragma unroll
        for (int p = 0; p < SLICE_KPT; ++p) {
            const int k = tid + p * nt;
            active[p] = false;
            if (k < half && ws[p].row >= 0) {
                const double* res = t_lds + (size_t)ws[p].row * nc;
                double v[SLICE_MAXC], vr[SLICE_MAXC / 2];
#pragma unroll
                for (int s = 0; s < SLICE_MAXC; ++s) v[s] = res[s < nc ? s : nc - 1];
#pragma unroll
                for (int s = 0; s < SLICE_MAXC / 2; ++s) vr[s] = res[hs + s < nc ? hs + s : nc - 1];
                const int acc = slice_consume(ws[p], nc, v, vr, T[p], nev, nex, nco, nan, vacc[p]);
                if (acc >= 0) acc_row[p] = ws[p].row * nc + acc;
                active[p] = ws[p].phase != SL_DONE;
            }
        }
        {   // the round's counts: per wave, then one LDS atomic each
            const int wnan = __any(nan);
            const unsigned long long cs = wave_sum((unsigned long long)nev | ((unsigned long long)nex << 21) | ((unsigned long long)nco << 42));
            if ((tid & 63) == 0) {
                if (wnan) atomicExch(&s_nan, 1);
                atomicAdd(&s_ev, (int)(cs & 0x1fffff)); atomicAdd(&s_exp, (int)((cs >> 21) & 0x1fffff)); atomicAdd(&s_con, (int)(cs >> 42));
            }
        }
        total = block_rank(active, a, lds_counts, half);   // (behind its barrier the counts above are complete, too)
        ++rounds_half; ++rounds_seg;
        SL_STAMP(3);
        c_exp += s_exp; c_con += s_con; c_ev += s_ev;
        late_accept = total != 0;
        if (!late_accept) {                          // the half-step ends here: the positions are needed at once
#pragma unroll
            for (int p = 0; p < SLICE_KPT; ++p) {
                if (acc_row[p] >= 0) {
                    const double* __restrict__ src = P.trial_in + (size_t)acc_row[p] * D;
                    double* __restrict__ xw = P.pos + (size_t)ws[p].widx * D;
                    for (int d = 0; d < D; ++d) xw[d] = src[d];
                    P.lp[ws[p].widx] = vacc[p];
                }
            }
        }
        if (s_nan) { err = 1; finished = true; }
        else if (total == 0) {
            if (h == 1) {                            // the iteration is complete
                __syncthreads();                     // (the accepted rows are in place)
                if (P.chain) {
                    double* __restrict__ cp = P.chain + (size_t)it * P.W * D;
                    double* __restrict__ cl = P.chain_lp + (size_t)it * P.W;
                    for (int i = tid; i < P.W * D; i += nt) cp[i] = P.pos[i];
                    for (int i = tid; i < P.W; i += nt) cl[i] = P.lp[i];
                }
                if (tid == 0) { *cn.nexp = c_exp; *cn.ncon = c_con; slice_tune(cn, P.tolerance, P.patience, P.mu_hist + it); }
                c_exp = 0; c_con = 0;
                ++it;
                h = 0;
                if (it >= P.n) finished = true;
            } else {
                h = 1;
            }
            fresh = true;
            rounds_half = 0;
        } else if (rounds_half > P.round_limit) { err = 2; finished = true; }
    }
    if (finished) {
        fresh = false;
        total = 0;
#pragma unroll
        for (int p = 0; p < SLICE_KPT; ++p) active[p] = false;
    }
    const uint64_t step = P.step_base + (uint64_t)it;
    SL_STAMP(4);
    if (fresh) {
        __syncthreads();                             // positions, lnprob and mu of this launch are final
        const int* __restrict__ perm = P.perm_tab + (size_t)it * P.W;
#pragma unroll
        for (int p = 0; p < SLICE_KPT; ++p) {
            const int k = tid + p * nt;
            active[p] = k < half;
            a[p] = k;
            if (k < half) ws[p] = slice_init_walker(P, st, cn, perm, h, step, k);
        }
        total = half;
    }
    SL_STAMP(5);
    // ---- emit -------------------------------------------------------------------------------------
    int nc = total > 0 ? (P.batch_rows / total) & ~1 : 2;
    nc = nc < 2 ? 2 : (nc > SLICE_MAXC ? SLICE_MAXC : nc);
    const int nrows = total * nc;                    // <= batch_rows
#pragma unroll
    for (int p = 0; p < SLICE_KPT; ++p) {
        ws[p].row = active[p] ? a[p] : -1;
        if (active[p]) {
            o_k[a[p]] = tid + p * nt;
            o_w[a[p]] = ws[p].widx;
            o_c[a[p]] = ws[p].phase == SL_OUT ? -1 : ws[p].nshr;
        }
    }
    lds_barrier();
    SL_STAMP(6);
    for (int r = tid; r < nrows; r += nt) {          // the shrink candidates' uniform draws, one Philox block per batch row
        const int q = r / nc, s = r - q * nc, c0 = o_c[q];
        double u = 0.0;
        if (c0 >= 0) {
            const Philox4 g = draw(P.seed, step, h, o_w[q], 32u + (uint32_t)(c0 + s));
            u = u01(g.v[0], g.v[1]);
        }
        t_lds[r] = u;
    }
    lds_barrier();
    SL_STAMP(7);
#pragma unroll
    for (int p = 0; p < SLICE_KPT; ++p) {            // the bracket bookkeeping, one thread per walker
        if (!active[p]) continue;
        const int k = tid + p * nt;
        double* __restrict__ T = st.T + (size_t)k * SLICE_MAXC;
        double* tl = t_lds + (size_t)a[p] * nc;
        if (ws[p].phase == SL_OUT) {
            const int hs = nc >> 1, sd = ws[p].sides, J = ws[p].J, K = ws[p].K;
            const double L = ws[p].L, R = ws[p].R;
#pragma unroll
            for (int s = 0; s < SLICE_MAXC; ++s) {
                if (s < nc) {
                    const bool left = s < hs;
                    const int i = left ? s : s - hs;
                    const double t = left ? L - (double)i : R + (double)i;
                    const bool valid = left ? (!(sd & 1) && J > i) : (!(sd & 2) && K > i);
                    T[s] = t;
                    tl[s] = valid ? t : __builtin_nan("");
                }
            }
        } else {
            double Lc = ws[p].L, Rc = ws[p].R;
#pragma unroll
            for (int s = 0; s < SLICE_MAXC; ++s) {
                if (s < nc) {
                    const double wd = Lc + tl[s] * (Rc - Lc);
                    T[s] = wd;
                    tl[s] = wd;
                    if (wd < 0.0) Lc = wd; else Rc = wd;
                }
            }
        }
    }
#pragma unroll
    for (int p = 0; p < SLICE_KPT; ++p) {
        const int k = tid + p * nt;
        if (k < half) slice_store(st, k, ws[p]);
    }
    SL_STAMP(8);
    if (fresh) __syncthreads(); else lds_barrier();  // (fresh: X0 / eta of this launch's slice_init_walker)
    SL_STAMP(9);
    // An accepted trial point is row (rank * nc + acc) of the batch just evaluated: copied into the ensemble.  The loads
    // go out in front of the rows' X0 / eta and the stores follow behind the rows, so the two round trips are one.
    constexpr int ACH = 8;
    double abuf[SLICE_KPT][ACH];
    if (late_accept) {
#pragma unroll
        for (int p = 0; p < SLICE_KPT; ++p) {
            if (acc_row[p] >= 0) {
                const double* __restrict__ src = P.trial_in + (size_t)acc_row[p] * D;
#pragma unroll
                for (int j = 0; j < ACH; ++j) abuf[p][j] = src[j < D ? j : D - 1];
            }
        }
    }
    {   // the rows, element by element, four per thread at a time (loads first: one round trip per four)
        const int ne = P.batch_rows * D;
        const double* __restrict__ X0 = st.X0;
        const double* __restrict__ eta = st.eta;
        double* __restrict__ out = P.trial;
        for (int i0 = 0; i0 < ne; i0 += 4 * nt) {
            double x0v[4], ev[4], tv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = i0 + j * nt + tid;
                const int r = i / D, d = i - r * D;
                tv[j] = __builtin_nan(""); x0v[j] = 0.0; ev[j] = 0.0;
                if (i < ne && r < nrows) {
                    tv[j] = t_lds[r];
                    if (tv[j] == tv[j]) {
                        const size_t o = (size_t)o_k[r / nc] * D + d;
                        x0v[j] = X0[o]; ev[j] = eta[o];
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = i0 + j * nt + tid;
                if (i < ne) out[i] = tv[j] == tv[j] ? x0v[j] + tv[j] * ev[j] : __builtin_inf();
            }
        }
    }
    if (late_accept) {
#pragma unroll
        for (int p = 0; p < SLICE_KPT; ++p) {
            if (acc_row[p] >= 0) {
                const double* __restrict__ src = P.trial_in + (size_t)acc_row[p] * D;
                double* __restrict__ xw = P.pos + (size_t)ws[p].widx * D;
#pragma unroll
                for (int j = 0; j < ACH; ++j)
                    if (j < D) xw[j] = abuf[p][j];
                for (int d = ACH; d < D; ++d) xw[d] = src[d];
                P.lp[ws[p].widx] = vacc[p];
            }
        }
    }
    if (finished) {                                  // left-over launches find nothing to evaluate in either batch
        double* other = const_cast<double*>(P.trial_in);
        for (int i = tid; i < P.batch_rows * D; i += nt) other[i] = __builtin_inf();
    }
    SL_STAMP(10);
    if (tid == 0) {
        *cn.n_active = total;
        *cn.ncand = nc;
        if (!start) { *cn.nexp = c_exp; *cn.ncon = c_con; *cn.n_evals = c_ev; }
        if (err) atomicExch(cn.nanflag, err);
        P.prog[0] = it; P.prog[1] = h; P.prog[2] = finished ? 1 : 0; P.prog[3] = err; P.prog[4] = rounds_half; P.prog[5] = rounds_seg;
        if (P.host) {
            __hip_atomic_store(P.host, rounds_seg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (finished) __hip_atomic_store(P.host + 1, err ? 2 : 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    SL_STAMP(11);
}

}  // namespace vp
