// HIP kernels of the Voigt lnprob path for gfx950 (MI355X).
//
// Data flow of one lnprob batch (W walkers; instruments are launched back to back on one stream):
//
//   prep_lines_kernel   one LANE per record   theta row -> per-(walker,line) 512-B record (vp::LC_*),
//                                           box-prior flag per walker + -inf rows (vfit_mcmc.py:291-295),
//                                           optional flag for walkers with lines outside the fast domain
//   tile_kernel         grid (W, tiles) x 64/128/256 thr   workgroup = walker x pixel tile
//     phase A  tau(p) = sum_l tau_l(p) for every line >= 8 Doppler widths away  (voigt_model.py:207-214)
//              lines outer loop, record fields by scalar loads (prefetched one line ahead), RB chunks
//              of 64 pixels per wave pass with one tier per (line, wave pass)
//     phase B  chunks that touch a line core: Taylor-in-a core series (Dawson tables in LDS)
//     then     exp(-tau) (table-driven, LDS)                                  (:217)
//              K-tap LSF from LDS, taps as LDS broadcasts                     (:220-230)
//              chi^2 term vs flux / inv_sigma2, wave reduce -> partial[w][tile] (vfit_mcmc.py:309-311)
//              last-arriving tile of a walker (agent-scope atomics) sums the partials in fixed order
//              and writes lnprob = -0.5 (sum - sum log w)                      (vfit_mcmc.py:348-353)
//              -- or, for batches that fill the wave slots, finalize_kernel does (one lane per walker)
//   Two instances: GENERIC=false (no out-of-line generic Faddeeva, 77 VGPRs, 6 waves/SIMD) and
//   GENERIC=true (handles walkers flagged by prep; launched only when the prior box allows a > 0.1).
//   walker_kernel       grid W x (64 x tiles) thr   ONE launch per batch for small single-instrument batches:
//                       workgroup = walker, wave t = tile t (the same tile work), records formed in the
//                       workgroup, tile sums reduced through LDS -- no prep launch, no ticket, no finalize launch
//
// HBM layout: spectra (wave, 1/wave, flux, inv_sigma2) are 4 dense fp64 arrays per instrument,
// read coalesced (8 B/lane) once per (walker, tile) and shared by all walkers through L2/MALL;
// line records are read wave-uniformly (scalar loads, SGPR operands), never through VGPRs.
// No float atomics anywhere: results are bit-reproducible.
#pragma once
#include "voigt_device.h"
#include "sampler_kernels.h"   // Philox draws + the stretch move's arithmetic (walker_kernel's sampler form)

namespace vp {

// Diagnostic builds (-DVP_DIAG=<sum of ABL_* bits>, scripts/r5_ablate.sh): timing and instruction-count experiments whose RESULTS ARE
// WRONG -- what a phase costs is read off the difference to the product build (profiles/r05_C1_budget.txt).  The product build has
// VP_DIAG = 0: every `if (VP_ABL(..))` below is a compile-time false and leaves no trace in the kernels.
#ifndef VP_DIAG
#define VP_DIAG 0
#endif
enum { ABL_NOFAR = 1,       // no evaluation in the |x| >= 30 tiers (x and the tier decisions stay)
       ABL_NONEAR = 2,      // none in the 8 <= |x| < 30 band: reference-rounded x + 9- / 14-term series
       ABL_NOCORE = 4,      // no phase B (line cores)
       ABL_NOEXP = 8,       // exp(-tau) -> 1 - tau
       ABL_NOLSF = 16,      // the taps are not applied (the chi^2 terms are formed from zeros)
       ABL_NOCHI = 32,      // nothing behind phase B: no LSF, no chi^2 terms
       ABL_ENTRY = 64,      // walker_kernel ends behind its first barrier: what the entry costs alone
       ABL_NOSMALLEXP = 128,  // no short Taylor form of exp where a chunk is shallow
       ABL_NOPRIO = 256,      // no raised issue priority for waves with line cores
       ABL_EMPTY = 512,       // walker_kernel returns at once: what a launch of this shape costs
       ABL_NOFFPOLY = 1024,   // the blocks' far-field polynomials are not evaluated (FF instance)
       ABL_NOMP = 2048 };     // no multipole evaluation of far clusters (their members are skipped all the same)
#define VP_ABL(bit) ((VP_DIAG & (bit)) != 0)

struct LinesDev {          // static per-instrument line tables (CompiledModelData, voigt_model.py:265-280)
    int L;
    const double* lambda0;
    const double* freq0;      // C_FREQ / lambda0 (IEEE quotient, formed on the host)
    const double* gamma;
    const double* f;
    const double* zfac;
    const int* N_idx;
    const int* b_idx;
    const int* v_idx;
    // multipole clusters (components of one transition): first line and member count per cluster
    int NCm;
    const int* cl_first;
    const int* cl_count;
    const int* cl_mp;      // per line: cluster index if the line is the FIRST of a multipole cluster, else -1
    const int* cl_end;     // per line: one past the last line of its cluster
    // the members of all clusters in a row (prep_lines_kernel forms a cluster's record with one lane per member):
    int M;                 // members of all clusters together
    const int4* mem_info;  // (M) per member: cluster, line, first member of the cluster in the row, members of the cluster
                           // (one 16-byte load: what a member lane needs to find its line tables)
};

struct FinalizeArgs {      // fused final reduction (last-arriving workgroup of a walker)
    unsigned int* ticket;        // (W) arrival counters, zero between launches
    const int* tile_off;         // (n_inst + 1) offsets into a walker's partial row; NULL: one instrument, slots [0, total_tiles)
    const double* sum_logw;      // (n_inst) sum of log inv_sigma2
    double* lnprob;              // (W) output
    int n_inst;
    int total_tiles;             // arrivals per walker over all instruments
    int mode;                    // 0: finalize_kernel sums; 1: the last-arriving tile of the walker does (ticket)
};

struct InstDev {
    int P;         // pixels
    int L;         // lines
    int K;         // taps (>= 1; K = 1 with tap 1.0 when there is no LSF)
    int halo_lo;   // K-1-c : evaluated pixels before the first output pixel of a tile
    int span;      // evaluated pixels per full tile (multiple of 64)
    int TP;        // output pixels per tile = span - (K-1)
    int ntiles;
    int method;    // VP_VOIGT_*
    int line_sel;  // -1: all lines; >= 0: only this line (per-line component flux); -2: line = blockIdx.z (OUT = 2)
    int NCm;       // number of multipole clusters (records follow the line records of a walker)
    const double* wave;
    const double* ginv;    // RN(1/wave)
    const double* flux;
    const double* w;       // inv_sigma2
    const double* kflip;   // taps flipped (and normalised for the astropy branch): out[p] = sum_j kflip[j] f[p-halo_lo+j]
    // far-field expansions (FF instance of tile_kernel, see farfield_kernel): per block of 64*RB evaluated pixels
    // [g_c, hw, 1/hw, g_c/hw] of 1/wave over the block (instrument tables, ntiles x ff_nblk x 4), and the per-launch
    // workspace (W, ntiles x ff_nblk, FF_STRIDE) of expansion coefficients + masks of the lines they cover
    const double* ff_tab;
    double* ff;
    int ff_nblk;           // blocks per tile = ceil(span / (64 RB))
    int ff_members;        // 1: members of clusters too near for their multipole may enter a block's expansion line by line (spectra
                           // whose blocks are a few Doppler widths wide: C4; pointless, and a little slower, on coarse grids)
    const double* rbot;    // NULL, or (P): instruments on the astropy branch whose wavelength grid holds NaN samples (tile_work<..., NANFIX>):
                           // astropy's convolve leaves NaN model pixels out and divides every output by the kernel weight it did use
                           // (nan_treatment='interpolate', voigt_model.py:227,230) -- 1 / that weight per output pixel, made on the host
                           // from the (static) NaN pattern; NaN where a whole window is NaN
    int* core_hint;        // (16) walker_kernel: tile t met line cores in an earlier launch -> its wave stages the Dawson
                           // table while it waits for the records instead of between phase A and phase B (a hint only:
                           // results never depend on it); behind them (TILE_ORDER_AT ...) tile_kernel's order of the
                           // geometry's tiles: entry i = 1 + the tile handed to blockIdx.y = i, or 0 = tile i
};
constexpr int TILE_ORDER_AT = 16;

// ---------------------------------------------------------------------------------------------
// compile-time table of the wing-series polynomials  C_m(a^2) = sum_i WC[m][i] a^(2i)
//   WC[m][i] = binom(2m+1, 2i+1) (-1)^i (2(m-i)-1)!! / 2^(m-i)
// ---------------------------------------------------------------------------------------------
// Multipole record of a cluster (same 64-double stride as a line record):
//   [0] A_c  [1] B_c  (y = A_c/wave - B_c)
//   [2..6] Y_0..Y_4: tier k of the expansion may be used where every |y| >= Y_k (inf: never)
//   [7..32] Q_2..Q_27 :  tau_cluster(y) = sum_j Q_j y^-j
// Tiers (member |x| floor X_k, ratio floor |y|/max|delta| R_k, terms J_k): the farther the pixels, the
// fewer terms of the same series are needed (worst relative truncation error over random clusters of
// 2..8 members against scipy's wofz: 2e-14, 4e-13, 8e-14, 9e-14, 1e-14 -- scripts/multipole_check.py).
// The nearest tier (ratio 1/4, 26 terms) is what brings the expansion right up to the 6-term radius of its
// members (|x| >= 30) for clusters as wide as 7.5 Doppler widths: 26 FMAs per chunk instead of n_members x ~25.
constexpr int MP_A = 0, MP_B = 1, MP_Y0 = 2, MP_NTIER = 5, MP_Q0 = 7, MP_NQ = 26, MP_JMIN = 2;
constexpr int MP_MWING = 6;      // asymptotic terms kept per member (valid for |x| >= 30)
constexpr double MP_X[MP_NTIER] = {30.0, 30.0, 100.0, 500.0, 3000.0};
constexpr double MP_R[MP_NTIER] = {4.0, 10.0, 30.0, 100.0, 1000.0};
constexpr int MP_J0 = 26, MP_J1 = 15, MP_J2 = 10, MP_J3 = 7, MP_J4 = 5;

struct WingTable { double c[NWING][NWING]; };
constexpr WingTable make_wing_table() {
    WingTable t{};
    unsigned long long binom[2 * NWING + 2][2 * NWING + 2] = {};
    for (int n = 0; n < 2 * NWING + 2; ++n) {
        binom[n][0] = 1;
        for (int k = 1; k <= n; ++k) binom[n][k] = binom[n - 1][k - 1] + (k <= n - 1 ? binom[n - 1][k] : 0);
    }
    double dfo2[NWING] = {};   // (2n-1)!!/2^n
    dfo2[0] = 1.0;
    for (int n = 1; n < NWING; ++n) dfo2[n] = dfo2[n - 1] * (double)(2 * n - 1) * 0.5;
    for (int m = 0; m < NWING; ++m)
        for (int i = 0; i <= m; ++i)
            t.c[m][i] = (double)binom[2 * m + 1][2 * i + 1] * ((i & 1) ? -1.0 : 1.0) * dfo2[m - i];
    return t;
}
__constant__ const WingTable g_wing = make_wing_table();

// A kernel argument read from the kernarg segment AT THE POINT OF USE (a scalar load; the empty asm keeps the compiler from
// moving it up).  By-value kernel parameters are loaded at entry and then live -- in these kernels: are spilled to VGPR lanes and
// reloaded, a VALU instruction each time -- until their last use; for arguments that are needed once per pass or only when the
// tile is done, loading them again is cheaper than keeping them.  tile_kernel1 (InstDev at offset 0 of its argument struct).
template <class T>
__device__ __forceinline__ T late_kernarg(size_t byte_offset) {
    static_assert(sizeof(T) % 4 == 0 && alignof(T) >= 4, "read as 32-bit words");
    T v;
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned int __attribute__((address_space(4)))* p =
        (const unsigned int __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    unsigned int* d = reinterpret_cast<unsigned int*>(&v);
#pragma unroll
    for (size_t i = 0; i < sizeof(T) / 4; ++i) d[i] = p[byte_offset / 4 + i];
#else
    (void)byte_offset;
    __builtin_memset(&v, 0, sizeof(T));
#endif
    return v;
}
#define VP_LATE_FIELD(LATE, I, field) ((LATE) ? late_kernarg<decltype((I).field)>(offsetof(InstDev, field)) : (I).field)
// ... as a pointer in the GLOBAL address space: a pointer the compiler did not see come in as a kernel argument is a generic one
// to it, and loads through it would be flat_load (counted in lgkmcnt as well as vmcnt: they stall the LDS and scalar-load waits)
typedef const double __attribute__((address_space(1)))* gptr_t;
#define VP_LATE_GPTR(LATE, I, field) ((gptr_t)reinterpret_cast<unsigned long long>(VP_LATE_FIELD(LATE, I, field)))

// Sum over the 64 lanes of a wave, the same value in every lane.  Data-parallel-primitive moves inside the 16-lane rows
// (lane ^ 1, lane ^ 2, 7 - lane, 15 - lane), then row 0 into row 1 and row 2 into row 3 (row_bcast:15), rows 0-1 into rows 2-3
// (row_bcast:31) and lane 63 read back: 18 VALU instructions and two v_readlane, no LDS traffic.  The shuffle form
// (__shfl_xor = ds_bpermute: six dependent LDS round trips with a compare / select / shift each, ~42 instructions) sat at the very
// end of every tile wave's life -- in walker_kernel on the workgroup's critical path.  The tree is fixed, so results stay
// reproducible run to run; it is not the butterfly's tree, so the last bit of a tile's partial sum can differ from round 3's.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_f64<0xB1, 0xf>(v);          // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E, 0xf>(v);          // quad_perm [2,3,0,1]
    v += dpp_f64<0x141, 0xf>(v);         // row_half_mirror
    v += dpp_f64<0x140, 0xf>(v);         // row_mirror: every lane holds its row's sum
    v += dpp_f64<0x142, 0xa>(v);         // row_bcast:15 into rows 1 and 3 (the other rows add the 0 of the masked move)
    v += dpp_f64<0x143, 0xc>(v);         // row_bcast:31 into rows 2 and 3: lane 63 holds the wave's sum
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

// Fill one record from (T, a).  ONE LANE PER RECORD: a wave prepares up to 64 records at once, so
// the scalar prologue of a record is executed once instead of once per lane.
__device__ __forceinline__ void fill_record(double* __restrict__ rec, double T, double a) {
    int mode = 0;
    if (!(a >= 0.0) || !(a < 7.0)) mode = 2;         // a < 0, a >= 7
    else if (a > 0.1) mode = 1;
    if (!(fabs(a) <= 1.79e308) || !(fabs(T) <= 1.79e308)) mode = 3;   // NaN/inf constants: the line poisons tau
                                                                       // with NaN, as the reference's arithmetic does
    const double a2 = a * a;
    // Gaussian-sum table and its plain sum: only lines outside the fast domain need them
    double S1 = 0.0;
    if (mode == 1) {
        for (int n = 0; n < NCORE; ++n) {
            const double hn = ALG916_H * (double)(n + 1);
            const double hn2 = hn * hn;
            const double tb = exp(-hn2) / (hn2 + a2);
            S1 += tb;
            rec[LC_TBL0 + n] = (0.5 * ALG916_C * a) * tb;
        }
    }
    // wing coefficients
    const double pref = T * (a * INV_SQRT_PI);
#pragma unroll
    for (int m = 0; m < NWING; ++m) {
        double cm = 0.0;
#pragma unroll
        for (int i = m; i >= 0; --i) cm = __builtin_fma(cm, a2, g_wing.c[m][i]);
        rec[LC_K0 + m] = pref * cm;
    }
    rec[LC_T] = T;
    rec[LC_Y] = a;
    rec[LC_ACOS] = (mode == 1) ? erfcx(a) - (ALG916_C * a) * S1 : 0.0;
    reinterpret_cast<int*>(rec + LC_MODE)[0] = mode;
    reinterpret_cast<int*>(rec + LC_MODE)[1] = core_terms(a);
    // exp(a^2): five Taylor terms in the fast domain (a <= 0.1, remainder 1e-22)
    rec[LC_EA2] = (mode == 0) ? 1.0 + a2 * (1.0 + a2 * (0.5 + a2 * (1.0 / 6 + a2 * (1.0 / 24)))) : exp(a2);
}

// Per-line scalars of one (walker, line): follows _evaluate_compiled_model's scalar prologue
// (voigt_model.py:192-200) and _vectorized_voigt_tau's per-line constants (:142-149) operation by
// operation (the file is compiled with -ffp-contract=off, so nothing here is fused).
struct LineScalars { double d, freq0, ibf, a, Tl, cfd, Ax, Bx; };
__device__ __forceinline__ LineScalars line_scalars_nbv(double thN, double b, double v, const LinesDev& T, int l) {
    LineScalars s;
    const double lam0 = T.lambda0[l], gam = T.gamma[l], fo = T.f[l], zf = T.zfac[l];
    const double N = exp10(thN);                       // :192  (10**theta; <= 1 ulp, only scales tau)
    // quantities that enter x are formed with the reference's operations and roundings ...
    const double z_total = zf * (1.0 + v / C_KMS) - 1.0;   // :200
    s.d = 1.0 + z_total;                               // :204
    const double b_f = b / lam0 * 1e13;                // :142
    s.freq0 = T.freq0[l];                              // :143  C_FREQ / lambda0, IEEE quotient from the host
    // ... the others only need ~1e-16 relative accuracy (reciprocal + Newton instead of IEEE division)
    s.ibf = fast_rcp(b_f);
    const double constant = ATOMIC_CONSTANT * fast_rcp(s.freq0 * b);   // :146
    s.a = gam * (s.ibf * 0.079577471545947667884);     // :149  gamma / (4 pi b_f)
    s.Tl = (N * fo) * constant;                        // :158
    s.cfd = C_FREQ * s.d;
    s.Ax = s.cfd * s.ibf;
    s.Bx = s.freq0 * s.ibf;
    return s;
}
__device__ __forceinline__ LineScalars line_scalars(const double* __restrict__ th, const LinesDev& T, int l) {
    return line_scalars_nbv(th[T.N_idx[l]], th[T.b_idx[l]], th[T.v_idx[l]], T, l);     // :192-194
}

// Multipole expansion of one cluster (the components of one transition) about its centre, for the
// pixels that are far from all of its members:  with y = A_c/wave - B_c and x_l = alpha_l (y + delta_l),
//   sum_l sum_{m<4} K_{l,m} x_l^(-2m-2)  =  sum_{j=2..14} Q_j y^-j,
//   Q_j = sum_l sum_{n=2m+2<=j} K_{l,m} alpha_l^-n (-1)^(j-n) C(j-1, n-1) delta_l^(j-n).
// A_c is the smallest A_l of the cluster (alpha_l >= 1), B_c centres the delta_l.  The record is used
// only where every |y| >= Y_0 = max(30 + max|delta|, 10 max|delta|): there each member is in its
// 6-term asymptotic regime (|x_l| >= 30) and the expansion ratio is <= 0.1; farther out shorter
// truncations of the same series serve (tiers above).  One lane per cluster record; two passes over
// the members.
__device__ __forceinline__ void prep_cluster(const double* __restrict__ th, const LinesDev& T, int k,
                                             double* __restrict__ rec) {
    const int first = T.cl_first[k], n = T.cl_count[k];
    bool allok = true;
    double Ac = 1.79e308, g0sum = 0.0;
    for (int j = 0; j < n; ++j) {
        const LineScalars s = line_scalars(th, T, first + j);
        allok = allok && (s.a >= 0.0) && (s.a <= 0.1) && (fabs(s.Tl) <= 1.79e308) && (fabs(s.Ax) <= 1.79e308)
                && (fabs(s.Bx) <= 1.79e308) && (s.Ax > 0.0);
        Ac = fmin(Ac, s.Ax);
        g0sum += s.Bx * fast_rcp(s.Ax);                         // line centre in 1/wave units
    }
    const double g0c = g0sum / (double)n;
    const double Bc = Ac * g0c;
    double dmax = 0.0;
    double Q[MP_NQ];
#pragma unroll
    for (int jq = 0; jq < MP_NQ; ++jq) Q[jq] = 0.0;
    for (int j = 0; j < n; ++j) {
        const LineScalars s = line_scalars(th, T, first + j);
        const double g0 = s.Bx * fast_rcp(s.Ax);
        const double alpha = s.Ax * fast_rcp(Ac);                // >= 1
        const double delta = Ac * (g0c - g0);                    // x_l = alpha (y + delta)
        dmax = fmax(dmax, fabs(delta));
        // member's wing coefficients K_m, m < MP_MWING, divided by alpha^(2m+2)
        double Km[MP_MWING];
        const double a2 = s.a * s.a, pref = s.Tl * (s.a * INV_SQRT_PI);
        const double ia2 = fast_rcp(alpha * alpha);
        double ipow = ia2;
#pragma unroll
        for (int m = 0; m < MP_MWING; ++m) {
            double cm = 0.0;
#pragma unroll
            for (int i = m; i >= 0; --i) cm = __builtin_fma(cm, a2, g_wing.c[m][i]);
            Km[m] = pref * cm * ipow;
            ipow *= ia2;
        }
        double dpow[MP_NQ + 1];                                  // (-delta)^k, k = 0..MP_NQ
        dpow[0] = 1.0;
#pragma unroll
        for (int kk = 1; kk <= MP_NQ; ++kk) dpow[kk] = dpow[kk - 1] * (-delta);
#pragma unroll
        for (int jq = 0; jq < MP_NQ; ++jq) {
            const int jj = jq + MP_JMIN;
#pragma unroll
            for (int m = 0; m < MP_MWING; ++m) {
                const int nn = 2 * m + 2;
                if (nn <= jj) {
                    double binom = 1.0;                          // C(jj-1, nn-1), folded at compile time
                    for (int t = 1; t <= nn - 1; ++t) binom = binom * (double)(jj - 1 - (nn - 1) + t) / (double)t;
                    Q[jq] = __builtin_fma(Km[m] * binom, dpow[jj - nn], Q[jq]);
                }
            }
        }
    }
#pragma unroll
    for (int jq = 0; jq < MP_NQ; ++jq) rec[MP_Q0 + jq] = Q[jq];
    rec[MP_A] = Ac;
    rec[MP_B] = Bc;
#pragma unroll
    for (int k = 0; k < MP_NTIER; ++k)                          // inf: never use the expansion
        rec[MP_Y0 + k] = allok ? fmax(MP_X[k] + dmax, MP_R[k] * dmax) : __builtin_inf();
}

// The same record by the lanes of one wave, one lane per MEMBER (lanes base .. base + n - 1, all in this wave): the chain of a
// lane is one member's share (~500 instructions) instead of all n of them in a row -- the record-preparation launch is as long as
// its cluster lanes' chains (C2: 3 members, C4: 8).  Centre, radius and validity are formed from the members' numbers in member
// order exactly as prep_cluster does; the Q_j are the members' own sums added in member order (prep_cluster keeps ONE
// accumulator through all members: the last bits of Q_j may differ between the two, both are the same series).
//   sh: LDS of the wave, 64 x (MPL_STRIDE doubles); j: this lane's member number, n: members, base: lane of member 0.
constexpr int MPL_STRIDE = 27;        // Q_2..Q_27 of a member (26) + 1: lane stride of 54 banks keeps 8-byte accesses conflict-free
__device__ __forceinline__ void prep_cluster_lanes(const double* __restrict__ th, const LinesDev& T, int line, int j, int n, int base,
                                                   int lane, bool valid, double* __restrict__ sh, double* __restrict__ rec) {
    const LineScalars s = line_scalars(th, T, line);
    const bool ok = (s.a >= 0.0) && (s.a <= 0.1) && (fabs(s.Tl) <= 1.79e308) && (fabs(s.Ax) <= 1.79e308)
                    && (fabs(s.Bx) <= 1.79e308) && (s.Ax > 0.0);
    const double g0 = s.Bx * fast_rcp(s.Ax);                     // line centre in 1/wave units
    double* __restrict__ mine = sh + (size_t)lane * MPL_STRIDE;
    mine[0] = s.Ax; mine[1] = g0; mine[2] = ok ? 1.0 : 0.0;
    __syncthreads();
    bool allok = true;
    double Ac = 1.79e308, g0sum = 0.0;
    for (int jj = 0; jj < n; ++jj) {
        const double* __restrict__ o = sh + (size_t)(base + jj) * MPL_STRIDE;
        allok = allok && (o[2] != 0.0);
        Ac = fmin(Ac, o[0]);
        g0sum += o[1];
    }
    const double g0c = g0sum / (double)n;
    const double Bc = Ac * g0c;
    const double alpha = s.Ax * fast_rcp(Ac);                    // >= 1
    const double delta = Ac * (g0c - g0);                        // x_l = alpha (y + delta)
    __syncthreads();                                             // (everyone has read the first exchange)
    // member's wing coefficients K_m, m < MP_MWING, divided by alpha^(2m+2)
    double Km[MP_MWING];
    const double a2 = s.a * s.a, pref = s.Tl * (s.a * INV_SQRT_PI);
    const double ia2 = fast_rcp(alpha * alpha);
    double ipow = ia2;
#pragma unroll
    for (int m = 0; m < MP_MWING; ++m) {
        double cm = 0.0;
#pragma unroll
        for (int i = m; i >= 0; --i) cm = __builtin_fma(cm, a2, g_wing.c[m][i]);
        Km[m] = pref * cm * ipow;
        ipow *= ia2;
    }
    double dpow[MP_NQ + 1];                                      // (-delta)^k, k = 0..MP_NQ
    dpow[0] = 1.0;
#pragma unroll
    for (int kk = 1; kk <= MP_NQ; ++kk) dpow[kk] = dpow[kk - 1] * (-delta);
#pragma unroll
    for (int jq = 0; jq < MP_NQ; ++jq) {
        const int jj = jq + MP_JMIN;
        double q = 0.0;
#pragma unroll
        for (int m = 0; m < MP_MWING; ++m) {
            const int nn = 2 * m + 2;
            if (nn <= jj) {
                double binom = 1.0;                              // C(jj-1, nn-1), folded at compile time
                for (int t = 1; t <= nn - 1; ++t) binom = binom * (double)(jj - 1 - (nn - 1) + t) / (double)t;
                q = __builtin_fma(Km[m] * binom, dpow[jj - nn], q);
            }
        }
        mine[jq] = q;
    }
    mine[MP_NQ] = fabs(delta);
    __syncthreads();
    if (!valid) return;
    // the members' sums, added in member order: lane j takes Q_j, Q_(j+n), ...
    for (int jq = j; jq < MP_NQ; jq += n) {
        double q = 0.0;
        for (int jj = 0; jj < n; ++jj) q += sh[(size_t)(base + jj) * MPL_STRIDE + jq];
        rec[MP_Q0 + jq] = q;
    }
    if (j == 0) {
        double dmax = 0.0;
        for (int jj = 0; jj < n; ++jj) dmax = fmax(dmax, sh[(size_t)(base + jj) * MPL_STRIDE + MP_NQ]);
        rec[MP_A] = Ac;
        rec[MP_B] = Bc;
#pragma unroll
        for (int kk = 0; kk < MP_NTIER; ++kk)                     // inf: never use the expansion
            rec[MP_Y0 + kk] = allok ? fmax(MP_X[kk] + dmax, MP_R[kk] * dmax) : __builtin_inf();
    }
}

// One line's whole record by one lane (every tier's constants: what prep_lines_kernel writes, and what a walker that leaves the
// fast domain needs of the walker kernel's flux form)
__device__ __forceinline__ LineScalars prep_line_record(const double* __restrict__ th, const LinesDev& T, int l, double* __restrict__ rec) {
    const LineScalars s = line_scalars(th, T, l);
    const bool xok = (fabs(s.Ax) <= 1.79e308) && (fabs(s.Bx) <= 1.79e308);
    fill_record(rec, xok ? s.Tl : __builtin_nan(""), s.a);
    rec[LC_A] = s.Ax;
    rec[LC_B] = s.Bx;
    rec[LC_D] = s.d;
    rec[LC_RD] = 1.0 / s.d;               // must be the correctly rounded reciprocal (faithful_x)
    rec[LC_CFD] = s.cfd;
    rec[LC_FREQ0] = s.freq0;
    rec[LC_IBF] = s.ibf;
    reinterpret_cast<int*>(rec + LC_CL)[0] = T.NCm > 0 ? T.cl_mp[l] : -1;
    reinterpret_cast<int*>(rec + LC_CL)[1] = T.NCm > 0 ? T.cl_end[l] : l + 1;
    return s;
}

// Record preparation, one LANE per record.  Block roles by blockIdx.x:
//   [0, nb_line)                 line records     index = blockIdx.x * rpw + lane  over (walker, line)
//   [nb_line, nb_line + nb_cl)   cluster records  one lane per MEMBER, cl_wpw walkers per wave (T.M <= 64 members per walker:
//                                                 prep_cluster_lanes), or -- cl_wpw = 0 -- one lane per record over (walker, cluster)
//   the rest                     box-prior flags  one wave per walker
// `rpw` (records per wave, <= 64) spreads a small batch over more CUs; lanes >= rpw idle.
struct PrepGrid { int rpw, nb_line, nb_cl, nb_flag, cl_wpw; };
__global__ __launch_bounds__(64) void prep_lines_kernel(const double* __restrict__ theta, int W, int D,
                                                        LinesDev T, const double* __restrict__ lb,
                                                        const double* __restrict__ ub,
                                                        double* __restrict__ lc, int* __restrict__ flags,
                                                        int do_flags, double* __restrict__ lnprob_out,
                                                        int* __restrict__ genflag, PrepGrid G, Replicas R,
                                                        int* __restrict__ genflag_clear = nullptr,
                                                        int* __restrict__ gen_any = nullptr, int* __restrict__ gen_any_clear = nullptr) {
    // direct-write gather (vp_gather_*): the batch before this one has arrived here from every rank before this one starts
    if (R.n > 1) replicas_handshake(R);
    const int nrec = T.L + T.NCm;                     // records per walker: lines, then clusters
    const int lane = threadIdx.x;
    int blk = blockIdx.x;
    if (blk < G.nb_line) {
        const int ridx = blk * G.rpw + lane;
        if (lane >= G.rpw || ridx >= W * T.L) return;
        const int w = ridx / T.L, l = ridx - w * T.L;
        double* rec = lc + ((size_t)w * nrec + l) * LC_STRIDE;
        const LineScalars s = prep_line_record(theta + (size_t)w * D, T, l, rec);
        // lines outside the fast domain (a > 0.1, a < 0): their walker goes to the generic kernel
        if (genflag && (!(s.a >= 0.0) || s.a > 0.1) && (fabs(s.a) <= 1.79e308)) {
            genflag[w] = 1;
            if (gen_any) *gen_any = 1;       // (mapped host memory: the host sizes the NEXT batch's generic launch by it)
        }
        // (the flags of the NEXT launch of this kind live in the other buffer: cleared here, so that no memset stands in front of it)
        if (genflag_clear && l == 0) genflag_clear[w] = 0;
        if (gen_any_clear && ridx == 0) *gen_any_clear = 0;
        return;
    }
    blk -= G.nb_line;
    if (blk < G.nb_cl && G.cl_wpw > 0) {              // ---- multipole records, a lane per member ----
        __shared__ double sh[64 * MPL_STRIDE];
        const int wl = lane / T.M, mi = lane - wl * T.M, w = blk * G.cl_wpw + wl;
        const bool valid = wl < G.cl_wpw && w < W;
        const int wv = min(w, W - 1), wlv = min(wl, G.cl_wpw - 1);                 // (idle lanes repeat a valid one's work, store nothing)
        const int4 mem = T.mem_info[mi];
        const int k = mem.x, m0 = mem.z, n = mem.w;
        prep_cluster_lanes(theta + (size_t)wv * D, T, mem.y, mi - m0, n, wlv * T.M + m0, lane, valid, sh,
                           lc + ((size_t)wv * nrec + T.L + k) * LC_STRIDE);
        return;
    }
    if (blk < G.nb_cl) {                              // ---- multipole records, a lane per record ----
        const int ridx = blk * G.rpw + lane;
        if (lane >= G.rpw || ridx >= W * T.NCm) return;
        const int w = ridx / T.NCm, k = ridx - w * T.NCm;
        prep_cluster(theta + (size_t)w * D, T, k, lc + ((size_t)w * nrec + T.L + k) * LC_STRIDE);
        return;
    }
    blk -= G.nb_cl;
    // ---- box prior (vfit_mcmc.py:291-295, 350-351): one WAVE per walker, the parameters across its lanes (a lane per
    //      walker walked its D parameters one load after the other -- 13 us of the launch at D = 96)
    const int w = blk;
    if (!do_flags || w >= W) return;
    const double* th = theta + (size_t)w * D;
    bool oob = false;
    for (int d = lane; d < D; d += 64) oob = oob || (th[d] < lb[d]) || (th[d] > ub[d]);
    const bool any = __ballot(oob) != 0ull;
    if (lane == 0) {
        flags[w] = any ? 1 : 0;
        if (any && lnprob_out) lnprob_out[w] = -__builtin_inf();
    }
}

// Test hook: records for a list of damping parameters with T = 1.
__global__ __launch_bounds__(64) void prep_h_kernel(const double* __restrict__ a, int na, double* __restrict__ lc) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= na) return;
    double* rec = lc + (size_t)i * LC_STRIDE;
    fill_record(rec, 1.0, a[i]);
    rec[LC_A] = 0; rec[LC_B] = 0; rec[LC_D] = 1; rec[LC_RD] = 1; rec[LC_CFD] = 0; rec[LC_FREQ0] = 0; rec[LC_IBF] = 1;
    reinterpret_cast<int*>(rec + LC_CL)[0] = -1;
    reinterpret_cast<int*>(rec + LC_CL)[1] = 0;
}


struct PixelX {    // x of one pixel for one line, computed from the spectrum grid
    double wv, g;
    __device__ __forceinline__ double cheap(rec_t rec) const { return __builtin_fma(rec[LC_A], g, -rec[LC_B]); }
    __device__ __forceinline__ double faithful(rec_t rec) const { return faithful_x(wv, g, rec); }
};
struct DirectX {   // x given (test hook)
    double x;
    __device__ __forceinline__ double cheap(rec_t) const { return x; }
    __device__ __forceinline__ double faithful(rec_t) const { return x; }
};

#define VP_NONE_BELOW(xa, thr) (__ballot((xa) < (thr)) == 0ull)

// Optical depth of one line at one pixel; tier chosen per wavefront from the cheap x.
// (Single-chunk form: used by the H test hook and by the cold path of the tile kernel.)
template <class XP>
__device__ __forceinline__ double line_tau_wofz(const XP& xp, rec_t rec) {
    const int mode = rec_int(rec, LC_MODE, 0);
    const double xc = xp.cheap(rec);
    const double xa = fabs(xc);
    rec_t K = rec + LC_K0;
    if (mode == 0) {
        if (VP_NONE_BELOW(xa, 30.0)) {             // far wings: the 1-FMA x is accurate enough
            if (VP_NONE_BELOW(xa, 3000.0)) return wing_tau<2>(xc, K);
            if (VP_NONE_BELOW(xa, 500.0)) return wing_tau<3>(xc, K);
            if (VP_NONE_BELOW(xa, 100.0)) return wing_tau<4>(xc, K);
            return wing_tau<6>(xc, K);
        }
        const double xf = xp.faithful(rec);        // reproduce the reference's rounding of x
        if (VP_NONE_BELOW(xa, 14.0)) return wing_tau<9>(xf, K);
        if (VP_NONE_BELOW(xa, X_CORE)) return wing_tau<NWING>(xf, K);
        // line core: evaluated for every lane of the chunk (no divergence); lanes beyond the core
        // radius of a mixed chunk take the 14-term wing value instead
        const int nodd = rec_int(rec, LC_MODE, 1);
        double r = rec[LC_T] * core_taylor_H(xf, rec[LC_Y], rec[LC_EA2], nodd);
        if (__ballot(xa >= X_CORE) != 0ull) {
            const double rw = wing_tau<NWING>(xf, K);
            r = (xa >= X_CORE) ? rw : r;
        }
        return r;
    }
    const double xf = xp.faithful(rec);
    return rec[LC_T] * generic_H(xf, rec, mode);
}

// Line records are read with SCALAR loads (wave-uniform addresses): the fields land in SGPRs and
// feed the fp64 VALU ops as scalar operands at no VALU cost.  (A previous version held the record
// across the lanes of a VGPR and broadcast fields with v_readlane; measured at ~4.3 cycles each, the
// readlanes were ~17 % of all VALU instructions, and a cross-lane read inside a divergent branch is
// unsafe because the compiler may reuse the VGPR in the inactive lanes.)

// Cold path of the tile kernel: one 64-pixel chunk of one line that touches the line core (or a
// line outside the fast domain).  Out of line on purpose: it keeps the hot loop's registers and
// instruction footprint small.
__device__ __attribute__((noinline)) double cold_line_tau(double wv, double g, rec_t rec) {
    // the pointer is wave-uniform but arrives in VGPRs: scalarise it so the record fields are
    // fetched with scalar loads (valid under any EXEC mask)
    const unsigned long long pv = (unsigned long long)rec;
    const unsigned int plo = __builtin_amdgcn_readfirstlane((unsigned int)pv);
    const unsigned int phi = __builtin_amdgcn_readfirstlane((unsigned int)(pv >> 32));
    rec_t urec = (rec_t)(((unsigned long long)phi << 32) | plo);
    PixelX xp{wv, g};
    return line_tau_wofz(xp, urec);
}

template <class XP>
__device__ __forceinline__ double line_tau_fast(const XP& xp, rec_t rec, const double* __restrict__ et) {
    const double xf = xp.faithful(rec);
    return rec[LC_T] * tepper_garcia_H(xf, rec[LC_Y], et);
}

// ---------------------------------------------------------------------------------------------
// far-field expansions
// ---------------------------------------------------------------------------------------------
// Far from a line (every |x| of the block >= 30) its optical depth is the asymptotic series
//   tau_l(x) = sum_{m<6} K_m x^-(2m+2),   x = A g - B  linear in g = 1/wave,
// a function so smooth over a block of 192 pixels that a short polynomial in t = (g - g_c)/hw, |t| <= 1, carries it:
// with x = x_c (1 + r t), r = A hw / x_c,
//   x^-k = x_c^-k sum_j (-1)^j C(k+j-1, j) r^j t^j .
// farfield_kernel adds the first FF_NC coefficients of every line that is far enough from the block (|r| <= 1/8, and
// the neglected tail K_0 x_c^-2 13 |r|^12 / (1-|r|)^2 below 1e-16 in optical depth) into ONE polynomial per (walker,
// block) and records which lines it covers; the tile kernel's FF instance then pays FF_NC FMAs per pixel for ALL
// those lines together and walks only the others (C2: 19 lines, of which 15-19 are far from most blocks).  Members of
// a multipole cluster go in together or not at all, so that the cluster's own expansion never counts a line twice.
constexpr int FF_NC = 12, FF_STRIDE = 16, FF_MASK0 = 12;      // per (walker, block): c_0..c_11 | 2 mask words | pad
constexpr int FF_M = 9;            // asymptotic terms a line's expansion can carry (rows of the table below).  farfield_kernel<M>: 6 terms
                                   // serve blocks with every |x| >= 30; 9 serve |x| >= 14 (the tile kernel's own tiers) and are used where
                                   // the blocks are narrow enough for that band to matter (InstDev::ff_members: C4 -17 % per pass, C2 / C3 +1 %
                                   // if they carried them)
#ifndef VP_FF_MIN_LANES
#define VP_FF_MIN_LANES 4
#endif
struct FFTable { double b[FF_M][FF_NC]; };
constexpr FFTable make_ff_table() {
    FFTable t{};
    for (int m = 0; m < FF_M; ++m) {
        const int k = 2 * m + 2;
        double c = 1.0;                                          // C(k+j-1, j), j = 0
        for (int j = 0; j < FF_NC; ++j) {
            t.b[m][j] = (j & 1) ? -c : c;
            c = c * (double)(k + j) / (double)(j + 1);           // -> C(k+j, j+1)
        }
    }
    return t;
}
__device__ const FFTable g_ff = make_ff_table();

// (-1)^i C(k+i-1, i) for the powers k = 2 .. FF_KMAX+1 of a cluster's multipole series
constexpr int FF_KMAX = MP_J1;              // cluster expansions: the multipole tiers of 5 ... 15 terms (the nearest, 26-term tier
                                            // almost never meets the 1/8 half-width rule: measured no gain)
struct FFTable2 { double b[FF_KMAX][FF_NC]; };
constexpr FFTable2 make_ff_table2() {
    FFTable2 t{};
    for (int jq = 0; jq < FF_KMAX; ++jq) {
        const int k = jq + 2;
        double c = 1.0;
        for (int i = 0; i < FF_NC; ++i) {
            t.b[jq][i] = (i & 1) ? -c : c;
            c = c * (double)(k + i) / (double)(i + 1);
        }
    }
    return t;
}
__device__ const FFTable2 g_ff2 = make_ff_table2();

// sum_{jq < J} p_jq (1 + r t)^-(jq+2) -> coefficients of t^i, added to c (J wave-uniform; a lane with fewer terms holds zeros)
template <int J>
__device__ __forceinline__ void ff_cluster_accum(const double (&pq)[FF_KMAX], double r, double (&c)[FF_NC]) {
    double ri = 1.0;
#pragma unroll
    for (int i = 0; i < FF_NC; ++i) {
        double inner = pq[J - 1] * g_ff2.b[J - 1][i];
#pragma unroll
        for (int jq = J - 2; jq >= 0; --jq) inner = __builtin_fma(pq[jq], g_ff2.b[jq][i], inner);
        c[i] = __builtin_fma(inner, ri, c[i]);
        ri *= r;
    }
}

// What farfield_kernel reads of a walker's records, staged in LDS by one round of vector loads (the records are 512 B
// apart and were read field by field with scalar loads: a memory round trip per item, in a launch that is a chain of them):
// per line K_0..K_8 | A | B | mode (FF_LF doubles), per cluster Q_2..Q_16 | A_c | B_c | Y_0..Y_4 | first, count (FF_CF).
// The workgroup can be FF_WAVES waves over the SAME 64 blocks: wave h takes the items (clusters, then lines) h, h + FF_WAVES,
// ..., and wave 0 adds the others' coefficients to its own, in wave order, through LDS.  Measured with 2: C2 -1 %, C3 -2 %,
// C4 +3 % per pass (profiles/r03_notes.md) -- the launch is bound by its instruction count (~150 per item and wave), not
// by the length of a wave's chain -- so it stays at 1.
constexpr int FF_LF = 12, FF_CF = 23, FF_WAVES = 1;
__host__ __device__ inline size_t farfield_lds_bytes(int L, int NCm) {
    return ((size_t)L * FF_LF + (size_t)NCm * FF_CF + (size_t)(FF_WAVES - 1) * 64 * (FF_NC + 2)) * sizeof(double);
}

template <int M, bool MEMBERS>      // MEMBERS: cluster members may be taken line by line (InstDev::ff_members); the plain instance is free of it
__device__ __forceinline__ void farfield_body(const InstDev& I, const LinesDev& T, const double* __restrict__ lc, int W, int bx) {
    constexpr double XMIN = M >= 9 ? 14.0 : 30.0;
    extern __shared__ double ffs[];
    // one workgroup = 64 blocks of ONE walker (grid.y), lane = block: the walker's records, the cluster tables and all
    // the bookkeeping on them are wave-uniform; per lane only the block's own numbers
    const int nb = I.ntiles * I.ff_nblk;
    const int w = blockIdx.y;
    const int lane = threadIdx.x & 63, half = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int b = min(bx * 64 + lane, nb - 1);
    const bool store = bx * 64 + lane < nb;                     // (lanes past the end repeat the last block)
    const long idx = (long)w * nb + b;
    double* __restrict__ out = I.ff + (size_t)idx * FF_STRIDE;
    const double gc = I.ff_tab[4 * b], hw = I.ff_tab[4 * b + 1];
    const double* __restrict__ recs = lc + (size_t)w * (T.L + T.NCm) * LC_STRIDE;      // (written by the launch before)
    double* __restrict__ sl = ffs;                               // (L, FF_LF)
    double* __restrict__ sq = ffs + (size_t)T.L * FF_LF;         // (NCm, FF_CF)
    {
        const int nl = T.L * FF_LF, nq = T.NCm * FF_CF;
        for (int e = threadIdx.x; e < nl + nq; e += 64 * FF_WAVES) {
            double v;
            if (e < nl) {
                const int l = e / FF_LF, f = e - l * FF_LF;
                const int off = f < 9 ? LC_K0 + f : (f == 9 ? LC_A : (f == 10 ? LC_B : LC_MODE));
                v = recs[(size_t)l * LC_STRIDE + off];           // (mode: the raw 8 bytes)
            } else {
                const int k = (e - nl) / FF_CF, f = (e - nl) - k * FF_CF;
                if (f < 22) {
                    const int off = f < FF_KMAX ? MP_Q0 + f : (f == 15 ? MP_A : (f == 16 ? MP_B : MP_Y0 + (f - 17)));
                    v = recs[(size_t)(T.L + k) * LC_STRIDE + off];
                } else {
                    v = __hiloint2double(T.cl_count[k], T.cl_first[k]);
                }
            }
            ffs[e] = v;
        }
        __syncthreads();
    }
    double c[FF_NC];
#pragma unroll
    for (int j = 0; j < FF_NC; ++j) c[j] = 0.0;
    unsigned long long mask[2] = {0ull, 0ull}, member[2] = {0ull, 0ull};
    // members of clusters that lie so NEAR this block that the tile kernel cannot use the cluster's multipole anywhere in it
    // (every pixel inside the nearest tier's radius): there the tile kernel walks the members one by one, and the far ones
    // among them -- most of a cluster of narrow and wide components -- can go into the block's expansion line by line.
    // solo: per lane (block); visit: members some lane of the wave may take (wave-uniform)
    unsigned long long solo[2] = {0ull, 0ull}, visit[2] = {0ull, 0ull};
    const bool live = hw >= 0.0 && T.L <= 128;                  // (hw < 0: the block holds no pixel)
    // ---- clusters: the multipole series of the whole cluster, where the block lies in one of its far tiers; its
    //      member lines are then covered together (and never one by one: the cluster's own expansion in the tile
    //      kernel must not count a line twice)
    for (int k = 0; k < T.NCm; ++k) {
        const double* __restrict__ mq = sq + (size_t)k * FF_CF;
        const int first = __builtin_amdgcn_readfirstlane(__double2loint(mq[22])), n = __builtin_amdgcn_readfirstlane(__double2hiint(mq[22]));
        unsigned long long rm[2] = {0ull, 0ull};                 // the members' bits (wave-uniform)
        for (int l = first; l < first + n; ++l) rm[l >> 6] |= 1ull << (l & 63);
        member[0] |= rm[0]; member[1] |= rm[1];
        double Qk[FF_KMAX];
#pragma unroll
        for (int jq = 0; jq < FF_KMAX; ++jq) Qk[jq] = mq[jq];
        const double Ac = mq[15], Bc = mq[16], Y0c = mq[17];
        const double yc = __builtin_fma(Ac, gc, -Bc), ayc = fabs(yc), hwy = fabs(Ac) * hw, ynear = ayc - hwy;
        int J = ynear >= mq[21] ? MP_J4 : (ynear >= mq[20] ? MP_J3 : (ynear >= mq[19] ? MP_J2 : (ynear >= mq[18] ? MP_J1 : 0)));
        {
            // (the tile kernel uses the multipole of a pass only if EVERY pixel has |y| >= Y_0: with the block's farthest
            //  pixel inside that radius -- a hair inside, for the roundings of y -- it cannot, whatever the pass looks like)
            const bool inside = MEMBERS && live && (ayc + hwy) * (1.0 + 1e-9) < Y0c;
            if (inside) { solo[0] |= rm[0]; solo[1] |= rm[1]; }
            if (__ballot(inside) != 0ull) { visit[0] |= rm[0]; visit[1] |= rm[1]; }
        }
        if (k % FF_WAVES != half) continue;                     // (the bookkeeping above: every wave; the expansion: one)
        bool ok = live && J > 0 && hwy <= 0.125 * ayc;
        const double iyc = fast_rcp(yc);
        if (ok) {
            const double rho = hwy * fabs(iyc), r2 = rho * rho, r4 = r2 * r2, r12 = r4 * r4 * r4, om = 1.0 - rho;
            ok = fabs(Qk[0]) * (iyc * iyc) * 13.0 * r12 <= 1e-16 * (om * om);
        }
        if (!ok) J = 0;
        const bool j15 = __ballot(J > MP_J2) != 0ull;
        const bool j10 = __ballot(J > MP_J3) != 0ull, j7 = __ballot(J > MP_J4) != 0ull, j5 = __ballot(J > 0) != 0ull;
        if (j5) {
            double pq[FF_KMAX], ip = iyc * iyc;
#pragma unroll
            for (int jq = 0; jq < FF_KMAX; ++jq) { pq[jq] = jq < J ? Qk[jq] * ip : 0.0; ip *= iyc; }
            const double r = (Ac * hw) * iyc;
            if (j15) ff_cluster_accum<MP_J1>(pq, r, c);
            else if (j10) ff_cluster_accum<MP_J2>(pq, r, c);
            else if (j7) ff_cluster_accum<MP_J3>(pq, r, c);
            else ff_cluster_accum<MP_J4>(pq, r, c);
            if (J > 0) { mask[0] |= rm[0]; mask[1] |= rm[1]; }
        }
    }
    // ---- the other lines one by one (and the members of clusters too near for their multipole, see above)
    for (int l = 0; l < T.L; ++l) {
        const bool is_member = (member[l >> 6] >> (l & 63)) & 1ull;
        if (is_member && !(MEMBERS && ((visit[l >> 6] >> (l & 63)) & 1ull))) continue;
        if ((T.NCm + l) % FF_WAVES != half) continue;
        const bool mine = !MEMBERS || !is_member || ((solo[l >> 6] >> (l & 63)) & 1ull);     // (per lane)
        const double* __restrict__ rl = sl + (size_t)l * FF_LF;
        double Kl[M];
#pragma unroll
        for (int mm = 0; mm < M; ++mm) Kl[mm] = rl[mm];
        const double A = rl[9], B = rl[10], K0 = Kl[0];
        const int mode = __double2loint(rl[11]);
        const double xc = __builtin_fma(A, gc, -B), axc = fabs(xc), hwx = fabs(A) * hw;
        bool ok = live && mine && mode == 0 && (axc - hwx >= XMIN) && (hwx <= 0.125 * axc);
        const double ixc = fast_rcp(xc), sc = ixc * ixc;
        if (ok) {
            const double rho = hwx * fabs(ixc), r2 = rho * rho, r4 = r2 * r2, r12 = r4 * r4 * r4, om = 1.0 - rho;
            ok = fabs(K0) * sc * 13.0 * r12 <= 1e-16 * (om * om);      // (NaN: not ok)
        }
        {
            // an item costs the wave ~150 instructions whatever the number of blocks that take it, a covered (block, line)
            // saves the tile kernel ~50: members of near clusters only where at least VP_MIN_LANES blocks want them
            const int nok = __builtin_popcountll(__ballot(ok));
            if (nok == 0 || (MEMBERS && is_member && nok < VP_FF_MIN_LANES)) continue;
        }
        double q[M], sp = sc;
#pragma unroll
        for (int mm = 0; mm < M; ++mm) { q[mm] = ok ? Kl[mm] * sp : 0.0; sp *= sc; }
        const double r = ok ? (A * hw) * ixc : 0.0;
        double rj = 1.0;
#pragma unroll
        for (int j = 0; j < FF_NC; ++j) {
            double inner = q[M - 1] * g_ff.b[M - 1][j];
#pragma unroll
            for (int mm = M - 2; mm >= 0; --mm) inner = __builtin_fma(q[mm], g_ff.b[mm][j], inner);
            c[j] = __builtin_fma(inner, rj, c[j]);
            rj *= r;
        }
        if (ok) mask[l >> 6] |= 1ull << (l & 63);
    }
    if (FF_WAVES > 1) {
        double* __restrict__ xs = ffs + (size_t)T.L * FF_LF + (size_t)T.NCm * FF_CF;      // (FF_WAVES - 1, FF_NC + 2, 64)
        if (half > 0) {
            double* __restrict__ mine = xs + (size_t)(half - 1) * (FF_NC + 2) * 64;
#pragma unroll
            for (int j = 0; j < FF_NC; ++j) mine[j * 64 + lane] = c[j];
            reinterpret_cast<unsigned long long*>(mine)[FF_NC * 64 + lane] = mask[0];
            reinterpret_cast<unsigned long long*>(mine)[(FF_NC + 1) * 64 + lane] = mask[1];
        }
        __syncthreads();
        if (half > 0) return;
#pragma unroll
        for (int h = 1; h < FF_WAVES; ++h) {
            const double* __restrict__ oth = xs + (size_t)(h - 1) * (FF_NC + 2) * 64;
#pragma unroll
            for (int j = 0; j < FF_NC; ++j) c[j] += oth[j * 64 + lane];
            mask[0] |= reinterpret_cast<const unsigned long long*>(oth)[FF_NC * 64 + lane];
            mask[1] |= reinterpret_cast<const unsigned long long*>(oth)[(FF_NC + 1) * 64 + lane];
        }
    }
    if (!store) return;
#pragma unroll
    for (int j = 0; j < FF_NC; ++j) out[j] = c[j];
    reinterpret_cast<unsigned long long*>(out)[FF_MASK0] = mask[0];
    reinterpret_cast<unsigned long long*>(out)[FF_MASK0 + 1] = mask[1];
}
template <int M, bool MEMBERS>
__global__ __launch_bounds__(64 * FF_WAVES) void farfield_kernel(InstDev I, LinesDev T, const double* __restrict__ lc, int W) {
    farfield_body<M, MEMBERS>(I, T, lc, W, (int)blockIdx.x);
}
// Two instruments of a joint fit that share their records (same line tables): the blocks of both in ONE launch
// (grid.x = nbx0 + nbx1 waves of blocks; C3: two launches of 22 us that each leave the GPU half empty).
template <int M, bool MEMBERS>
__global__ __launch_bounds__(64 * FF_WAVES) void farfield_kernel2(InstDev I0, InstDev I1, int nbx0, LinesDev T, const double* __restrict__ lc, int W) {
    if ((int)blockIdx.x < nbx0) farfield_body<M, MEMBERS>(I0, T, lc, W, (int)blockIdx.x);
    else farfield_body<M, MEMBERS>(I1, T, lc, W, (int)blockIdx.x - nbx0);
}

// ---------------------------------------------------------------------------------------------
// tile work
// ---------------------------------------------------------------------------------------------
constexpr int TILE_THREADS_MAX = 256;   // tile_kernel: 1, 2 or 4 waves per workgroup
constexpr int WALKER_THREADS_MAX = 1024; // walker_kernel: one wave per tile, up to 16 tiles per walker
constexpr int FL_PAD = 10;        // LDS doubles after a tile's flux that the zero-padded taps may read
// Diagnostic build (-DVP_STAMPS, scripts/walker_timeline.py): every wave of walker_kernel leaves the shader clock at its
// phase boundaries in g_stamps[walker][wave][stage]; read back with vp_debug_read_stamps.  Not compiled otherwise.
// Stages: 0 entry, 1 records ready (behind the first barrier), 2 end of phase A, 3 end of phase B, 4 end of LSF + chi^2,
// 5 behind the last barrier, 6 start of phase B (tiles with line cores), 7 HW_ID | XCC_ID << 32 (where the wave runs),
// 8 kernel arguments have arrived, 9 theta row has arrived, 10 record tasks issued, 11 stores drained and preloads back.
// (Ablation builds: VP_DIAG at the top of this file.)
#ifdef VP_STAMPS
constexpr int STAMP_W = 1024, STAMP_WAVES = 16, STAMP_STAGES = 16;
__device__ long long g_stamps[STAMP_W * STAMP_WAVES * STAMP_STAGES];
#define VP_STAMP(stage) do { if (g_stamp_w >= 0 && g_stamp_w < STAMP_W && (threadIdx.x & 63) == 0) \
    g_stamps[(g_stamp_w * STAMP_WAVES + (int)(threadIdx.x >> 6)) * STAMP_STAGES + (stage)] = (long long)__builtin_readcyclecounter(); } while (0)
#define VP_STAMP_RT(stage) do { if (g_stamp_w >= 0 && g_stamp_w < STAMP_W && (threadIdx.x & 63) == 0) \
    g_stamps[(g_stamp_w * STAMP_WAVES + (int)(threadIdx.x >> 6)) * STAMP_STAGES + (stage)] = (long long)wall_clock64(); } while (0)   /* 100 MHz, one base for all XCDs */
#define VP_STAMP_DECL const int g_stamp_w = (int)blockIdx.x;
#define VP_STAMP_ARG , const int g_stamp_w
#define VP_STAMP_PASS , g_stamp_w
#define VP_STAMP_NONE , -1
#else
#define VP_STAMP(stage) do { } while (0)
#define VP_STAMP_RT(stage) do { } while (0)
#define VP_STAMP_DECL
#define VP_STAMP_ARG
#define VP_STAMP_PASS
#define VP_STAMP_NONE
#endif
#ifndef VP_TILE_WPE
#define VP_TILE_WPE 1             // tile_kernel: waves per SIMD the register allocation must leave room for
#endif
#ifndef VP_PRIO_LEVEL
#define VP_PRIO_LEVEL 2           // issue priority of a wave whose tile holds line cores (ablations: 1, 3)
#endif
#ifndef VP_CORE_ILP
#define VP_CORE_ILP 2             // phase B of single-wave tiles in walker_kernel: flagged chunks evaluated side by side
#endif
constexpr int RB = 3;             // 64-pixel chunks per wave pass (register blocking / ILP); measured: RB=3
                                  // beats RB=2 and RB=4 by 3-5 %

// s * Horner_M(K, s) for RB independent chunks with the same M; K wave-uniform (SGPR operands).
template <int M, class KP>     // KP: rec_t (scalar loads on demand) or a local array already held in SGPRs
__device__ __forceinline__ void wing_rb(const double (&x)[RB], KP K, double (&tau)[RB]) {
    double s[RB], acc[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) s[r] = fast_rcp1(x[r] * x[r]);
    // (one reciprocal for the pass's three chunks -- 1 / (x0^2 x1^2 x2^2), then products; v_rcp_f64 issues at a quarter of the FMA
    //  rate -- measured 1 % faster unguarded, but a NaN wavelength sample would then poison its lane's other two pixels, and
    //  both guards that keep it in its own pixel cost more than the trick saves: profiles/r05_experiments/ab3..ab5)
    const double kt = K[M - 1];
#pragma unroll
    for (int r = 0; r < RB; ++r) acc[r] = kt;
#pragma unroll
    for (int m = M - 2; m >= 0; --m) {
        const double km = K[m];
#pragma unroll
        for (int r = 0; r < RB; ++r) acc[r] = __builtin_fma(acc[r], s[r], km);
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) tau[r] = __builtin_fma(acc[r], s[r], tau[r]);
}

// sum_{j=2}^{J+1} Q_j y^-j for RB independent chunks: the first J terms of a cluster's multipole series.
template <int J>
__device__ __forceinline__ void multipole_rb(const double (&y)[RB], rec_t Q, double (&tau)[RB]) {
    double q[RB], acc[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) q[r] = fast_rcp1(y[r]);
    const double qt = Q[J - 1];
#pragma unroll
    for (int r = 0; r < RB; ++r) acc[r] = qt;
#pragma unroll
    for (int jq = J - 2; jq >= 0; --jq) {
        const double qj = Q[jq];
#pragma unroll
        for (int r = 0; r < RB; ++r) acc[r] = __builtin_fma(acc[r], q[r], qj);
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) tau[r] = __builtin_fma(acc[r], q[r] * q[r], tau[r]);
}

// The eager block of a record: A, B, K0..K5 (one 64-byte scalar load) + mode.
struct Eager {            // prefetched one line ahead: what the tier decision needs.  (Fetching K0..K5 along with it, so
    double A, B;          // that the far tiers issue no load behind their decision, measured 0.5-2 % SLOWER on C1-C4:
    int mode, mp, cl_end; // the exposed scalar-load latency is covered by the other waves, the 20 extra SGPR spills are not.)
    int touch1, touch2;   // TOUCH (walker_kernel): see load_eager
};
template <bool TOUCH>
__device__ __forceinline__ Eager load_eager(rec_t rec) {
    Eager e;
    e.A = rec[LC_A];
    e.B = rec[LC_B];
    e.mode = rec_int(rec, LC_MODE, 0);
    e.touch1 = 0; e.touch2 = 0;
    if (TOUCH) {
        // pull the record's second and third 64-byte lines (K6..K13; the faithful-x constants) into the scalar cache
        // with this prefetch: the loads behind a tier decision then hit.  (Two dwords nobody computes with; they are
        // "used" one line later, when they have long arrived, so that the compiler keeps the loads.)  Worth 0.3 us of
        // 18.3 in the latency-bound walker kernel; costs 1-3 % where the SIMDs are full (tile_kernel: off).
        e.touch1 = rec_int(rec, 8, 0);
        e.touch2 = rec_int(rec, 16, 0);
    }
    e.mp = rec_int(rec, LC_CL, 0);
    e.cl_end = rec_int(rec, LC_CL, 1);
    return e;
}

// Phase B of one flagged 64-pixel chunk c of a tile (its raw tau is in the tile's LDS window `fl`, the lines to add
// in its mask words `cm`): core series for every lane (no divergence), 14-term wing value for lanes of a mixed
// chunk beyond the core radius, then exp.  `daw` / `etab`: the Dawson and exp tables in LDS.
template <bool GENERIC>
__device__ __forceinline__ void core_chunk(const InstDev& I, rec_t lcw, double* __restrict__ fl,
                                           const unsigned long long* __restrict__ cm, int nwords, int q0, int n_eval, int c,
                                           int lane, const double* __restrict__ daw, const double* __restrict__ etab) {
    const int i = c * 64 + lane;
    const int ic = min(i, n_eval - 1);
    const int q = min(max(q0 + ic, 0), I.P - 1);
    const double gq = I.ginv[q], wq = I.wave[q];
    double tau = fl[ic];
    for (int wd = 0; wd < nwords; ++wd) {
        unsigned long long m = cm[c * nwords + wd];
        {   // scalarise (mind the sign: readfirstlane returns int)
            const unsigned int mlo = (unsigned int)__builtin_amdgcn_readfirstlane((unsigned int)m);
            const unsigned int mhi = (unsigned int)__builtin_amdgcn_readfirstlane((unsigned int)(m >> 32));
            m = ((unsigned long long)mhi << 32) | (unsigned long long)mlo;
        }
        while (m) {
            const int l = (wd << 6) + __builtin_ctzll(m);
            m &= m - 1;
            rec_t rec = lcw + (size_t)l * LC_STRIDE;
            const int mode = rec_int(rec, LC_MODE, 0);
            if (mode != 0) {
                if (GENERIC) tau += cold_line_tau(wq, gq, rec);
                else tau = __builtin_nan("");          // poisoned line (non-finite constants)
                continue;
            }
            const double xf = faithful_x(wq, gq, rec);
            const double xa = fabs(xf);
            const int nodd = rec_int(rec, LC_MODE, 1);
            double h = rec[LC_T] * core_taylor_H_lds(xf, rec[LC_Y], rec[LC_EA2], nodd, daw, etab);
            if (__ballot(xa >= X_CORE) != 0ull) {
                const double rw = wing_tau<NWING>(xf, rec + LC_K0);
                h = (xa >= X_CORE) ? rw : h;
            }
            tau += h;
        }
    }
    if (i < n_eval) fl[i] = (tau != tau) ? tau : exp_neg_tab(tau, etab);   // NaN survives (poisoned lines)
}

// N flagged chunks at once: per line of the UNION of their masks the core series runs for all of them side by side
// (core_taylor_H_lds_n<N>: N times the instruction-level parallelism for a wave that has its SIMD to itself -- the
// walker kernel's waves with line cores are the last ones running); a chunk that did not flag the line has its wing
// value from phase A already and drops the result.  Per chunk the same operations in the same order as core_chunk:
// bit-identical.
template <bool GENERIC, int N>
__device__ __forceinline__ void core_chunk_multi(const InstDev& I, rec_t lcw, double* __restrict__ fl,
                                                 const unsigned long long* __restrict__ cm, int nwords, int q0, int n_eval,
                                                 const int (&cc)[N], int lane, const double* __restrict__ daw,
                                                 const double* __restrict__ etab) {
    int i[N];
    double gq[N], wq[N], tau[N];
#pragma unroll
    for (int r = 0; r < N; ++r) {
        i[r] = cc[r] * 64 + lane;
        const int ic = min(i[r], n_eval - 1);
        const int q = min(max(q0 + ic, 0), I.P - 1);
        gq[r] = I.ginv[q]; wq[r] = I.wave[q];
        tau[r] = fl[ic];
    }
    for (int wd = 0; wd < nwords; ++wd) {
        unsigned long long mr[N], m = 0ull;
#pragma unroll
        for (int r = 0; r < N; ++r) {
            const unsigned long long mw = cm[cc[r] * nwords + wd];
            const unsigned int mlo = (unsigned int)__builtin_amdgcn_readfirstlane((unsigned int)mw);
            const unsigned int mhi = (unsigned int)__builtin_amdgcn_readfirstlane((unsigned int)(mw >> 32));
            mr[r] = ((unsigned long long)mhi << 32) | (unsigned long long)mlo;
            m |= mr[r];
        }
        while (m) {
            const int b = __builtin_ctzll(m);
            const int l = (wd << 6) + b;
            m &= m - 1;
            rec_t rec = lcw + (size_t)l * LC_STRIDE;
            const int mode = rec_int(rec, LC_MODE, 0);
            if (mode != 0) {
#pragma unroll
                for (int r = 0; r < N; ++r) {
                    if ((mr[r] >> b) & 1ull) {
                        if (GENERIC) tau[r] += cold_line_tau(wq[r], gq[r], rec);
                        else tau[r] = __builtin_nan("");      // poisoned line (non-finite constants)
                    }
                }
                continue;
            }
            double xf[N], h[N];
            bool outside = false;
#pragma unroll
            for (int r = 0; r < N; ++r) {
                xf[r] = faithful_x(wq[r], gq[r], rec);
                outside = outside || (fabs(xf[r]) >= X_CORE);
            }
            const int nodd = rec_int(rec, LC_MODE, 1);
            core_taylor_H_lds_n<N>(xf, rec[LC_Y], rec[LC_EA2], nodd, daw, etab, h);
            const double T = rec[LC_T];
#pragma unroll
            for (int r = 0; r < N; ++r) h[r] = T * h[r];
            if (__ballot(outside) != 0ull) {
#pragma unroll
                for (int r = 0; r < N; ++r) {
                    const double rw = wing_tau<NWING>(xf[r], rec + LC_K0);
                    h[r] = (fabs(xf[r]) >= X_CORE) ? rw : h[r];
                }
            }
#pragma unroll
            for (int r = 0; r < N; ++r)
                if ((mr[r] >> b) & 1ull) tau[r] += h[r];
        }
    }
#pragma unroll
    for (int r = 0; r < N; ++r)
        if (i[r] < n_eval) fl[i[r]] = (tau[r] != tau[r]) ? tau[r] : exp_neg_tab(tau[r], etab);
}

// Synchronisation among the threads that share one tile.  SOLO: the tile belongs to ONE wave of a
// larger workgroup (walker_kernel) and must not wait for the others: the LDS operations of a wave
// execute in program order, so only the compiler has to be kept from reordering them.
template <bool SOLO>
__device__ __forceinline__ void tile_sync() {
    if (SOLO) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

// What a tile's threads fetch before they need the line records (LSF tap, exp table entry, the first
// pass's pixel data): issued first, so that a wave's start-up is one memory round trip instead of
// four -- and, in walker_kernel, runs under the record preparation.
struct TilePre { double e2, g[RB], wv[RB]; };
__device__ __forceinline__ TilePre tile_preload(const InstDev& I, int p0, int nout, int tid) {
    const int lane = tid & 63, wid = tid >> 6;
    const int n_eval = nout + I.K - 1;
    const int q0 = p0 - I.halo_lo;
    TilePre pre;
    pre.e2 = exp2_eighth(tid & 7);
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const int i = min(wid * (64 * RB) + r * 64 + lane, n_eval - 1);
        const int q = min(max(q0 + i, 0), I.P - 1);
        pre.g[r] = I.ginv[q];
        pre.wv[r] = I.wave[q];
    }
    return pre;
}

// LSF + chi^2 (or flux output) for one block of LSF_PX * nthreads output pixels starting at `ob`: each lane produces LSF_PX
// CONSECUTIVE pixels from one sliding window of the flux held in registers -- per group of 8 taps 8 + LSF_PX - 1 doubles
// read for 8 LSF_PX FMAs.  The phase is bound by LDS bandwidth, not by its FMAs (C2: 22 of the tile kernel's 102 us with
// two pixels per lane, which read 10 doubles per 16 FMAs): six pixels per lane read 14 per 48, and their lane stride of
// 48 B keeps the 16-byte reads conflict-free (8 lanes x 16 B cover the 32 banks once).  Per output the taps are
// accumulated in ascending order, as in the plain loop.
constexpr int LSF_PX = 6;
constexpr int OBS_PAD = 8;         // zeroed doubles behind InstDev::flux / w (capi: vp_add_instrument)
typedef double obs2_t __attribute__((ext_vector_type(2)));
typedef obs2_t obs2_u_t __attribute__((aligned(8)));                              // (8-byte aligned: any even OR odd pixel)
typedef const obs2_u_t __attribute__((address_space(1)))* obs2_ptr_t;             // ... in global memory
template <int OUT, bool EARLY, bool LATE = false, bool NANFIX = false>    // EARLY: the observed pixels are requested ahead of the taps (walker_kernel, below)
                                                     // LATE: spectrum and tap pointers re-read from the kernarg segment (tile_kernel1)
                                                     // NANFIX: outputs times InstDev::rbot
__device__ __forceinline__ void lsf_block6(const InstDev& I, const double* __restrict__ fl, int kn, int ob, int nout, int p0, int w,
                                           int tid, double* __restrict__ out, int out_stride, double& acc) {
    const int o0 = ob + LSF_PX * tid;                                         // even
    const int oc = min(o0, ((nout - 1) / LSF_PX) * LSF_PX);                   // lanes past the end repeat the last group
    const double2* __restrict__ fw = reinterpret_cast<const double2*>(fl + oc);
    double m[LSF_PX];
#pragma unroll
    for (int p = 0; p < LSF_PX; ++p) m[p] = 0.0;
    const gptr_t pflux = VP_LATE_GPTR(LATE, I, flux);
    const gptr_t pw = VP_LATE_GPTR(LATE, I, w);
    const double* __restrict__ pk = VP_LATE_FIELD(LATE, I, kflip);
    // the observed spectrum and its weights for this lane's pixels: in walker_kernel requested now, so that they travel
    // while the taps are applied -- behind the loop they are a memory round trip at the very end of the wave's, and the
    // workgroup's, life (C1 at 256 walkers 16.6 -> 16.3 us); the tile launches, whose workgroups cover for each other,
    // lose 1 % to the registers this holds (C2 / C3) and ask behind the loop
    double fobs[LSF_PX], wobs[LSF_PX];
    // six consecutive pixels per lane as three 16-byte loads per array (the arrays end in OBS_PAD zeroed doubles: a lane past the
    // tile's last output reads into them -- or into the next tile's pixels -- and its terms are dropped below): one address
    // instead of six clamped ones, 6 load instructions instead of 12 (C1 at 512 walkers 21.6 -> 21.3 us)
    auto load_obs = [&]() {
        const int pxb = min(p0 + o0, I.P + OBS_PAD - LSF_PX);
        const obs2_ptr_t f2 = (obs2_ptr_t)(pflux + pxb);
        const obs2_ptr_t w2 = (obs2_ptr_t)(pw + pxb);
#pragma unroll
        for (int p = 0; p < LSF_PX; p += 2) {
            const obs2_t fv = f2[p >> 1], wv2 = w2[p >> 1];
            fobs[p] = fv.x; fobs[p + 1] = fv.y; wobs[p] = wv2.x; wobs[p + 1] = wv2.y;
        }
    };
    if (OUT == 0 && EARLY) load_obs();
    constexpr int NW = (8 + LSF_PX - 1 + 1) / 2;                              // 16-byte reads per group of 8 taps
    for (int j = 0; j < kn; j += 8) {
        rec_t kb = as_rec(pk) + j;                       // uniform address: scalar loads, SGPR operands of the FMAs
        double f[2 * NW];
#pragma unroll
        for (int q = 0; q < NW; ++q) {
            const double2 v = fw[(j >> 1) + q];
            f[2 * q] = v.x; f[2 * q + 1] = v.y;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const double kj = kb[u];
#pragma unroll
            for (int p = 0; p < LSF_PX; ++p) m[p] = __builtin_fma(kj, f[u + p], m[p]);
        }
    }
    if (OUT == 0 && !EARLY) load_obs();
    if (OUT == 0 && !NANFIX) {
        // outputs of this lane that do not exist (the tile's last lanes) enter with weight zero -- their model values are finite
        // wherever the tile's own are (the window's tail is the zero padding), their observed pixels the next tile's or the arrays'
        // padding: a compare and two selects per pixel instead of the masked accumulate's six instructions
        const int nv = nout - o0;
#pragma unroll
        for (int p = 0; p < LSF_PX; ++p) {
            const double wp = p < nv ? wobs[p] : 0.0;
            const double d = fobs[p] - m[p];
            acc = __builtin_fma(d * d, wp, acc);           // (flux-model)^2 * inv_sigma2
        }
        return;
    }
    if (OUT != 0 && !NANFIX && o0 + LSF_PX <= nout) {
        // flux output, a lane with all six pixels inside the tile: three 16-byte stores from one address (the row's last lanes,
        // which may not write past the tile, take the pixel-by-pixel form below)
        typedef obs2_u_t __attribute__((address_space(1)))* out2_ptr_t;
        const out2_ptr_t o2 = (out2_ptr_t)((const double __attribute__((address_space(1)))*)reinterpret_cast<unsigned long long>(out) +
                                           ((size_t)w * out_stride + p0 + o0));
#pragma unroll
        for (int p = 0; p < LSF_PX; p += 2) {
            obs2_t v;
            v.x = m[p]; v.y = m[p + 1];
            o2[p >> 1] = v;
        }
        return;
    }
#pragma unroll
    for (int p = 0; p < LSF_PX; ++p) {
        const int px = p0 + o0 + p;
        if (o0 + p < nout) {
            const double mp = NANFIX ? m[p] * I.rbot[px] : m[p];
            if (OUT == 0) {
                const double d = fobs[p] - mp;
                acc = __builtin_fma(d * d, wobs[p], acc);     // (flux-model)^2 * inv_sigma2
            } else {
                out[(size_t)w * out_stride + px] = mp;
            }
        }
    }
}

// FF: the first line >= `from` of the 64-line word [l0, l1) whose bit in `nearm` is set (not covered by the block's
// far-field expansion), or l1.
__device__ __forceinline__ int next_near_line(unsigned long long nearm, int from, int l0, int l1) {
    const int k = from - l0;
    if (k >= 64) return l1;
    const unsigned long long m = nearm >> k;
    return m ? min(l1, from + (int)__builtin_ctzll(m)) : l1;
}

// One tile of one walker -- output pixels [p0, p0 + nout) -- : tau -> exp -> LSF -> chi^2 (OUT = 0; returns this
// LANE's sum of chi^2 terms) or model flux written to `out` (OUT = 1 convolved, 2 unconvolved).  `first` (uniform):
// the LDS tables (taps, exp table) are staged; a wave that works through several tiles in a row passes false from
// its second tile on.  PRE: the tables' entries and the first pass's pixels come from `pre` (tile_preload, issued
// before the records were needed); without it they are loaded here.  PAIR (single-wave tiles): phase B takes the
// flagged chunks VP_CORE_ILP at a time (core_chunk_multi).
// `nthreads` threads (tid = 0..nthreads-1, whole waves) share the LDS block `fl`:
//   fl[span + FL_PAD] tau -> flux (+ zeros) | red[4] | Dawson table | exp table | per-chunk "line core" masks
// GENERIC = false: the fast instance (no out-of-line generic Faddeeva, 77 VGPRs); lines outside the
// fast domain poison tau with NaN there, their walkers belong to the GENERIC = true launch.
template <int METHOD, int OUT, bool GENERIC, bool SOLO, bool PRE = true, bool PAIR = false, bool FF = false, bool W1 = false, bool NANFIX = false,
          bool MP = true>      // MP = false: the multipole path is not compiled.  (Tried for walker_kernel's plain instances, whose records hold no
                               //  clusters: 1.4 KB less code, 256 walkers 15.55 -> 15.47 us, 512 unchanged, but 1.3 % MORE VALU instructions
                               //  issued per eval -- another schedule of the pass loop; not used.  profiles/r05_experiments/ab3)
__device__ __forceinline__ double tile_work(const InstDev& I, rec_t lcw, double* __restrict__ fl, int p0, int nout, int w,
                                            int tid, int nthreads, const TilePre& pre, bool first,
                                            double* __restrict__ out, int out_stride VP_STAMP_ARG, bool daw_ready = false,
                                            int tix = 0) {
    const int n_eval = nout + I.K - 1;
    const int q0 = p0 - I.halo_lo;
    // ONE: the tile's threads are ONE wave, known at compile time -- SOLO (a wave of walker_kernel) or W1 (tile_kernel1: the
    // single-wave workgroups of the tile launches): no wave index, no workgroup barriers, fewer scalars to keep
    constexpr bool ONE = SOLO || W1;
    const int lane = ONE ? tid : (tid & 63);                                   // ONE: tid IS the lane
    const int wid = ONE ? 0 : __builtin_amdgcn_readfirstlane(tid >> 6);        // wave-uniform, and the compiler knows it
    const int TILE_THREADS = ONE ? 64 : nthreads, nwaves = ONE ? 1 : (nthreads >> 6);
    double* __restrict__ daw = fl + I.span + FL_PAD + 4;   // Dawson table for the line cores (16-B aligned)
    const int Kp = (I.K + 7) & ~7;                     // taps are zero-padded to whole groups of 8 (in HBM, host side)
    double* __restrict__ etab = daw + DAW_LDS_DOUBLES; // 2^(j/64) for the table-driven exp
    unsigned long long* __restrict__ cmask = reinterpret_cast<unsigned long long*>(etab + EXP_LDS_DOUBLES);
    const int nwords = (I.L + 63) >> 6;                // 64 lines per mask word
    const int nchunks = (n_eval + 63) >> 6;
    if (first && tid < EXP_LDS_DOUBLES) etab[tid] = PRE ? pre.e2 : exp2_eighth(tid);
    if (tid < FL_PAD) fl[n_eval + tid] = 0.0;   // what the zero taps multiply must be finite
    // Multi-wave tiles: the exp table is staged by the first wave and read by ALL waves at the end of their first pass.
    // A pass used to be long enough for that never to matter; with the far-field expansions a pass over a block
    // without near lines is ~60 instructions, and a wave could read the table before it was there (C3's 4-wave
    // instrument: one wrong walker in ~1000 launches, caught by the full-size test's repeat check).
    if (!ONE && nwaves > 1) __syncthreads();

    // ---- phase A: optical depth of every line that is >= 8 Doppler widths away from the chunk.
    //      Each wave owns 3 x 64 consecutive evaluated pixels per pass (RB chunks); lines are the
    //      outer loop so that a line's constants are fetched once and feed RB independent
    //      evaluations.  Chunks that touch a line core (or a line outside the fast domain) are only
    //      flagged; their tau goes to LDS raw and phase B finishes them.
    bool wave_any = false;                  // did this wave flag any (chunk, line) for phase B?  (wave-uniform)
    for (int base = wid * (64 * RB); base < n_eval; base += TILE_THREADS * RB) {
        double g[RB], wv[RB], tau[RB];
        if (PRE && base == wid * (64 * RB)) {             // first pass: loaded by tile_preload
#pragma unroll
            for (int r = 0; r < RB; ++r) { g[r] = pre.g[r]; wv[r] = pre.wv[r]; tau[r] = 0.0; }
        } else {
            const gptr_t pg = VP_LATE_GPTR(W1, I, ginv), pwv = VP_LATE_GPTR(W1, I, wave);      // (once per pass)
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const int i = min(base + r * 64 + lane, n_eval - 1);
                const int q = min(max(q0 + i, 0), I.P - 1);   // edge replication = evaluate the clamped pixel
                g[r] = pg[q];
                wv[r] = pwv[q];
                tau[r] = 0.0;
            }
        }
        bool pending[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) pending[r] = false;
        // FF: this pass's block of the walker's far-field expansions (tile tix, block base / (64 RB))
        const int fblk = FF ? tix * I.ff_nblk + base / (64 * RB) : 0;
        rec_t ffr = FF ? as_rec(VP_LATE_FIELD(W1, I, ff) + ((size_t)w * (I.ntiles * I.ff_nblk) + fblk) * FF_STRIDE) : (rec_t)0;
        if (METHOD == 0) {
            for (int l0 = 0; l0 < I.L; l0 += 64) {
                unsigned long long todo[RB];
#pragma unroll
                for (int r = 0; r < RB; ++r) todo[r] = 0ull;
                const int l1 = min(I.L, l0 + 64);
                // FF: the lines of this word that the block's expansion covers are not walked at all
                unsigned long long nearm = ~0ull;
                if (FF) {
                    const unsigned int flo = (unsigned int)rec_int(ffr, FF_MASK0 + (l0 >> 6), 0);
                    const unsigned int fhi = (unsigned int)rec_int(ffr, FF_MASK0 + (l0 >> 6), 1);
                    nearm = ~(((unsigned long long)fhi << 32) | (unsigned long long)flo);
                }
#define VP_NEXT_LINE(from) (FF ? next_near_line(nearm, (from), l0, l1) : (from))
                const int lfirst = VP_NEXT_LINE(l0);
                Eager nxt = load_eager<SOLO>(lcw + (size_t)min(lfirst, I.L - 1) * LC_STRIDE);
                for (int l = lfirst; l < l1; l = VP_NEXT_LINE(l + 1)) {
                    // far from a whole cluster of components?  one multipole evaluation replaces all of
                    // its member lines (only tried at the first line of a cluster that fits this block)
                    const int mp = (MP && I.line_sel < 0) ? nxt.mp : -1;
                    if (MP && mp >= 0 && nxt.cl_end <= l1) {
                        rec_t mrec = lcw + (size_t)(I.L + mp) * LC_STRIDE;
                        // A, B and the four tier radii arrive with ONE scalar load (no load behind a tier decision
                        // except the Q_j themselves)
                        const double Ac = mrec[MP_A], Bc = mrec[MP_B];
                        const double y0 = mrec[MP_Y0], y1 = mrec[MP_Y0 + 1], y2 = mrec[MP_Y0 + 2], y3 = mrec[MP_Y0 + 3],
                                     y4 = mrec[MP_Y0 + 4];
                        double y[RB];
#pragma unroll
                        for (int r = 0; r < RB; ++r) y[r] = __builtin_fma(Ac, g[r], -Bc);
                        double ym = fabs(y[0]);
#pragma unroll
                        for (int r = 1; r < RB; ++r) ym = fmin(ym, fabs(y[r]));
                        const bool far0 = __ballot(!(ym >= y0)) == 0ull;      // every lane far enough (NaN counts as near)
                        const bool far1 = VP_NONE_BELOW(ym, y1), far2 = VP_NONE_BELOW(ym, y2), far3 = VP_NONE_BELOW(ym, y3),
                                   far4 = VP_NONE_BELOW(ym, y4);
                        if (far0 && VP_ABL(ABL_NOMP)) {
                            l = max(l, nxt.cl_end - 1);
                            nxt = load_eager<SOLO>(lcw + (size_t)min(VP_NEXT_LINE(l + 1), I.L - 1) * LC_STRIDE);
                            continue;
                        }
                        if (far0) {
                            if (far2) {
                                if (far3) {
                                    if (far4) multipole_rb<MP_J4>(y, mrec + MP_Q0, tau);
                                    else multipole_rb<MP_J3>(y, mrec + MP_Q0, tau);
                                } else {
                                    multipole_rb<MP_J2>(y, mrec + MP_Q0, tau);
                                }
                            } else if (far1) {
                                multipole_rb<MP_J1>(y, mrec + MP_Q0, tau);
                            } else {
                                multipole_rb<MP_J0>(y, mrec + MP_Q0, tau);
                            }
                            l = max(l, nxt.cl_end - 1);                // skip the member lines (never backwards, whatever a record holds)
                            nxt = load_eager<SOLO>(lcw + (size_t)min(VP_NEXT_LINE(l + 1), I.L - 1) * LC_STRIDE);
                            continue;
                        }
                    }
                    const Eager cur = nxt;
                    if (SOLO) asm volatile("" :: "s"(cur.touch1), "s"(cur.touch2));
                    if (I.line_sel >= 0 && l != I.line_sel) {
                        nxt = load_eager<SOLO>(lcw + (size_t)min(VP_NEXT_LINE(l + 1), I.L - 1) * LC_STRIDE);
                        continue;
                    }
                    rec_t rec = lcw + (size_t)l * LC_STRIDE;
                    nxt = load_eager<SOLO>(lcw + (size_t)min(VP_NEXT_LINE(l + 1), I.L - 1) * LC_STRIDE);   // scalar prefetch of the next line
                    const unsigned long long bit = 1ull << (l - l0);
                    const double A = cur.A, B = cur.B;
                    rec_t K = rec + LC_K0;
                    double x[RB];
#pragma unroll
                    for (int r = 0; r < RB; ++r) x[r] = __builtin_fma(A, g[r], -B);
                    if (cur.mode != 0) {
#pragma unroll
                        for (int r = 0; r < RB; ++r) todo[r] |= bit;
                        continue;
                    }
                    double xm = fabs(x[0]);
#pragma unroll
                    for (int r = 1; r < RB; ++r) xm = fmin(xm, fabs(x[r]));
                    if (VP_NONE_BELOW(xm, 30.0)) {        // the 1-FMA x is accurate enough out here
                        if (VP_ABL(ABL_NOFAR)) continue;
                        if (VP_NONE_BELOW(xm, 100.0)) {
                            if (VP_NONE_BELOW(xm, 3000.0)) wing_rb<2>(x, K, tau);
                            else if (VP_NONE_BELOW(xm, 500.0)) wing_rb<3>(x, K, tau);
                            else wing_rb<4>(x, K, tau);
                        } else {
                            wing_rb<6>(x, K, tau);
                        }
                        continue;
                    }
                    if (VP_ABL(ABL_NONEAR) && VP_NONE_BELOW(xm, X_CORE)) continue;
                    double xf[RB];
#pragma unroll
                    for (int r = 0; r < RB; ++r) xf[r] = faithful_x(wv[r], g[r], rec);
                    if (VP_NONE_BELOW(xm, 14.0)) { wing_rb<9>(xf, K, tau); continue; }
                    if (VP_NONE_BELOW(xm, X_CORE)) { wing_rb<NWING>(xf, K, tau); continue; }
                    // some chunk of this wave touches the line core: flag it for phase B; the other
                    // chunks get their 14-term wing value here (wave-uniform branches)
#pragma unroll
                    for (int r = 0; r < RB; ++r) {
                        if (VP_NONE_BELOW(fabs(x[r]), X_CORE)) tau[r] += wing_tau<NWING>(xf[r], K);
                        else todo[r] |= bit;
                    }
                }
#pragma unroll
                for (int r = 0; r < RB; ++r) {
                    pending[r] = pending[r] || (todo[r] != 0ull);
                    wave_any = wave_any || (todo[r] != 0ull);
                    const int c = (base >> 6) + r;
                    if (lane == 0 && c < nchunks) cmask[c * nwords + (l0 >> 6)] = todo[r];
                }
            }
#undef VP_NEXT_LINE
            if (FF && !VP_ABL(ABL_NOFFPOLY)) {      // all the far lines of this block at once: FF_NC FMAs per pixel
                rec_t tb = as_rec(VP_LATE_FIELD(W1, I, ff_tab)) + 4 * fblk;
                const double ihw = tb[2], gci = tb[3];
                double t[RB], pa[RB];
                const double ctop = ffr[FF_NC - 1];
#pragma unroll
                for (int r = 0; r < RB; ++r) { t[r] = __builtin_fma(g[r], ihw, -gci); pa[r] = ctop; }
#pragma unroll
                for (int j = FF_NC - 2; j >= 0; --j) {
                    const double cj = ffr[j];
#pragma unroll
                    for (int r = 0; r < RB; ++r) pa[r] = __builtin_fma(pa[r], t[r], cj);
                }
#pragma unroll
                for (int r = 0; r < RB; ++r) tau[r] += pa[r];
            }
        } else if (METHOD == 1) {
            for (int l = 0; l < I.L; ++l) {
                if (I.line_sel >= 0 && l != I.line_sel) continue;
                rec_t rec = lcw + (size_t)l * LC_STRIDE;
                // Far from the line (every |x| >= 28, i.e. x^2 >= 784): G = exp(-x^2) underflows to exactly 0
                // in double, so the reference's expression (voigt_approx.py:79-81) reduces to
                //   H = (a/sqrt(pi)) * 1.5 / (x^2 (x^2+1)^2)      [its x^-6 wing, trap T9]
                // -- no exp, no division sequence, and the 1-FMA x is plenty (H <= 1e-9 a out here).
                const double a_l = rec[LC_Y];
                const double eps_l = fmax(1e-2, 100.0 * fabs(a_l) / 1.7724538509055160273);
                double xc[RB];
#pragma unroll
                for (int r = 0; r < RB; ++r) xc[r] = __builtin_fma(rec[LC_A], g[r], -rec[LC_B]);
                double xmf = fabs(xc[0]);
#pragma unroll
                for (int r = 1; r < RB; ++r) xmf = fmin(xmf, fabs(xc[r]));
                if (eps_l <= 784.0 && VP_NONE_BELOW(xmf, 28.0)) {
                    const double pref = rec[LC_T] * ((a_l / 1.7724538509055160273) * 1.5);
#pragma unroll
                    for (int r = 0; r < RB; ++r) {
                        const double x2 = xc[r] * xc[r];
                        const double sp1 = x2 + 1.0;
                        tau[r] = __builtin_fma(pref, fast_rcp1(x2 * (sp1 * sp1)), tau[r]);
                    }
                    continue;
                }
#pragma unroll
                for (int r = 0; r < RB; ++r) tau[r] += line_tau_fast(PixelX{wv[r], g[r]}, rec, etab);
            }
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int i = base + r * 64 + lane;
            if (VP_ABL(ABL_NOEXP)) { if (i < n_eval) fl[i] = 1.0 - tau[r]; continue; }
            // a chunk in the far wings of every line (|tau| < 2^-10 in all its lanes; a NaN fails the test): short Taylor form
            if (!VP_ABL(ABL_NOSMALLEXP) && !pending[r] && __ballot(!(fabs(tau[r]) < EXP_SMALL_MAX)) == 0ull) {
                if (i < n_eval) fl[i] = exp_neg_small(tau[r]);
                continue;
            }
            // voigt_model.py:217; a NaN tau (NaN pixel, poisoned line) stays NaN as in the reference
            if (i < n_eval) fl[i] = (pending[r] || tau[r] != tau[r]) ? tau[r] : exp_neg_tab(tau[r], etab);
        }
    }
    tile_sync<ONE>();
    VP_STAMP(2);

    // ---- phase B: line cores.  The flagged chunks of the tile are dealt round-robin to the
    //      waves (balanced whatever their position), each chunk finished by one wave (core_chunk).
    //      Single instance of the core code keeps the hot loop's registers low.  All control flow
    //      here is wave-uniform, so lane-held records stay readable.
    if (METHOD == 0) {
        // anything to do?  A single wave knows from its own registers (and skips the whole phase, mask
        // scan included: ~1 us of a 10 us tile without line cores); wider workgroups scan the masks
        // every wave wrote to LDS.  Uniform over the tile's threads either way.
        bool anyc = wave_any;
        if (nwaves > 1) {
            unsigned int anycore = 0u;
            for (int k = lane; k < nchunks * nwords; k += 64) {
                const unsigned long long mw = cmask[k];
                anycore |= (unsigned int)mw | (unsigned int)(mw >> 32);
            }
            anyc = __ballot(anycore != 0u) != 0ull;
        }
        if (VP_ABL(ABL_NOCORE)) anyc = false;
        if (anyc) {
            if (!daw_ready) {                        // staged only when some chunk needs the core series
                // a wave with line cores has about twice the work of one without: it goes first on its SIMD from here on
                // (walker_kernel: the workgroup's critical path; single-wave tile workgroups: longest jobs first,
                // C1 at 1024 walkers 44.6 -> 43.0 us, neutral at 8192 and on C2-C4)
                if (!VP_ABL(ABL_NOPRIO) && (ONE || nwaves == 1)) __builtin_amdgcn_s_setprio(VP_PRIO_LEVEL);
                dawson_to_lds(daw, tid, TILE_THREADS);
                if (SOLO && tid == 0) I.core_hint[p0 / I.TP] = 1;      // (walker_kernel: next time, ahead of the records)
            }
            tile_sync<ONE>();
            VP_STAMP(6);
            int kth = 0;
            int hc0 = 0, hc1 = 0, nheld = 0;         // PAIR: flagged chunks waiting for partners (scalars, not an indexed array)
            for (int c = 0; c < nchunks; ++c) {
                unsigned int any = 0u;
                for (int wd = 0; wd < nwords; ++wd) {
                    const unsigned long long mw = cmask[c * nwords + wd];
                    any |= (unsigned int)mw | (unsigned int)(mw >> 32);
                }
                if (__builtin_amdgcn_readfirstlane(any) == 0u) continue;
                if (PAIR) {
                    if (nheld + 1 == VP_CORE_ILP) {
                        int held[VP_CORE_ILP];
                        held[0] = VP_CORE_ILP == 2 ? hc0 : hc0; held[1] = VP_CORE_ILP == 2 ? c : hc1; held[VP_CORE_ILP - 1] = c;
                        core_chunk_multi<GENERIC, VP_CORE_ILP>(I, lcw, fl, cmask, nwords, q0, n_eval, held, lane, daw, etab);
                        nheld = 0;
                    } else {
                        if (nheld == 0) hc0 = c; else hc1 = c;
                        ++nheld;
                    }
                    continue;
                }
                if ((kth++ & (nwaves - 1)) != wid) continue;
                core_chunk<GENERIC>(I, lcw, fl, cmask, nwords, q0, n_eval, c, lane, daw, etab);
            }
            if (PAIR) {
                if (VP_CORE_ILP > 2 && nheld == 2) {
                    const int h2[2] = {hc0, hc1};
                    core_chunk_multi<GENERIC, 2>(I, lcw, fl, cmask, nwords, q0, n_eval, h2, lane, daw, etab);
                } else if (nheld == 1) {
                    core_chunk<GENERIC>(I, lcw, fl, cmask, nwords, q0, n_eval, hc0, lane, daw, etab);
                }
            }
        }
        tile_sync<ONE>();
    }
    VP_STAMP(3);

    // ---- LSF from LDS, chi^2 term, reduce ------------------------------------------------------
    //      Each lane produces TWO adjacent output pixels from a sliding window of the flux held in
    //      registers: per group of 8 taps it reads 10 consecutive doubles as five 16-byte LDS reads
    //      (lane stride 16 B: conflict-free) for 16 FMAs -- 5 B of LDS traffic per output and tap
    //      instead of 8 B with one output per lane.  Taps come by scalar loads (SGPR operands of the
    //      FMAs; as LDS broadcasts they were 29 % of the phase's LDS cycles: C3 with its 101 taps +12 %,
    //      C1 +1.5-2 %), zero-padded to groups of 8; the flux is followed by FL_PAD zeros (the window
    //      reads at most 9 doubles past the last one).  Per output the taps are still accumulated in ascending order, so the result is
    //      bit-identical to the plain loop.
    double acc = 0.0;
    if (VP_ABL(ABL_NOCHI) && OUT == 0) { VP_STAMP(4); return fl[tid]; }
    if (NANFIX && OUT != 2) {
        // NaN wavelength samples (a static property of the grid) make NaN model pixels; the astropy branch leaves them out of the
        // kernel-weighted mean: zero here, and the outputs are renormalised by InstDev::rbot in lsf_block6.  (A row whose theta
        // is NaN is NaN in every other pixel as well and stays NaN, as in the reference.)
        for (int i = tid; i < n_eval; i += TILE_THREADS) {
            const int q = min(max(q0 + i, 0), I.P - 1);
            const double wq = I.wave[q];
            if (wq != wq) fl[i] = 0.0;
        }
        tile_sync<ONE>();
    }
    if (OUT != 2) {
        for (int ob = 0; ob < nout; ob += LSF_PX * TILE_THREADS)
            lsf_block6<OUT, SOLO, W1, NANFIX>(I, fl, VP_ABL(ABL_NOLSF) ? 0 : Kp, ob, nout, p0, w, tid, out, out_stride, acc);
    } else {
        for (int ib = tid; ib < nout; ib += TILE_THREADS) out[(size_t)w * out_stride + p0 + ib] = fl[ib + I.halo_lo];
    }
    VP_STAMP(4);
    return acc;
}

// Publish one partial chi^2 sum of walker w (slot `myslot` of its row); the partials of a walker are then summed in
// fixed order (bit-identical either way):
//   mode 0  by a finalize_kernel launch (the kernel boundary orders everything);
//   mode 1  by the workgroup that draws the walker's last ticket: hand-off by 8-byte agent-scope atomics on both
//           sides (write-through store, drained before the ticket; L1-bypassing loads after it) --
//           placement-independent, no fences, and no wave ever waits for another one.
__device__ __forceinline__ void publish_partial(const FinalizeArgs& F, double* __restrict__ out, int out_stride, int w,
                                                int myslot, double tile_sum) {
    double* row = out + (size_t)w * out_stride;
    if (F.mode == 0) {
        row[myslot] = tile_sum;
        return;
    }
    __hip_atomic_store(row + myslot, tile_sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned int ticket = __hip_atomic_fetch_add(F.ticket + w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (ticket == (unsigned int)(F.total_tiles - 1)) {
        double total = 0.0;
        for (int k = 0; k < F.n_inst; ++k) {
            const int t0 = F.tile_off ? F.tile_off[k] : 0, t1 = F.tile_off ? F.tile_off[k + 1] : F.total_tiles;
            double sk = 0.0;
            for (int tt = t0; tt < t1; ++tt)
                sk += (tt == myslot) ? tile_sum : __hip_atomic_load(row + tt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            total += -0.5 * (sk - F.sum_logw[k]);           // vfit_mcmc.py:309-311
        }
        F.lnprob[w] = 0.0 + total;                           // lp + lnlike (vfit_mcmc.py:353)
        __hip_atomic_store(F.ticket + w, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Tile kernel: grid (W, tiles) x 64/128/256 threads, workgroup = walker x pixel tile; records come from a
// prep_lines_kernel launch.  GENERIC = false skips walkers flagged in `genflag`, GENERIC = true processes
// ONLY the flagged walkers: every (walker, tile) is handled by exactly one of the two launches.
template <int METHOD, int OUT, bool GENERIC, bool FF = false, bool NANFIX = false>   // OUT: 0 = chi^2 partial, 1 = convolved flux, 2 = unconvolved flux
                                                                // FF: far lines come from the block's expansion (farfield_kernel)
                                                                // NANFIX: InstDev::rbot (NaN wavelength samples on the astropy branch)
__global__ __launch_bounds__(TILE_THREADS_MAX, VP_TILE_WPE) void tile_kernel(InstDev I, const double* __restrict__ lc,
                                                            const int* __restrict__ flags,
                                                            double* __restrict__ out, int out_stride,
                                                            int out_offset, FinalizeArgs F,
                                                            const int* __restrict__ genflag) {
    extern __shared__ double fl[];
    // walkers are the fast grid dimension; the tiles come in the order of the geometry's table (I.core_hint + TILE_ORDER_AT:
    // those with the most line cores first, capi.hip) or, where it says 0, in grid order -- the (short) last tile last --,
    // so the tail of the launch is filled with the cheapest workgroups.  (Which workgroup evaluates a tile does not enter
    // its result.)
    const int ot = I.core_hint[TILE_ORDER_AT + blockIdx.y];
    const int t = ot > 0 ? ot - 1 : (int)blockIdx.y, w = blockIdx.x;
    const int oob = (OUT == 0) ? flags[w] : 0;   // tested below, after the first loads are in flight
    const int gen = genflag ? genflag[w] : 0;
    // (the generic instance's launch is almost always empty -- no walker left the fast domain --: its workgroups leave before
    //  they ask for their pixels)
    if (GENERIC && gen == 0) return;
    const int p0 = t * I.TP, nout = min(p0 + I.TP, I.P) - p0;
    const TilePre pre = tile_preload(I, p0, nout, threadIdx.x);
    if (oob) return;                    // out-of-bounds walker: likelihood is not evaluated
    if (GENERIC ? (gen == 0) : (gen != 0)) return;   // the other launch owns this walker
    rec_t lcw = as_rec(lc + (size_t)w * (I.L + I.NCm) * LC_STRIDE);
    if (OUT == 2 && I.line_sel == -2) {          // per-line profiles of ALL lines in one launch: grid.z = line, out (W, L, P)
        InstDev J = I;
        J.line_sel = blockIdx.z;
        tile_work<METHOD, OUT, GENERIC, false>(J, lcw, fl, p0, nout, w, threadIdx.x, blockDim.x, pre, true,
                                               out + (size_t)blockIdx.z * I.P, out_stride VP_STAMP_NONE);
        return;
    }
    const double wsum = wave_sum(tile_work<METHOD, OUT, GENERIC, false, true, false, FF, false, NANFIX>(I, lcw, fl, p0, nout, w, threadIdx.x, blockDim.x,
                                                                                         pre, true, out, out_stride VP_STAMP_NONE, false, t));
    if (OUT == 0) {
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
        double* red = fl + I.span + FL_PAD;
        if (lane == 0) red[wid] = wsum;
        __syncthreads();
        if (threadIdx.x == 0) {
            double tile_sum = red[0];
            for (int k = 1; k < nwaves; ++k) tile_sum += red[k];
            publish_partial(F, out, out_stride, w, out_offset + t, tile_sum);
        }
    }
}

// The generic instance's launch (walkers with a line outside the fast domain, flagged by prep_lines_kernel; wofz only): in most
// batches NO walker is flagged, and a grid of W x tiles workgroups that all leave at once still costs its dispatch (C1 4.8 us,
// C2 11 us per pass).  Its grid is therefore GEN_SLOTS x tiles: a workgroup walks the walkers slot, slot + GEN_SLOTS, ... and
// evaluates its tile for the flagged ones, exactly as tile_kernel<0, OUT, true> did for its one (walker, tile).
constexpr int GEN_SLOTS = 64;
template <int OUT, bool REBUILD = false, bool NANFIX = false>      // REBUILD: behind walker_kernel's flux form (whole records are formed here first)
__global__ __launch_bounds__(TILE_THREADS_MAX, VP_TILE_WPE) void tile_generic_kernel(InstDev I, const double* __restrict__ lc,
                                                            const int* __restrict__ flags,
                                                            double* __restrict__ out, int out_stride,
                                                            int out_offset, FinalizeArgs F,
                                                            const int* __restrict__ genflag, int W,
                                                            LinesDev T, const double* __restrict__ theta_rebuild, int D) {
    extern __shared__ double fl[];
    const int ot = I.core_hint[TILE_ORDER_AT + blockIdx.y];
    const int t = ot > 0 ? ot - 1 : (int)blockIdx.y;
    const int p0 = t * I.TP, nout = min(p0 + I.TP, I.P) - p0;
    // the flags of this workgroup's walkers, 64 at a time across the lanes (every wave of the workgroup for itself): ONE memory
    // round trip in the empty case instead of one per walker
    const int lane_g = threadIdx.x & 63;
    const int nmine = ((int)blockIdx.x < W) ? (W - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    for (int base = 0; base < nmine; base += 64) {
      const int idx = base + lane_g;
      const int fl_i = idx < nmine ? genflag[blockIdx.x + idx * gridDim.x] : 0;
      unsigned long long todo = __ballot(fl_i != 0);
      while (todo) {
        const int bit = __ffsll((long long)todo) - 1;
        todo &= todo - 1ull;
        const int w = blockIdx.x + (base + bit) * gridDim.x;
        if (OUT == 0 && flags[w]) continue;          // out-of-bounds walker: likelihood is not evaluated
        if (REBUILD) {
            // the walker comes from walker_kernel's flux form, whose records hold the fast domain's constants only: whole records
            // first (every workgroup of the walker's tiles writes the same bytes; each reads what it has written itself)
            double* __restrict__ rows = const_cast<double*>(lc) + (size_t)w * (I.L + I.NCm) * LC_STRIDE;
            for (int l = threadIdx.x; l < I.L; l += (int)blockDim.x)
                (void)prep_line_record(theta_rebuild + (size_t)w * D, T, l, rows + (size_t)l * LC_STRIDE);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        const TilePre pre = tile_preload(I, p0, nout, threadIdx.x);
        rec_t lcw = as_rec(lc + (size_t)w * (I.L + I.NCm) * LC_STRIDE);
        if (OUT == 2 && I.line_sel == -2) {          // per-line profiles of ALL lines in one launch: grid.z = line, out (W, L, P)
            InstDev J = I;
            J.line_sel = blockIdx.z;
            tile_work<0, OUT, true, false>(J, lcw, fl, p0, nout, w, threadIdx.x, blockDim.x, pre, true,
                                           out + (size_t)blockIdx.z * I.P, out_stride VP_STAMP_NONE);
            __syncthreads();                         // (the LDS block is the next walker's)
            continue;
        }
        const double wsum = wave_sum(tile_work<0, OUT, true, false, true, false, false, false, NANFIX>(I, lcw, fl, p0, nout, w, threadIdx.x, blockDim.x,
                                                                                        pre, true, out, out_stride VP_STAMP_NONE, false, t));
        if (OUT == 0) {
            const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
            double* red = fl + I.span + FL_PAD;
            if (lane == 0) red[wid] = wsum;
            __syncthreads();
            if (threadIdx.x == 0) {
                double tile_sum = red[0];
                for (int k = 1; k < nwaves; ++k) tile_sum += red[k];
                publish_partial(F, out, out_stride, w, out_offset + t, tile_sum);
            }
        }
        __syncthreads();
      }
    }
}

// tile_kernel for the common case, compiled on its own: single-wave tile workgroups (LSFs of up to 33 taps), chi^2 partial
// (OUT = 0), fast instance.  The general kernel carries the wave index, the workgroup size and the cross-wave barriers of its
// 2- and 4-wave forms as run-time scalars; here they are constants (tile_work<..., W1 = true>), which is worth SGPRs in a
// kernel that spills them to VGPR lanes (every spill and reload is a VALU instruction).  Same arithmetic, same order: results
// are bit-identical to tile_kernel's.
//
// Arguments that are only needed when the tile is done (where its partial sum goes, the final reduction) are NOT taken from
// the by-value parameters: kernel arguments are loaded at entry and the ~16 scalars would sit in SGPRs -- or rather be
// spilled and reloaded -- through the whole tile.  late_kernarg reads them from the kernarg segment at the point of use
// (scalar loads: no VALU slot); the empty asm keeps the compiler from moving the load up.
struct TileTail {              // what a finished tile needs (kernel argument of tile_kernel1, read late)
    double* out;               // (W, stride) partial sums
    int out_stride, out_offset;
    FinalizeArgs F;
};
struct Tile1Args {             // the kernarg segment of tile_kernel1, as one struct (offsets by offsetof; InstDev FIRST: VP_LATE_FIELD)
    InstDev I;
    const double* lc;
    const int* flags;
    const int* genflag;
    TileTail tail;
};
static_assert(offsetof(Tile1Args, I) == 0, "VP_LATE_FIELD reads InstDev at offset 0 of the kernarg segment");
template <int METHOD, bool FF>
__global__ __launch_bounds__(64, VP_TILE_WPE) void tile_kernel1(Tile1Args A) {
    extern __shared__ double fl[];
    const InstDev& I = A.I;
    const int ot = I.core_hint[TILE_ORDER_AT + blockIdx.y];
    const int t = ot > 0 ? ot - 1 : (int)blockIdx.y, w = blockIdx.x;
    const int oob = A.flags[w];
    const int gen = A.genflag ? A.genflag[w] : 0;
    const int p0 = t * I.TP, nout = min(p0 + I.TP, I.P) - p0;
    const TilePre pre = tile_preload(I, p0, nout, threadIdx.x);
    if (oob || gen != 0) return;        // out-of-bounds walker: not evaluated; flagged walker: the generic launch owns it
    rec_t lcw = as_rec(A.lc + (size_t)w * (I.L + I.NCm) * LC_STRIDE);
    const double wsum = wave_sum(tile_work<METHOD, 0, false, false, true, false, FF, true>(I, lcw, fl, p0, nout, w, threadIdx.x, 64, pre, true, nullptr,
                                                                                           0 VP_STAMP_NONE, false, t));
    if (threadIdx.x == 0) {
        const TileTail T = late_kernarg<TileTail>(offsetof(Tile1Args, tail));
        publish_partial(T.F, T.out, T.out_stride, w, T.out_offset + t, wsum);
    }
}

// Tiles of up to four instruments in ONE launch (instruments of a joint fit that share their records -- same line tables --
// each in walker_kernel's geometry: single-wave tiles whatever the LSF length): grid (W, all tiles), tile t belongs to
// instrument k = the one whose first tile tb.t[k-1] <= t.  For batches too small to fill the GPU with one instrument's
// tiles -- C3's per-GPU share of 256 walkers spent 2 x 36 us in two launches that each left most CUs idle; running them
// on two streams costs more in cross-queue waits than it saves (profiles/r03_notes.md).  Same tile work, same slots of the
// partial sums, same final reduction as the per-instrument launches (the LSFs of more than 33 taps get more halo per tile).
struct TileMulti { int t[3]; int off[4]; };     // first grid tile of instruments 1..3; first slot of each instrument's partial sums
template <int METHOD>
__global__ __launch_bounds__(64, VP_TILE_WPE) void tile_kernel_multi(InstDev I0, InstDev I1, InstDev I2, InstDev I3, TileMulti tb, int ninst,
                                                                     const double* __restrict__ lc, const int* __restrict__ flags,
                                                                     double* __restrict__ out, int out_stride, FinalizeArgs F) {
    extern __shared__ double fl[];
    const int t = blockIdx.y, w = blockIdx.x;
    const int ki = (ninst > 1 && t >= tb.t[0]) ? ((ninst > 2 && t >= tb.t[1]) ? ((ninst > 3 && t >= tb.t[2]) ? 3 : 2) : 1) : 0;
    const InstDev& I = ki == 0 ? I0 : (ki == 1 ? I1 : (ki == 2 ? I2 : I3));
    const int lt = ki == 0 ? t : t - tb.t[ki - 1];
    const int oob = flags[w];
    const int p0 = lt * I.TP, nout = min(p0 + I.TP, I.P) - p0;
    const TilePre pre = tile_preload(I, p0, nout, threadIdx.x);
    if (oob) return;                    // out-of-bounds walker: likelihood is not evaluated
    rec_t lcw = as_rec(lc + (size_t)w * (I.L + I.NCm) * LC_STRIDE);
    const double wsum = wave_sum(tile_work<METHOD, 0, false, false>(I, lcw, fl, p0, nout, w, threadIdx.x, 64, pre, true, out, out_stride VP_STAMP_NONE,
                                                                   false, lt));
    if (threadIdx.x == 0) publish_partial(F, out, out_stride, w, tb.off[ki] + lt, wsum);
}

// ---------------------------------------------------------------------------------------------
// walker kernel: the whole lnprob of one walker in ONE workgroup, one launch per batch
// ---------------------------------------------------------------------------------------------
// Records of up to four lines by the 64 lanes of a wave: lane = 16 j + slot, j = line within the
// group; slot m < 14 forms the wing coefficient K_m (its Horner polynomial in a^2 with per-lane
// coefficients -- the same operations in the same order as fill_record, so the record is bit-identical
// to prep_lines_kernel's), slots 14 and 15 store the scalars.  ~100 instructions per wave instead of
// ~350 per record lane: this sits on the critical path of the walker's workgroup.
__device__ __forceinline__ void prep_record_lanes(double thv, const LinesDev& T, int l0, double* __restrict__ lcw, int lane) {
    const int j = lane >> 4, slot = lane & 15, l = min(l0 + j, T.L - 1);
    // the wing-series constants of this lane's coefficient do not depend on theta: requested first, so that they travel
    // with the theta row instead of costing a memory round trip of their own behind it
    double wc[NWING];
#pragma unroll
    for (int i = 0; i < NWING; ++i) wc[i] = g_wing.c[min(slot, NWING - 1)][i];
    // the theta row lives across the lanes (thv = theta[lane], D <= 64) and the three parameters of the line are
    // picked with lane shuffles: the index tables are fetched beside it, one memory round trip instead of index -> theta
    const int iN = T.N_idx[l], ib = T.b_idx[l], iv = T.v_idx[l];
    const LineScalars s = line_scalars_nbv(__shfl(thv, iN, 64), __shfl(thv, ib, 64), __shfl(thv, iv, 64), T, l);
    if (l0 + j >= T.L) return;
    const bool xok = (fabs(s.Ax) <= 1.79e308) && (fabs(s.Bx) <= 1.79e308);
    const double Tl = xok ? s.Tl : __builtin_nan(""), a = s.a;
    double* __restrict__ rec = lcw + (size_t)l * LC_STRIDE;
    if (slot < NWING) {
        const double a2 = a * a;
        const double pref = Tl * (a * INV_SQRT_PI);
        double cm = 0.0;
#pragma unroll
        for (int i = NWING - 1; i >= 0; --i)           // (zero above the diagonal; not used there)
            cm = (i <= slot) ? __builtin_fma(cm, a2, wc[i]) : cm;
        rec[LC_K0 + slot] = pref * cm;
    } else if (slot == 14) {
        rec[LC_A] = s.Ax;
        rec[LC_B] = s.Bx;
        rec[LC_D] = s.d;
        rec[LC_RD] = 1.0 / s.d;               // must be the correctly rounded reciprocal (faithful_x)
        rec[LC_CFD] = s.cfd;
        rec[LC_FREQ0] = s.freq0;
        rec[LC_IBF] = s.ibf;
    } else {
        int mode = 0;
        if (!(a >= 0.0) || !(a < 7.0)) mode = 2;
        else if (a > 0.1) mode = 1;
        if (!(fabs(a) <= 1.79e308) || !(fabs(Tl) <= 1.79e308)) mode = 3;
        const double a2 = a * a;
        rec[LC_T] = Tl;
        rec[LC_Y] = a;
        rec[LC_ACOS] = 0.0;                   // (the Alg. 916 table of mode 1 is only read by the GENERIC tile kernel)
        reinterpret_cast<int*>(rec + LC_CL)[0] = T.NCm > 0 ? T.cl_mp[l] : -1;
        reinterpret_cast<int*>(rec + LC_CL)[1] = T.NCm > 0 ? T.cl_end[l] : l + 1;
        reinterpret_cast<int*>(rec + LC_MODE)[0] = mode;
        reinterpret_cast<int*>(rec + LC_MODE)[1] = core_terms(a);
        // exp(a^2) by its Taylor terms as in fill_record (mode 0: a <= 0.1); other modes never read it here
        rec[LC_EA2] = 1.0 + a2 * (1.0 + a2 * (0.5 + a2 * (1.0 / 6 + a2 * (1.0 / 24))));
    }
}

// Sampler form of walker_kernel (vp_stretch_run): the workgroup of walker k of the active half forms ITS OWN
// stretch-move proposal from the ensemble in HBM, evaluates it, and accepts or rejects it -- the whole half-step of
// the ensemble is this one launch (csrc/sampler_kernels.h has the same arithmetic as separate kernels).
struct StretchArgs {
    double* pos;           // (W, D) ensemble; rows of the active half are updated in place
    double* lp;            // (W)
    long long* nacc;       // (W) accepted proposals
    int* nanflag;
    double* chain_pos;     // (W, D) row of the chain for this step, or NULL
    double* chain_lp;      // (W)
    double a;              // stretch scale
    uint64_t seed, step;
    int s0, c0, nC, half;  // active rows [s0, s0 + gridDim.x), complementary rows [c0, c0 + nC)
    Replicas rep;          // rep.n > 0 (vp_multi_stretch_run): moved rows are written to every replica of the ensemble
                           // (this context's own among them) instead of pos / lp alone
    // Overlapped half-steps (ovl != 0, vp_stretch_run): the launches of consecutive half-steps are on two streams and run side
    // by side; a walker's workgroup waits for ITS partner alone -- ver[c0 + j] >= need, the partner's workgroup of the half-step
    // before publishes `mine` behind its row -- instead of the whole launch before it.  Everything a workgroup does before it
    // needs theta (kernel arguments, pixel loads, table staging) then runs under the previous half-step's arithmetic, and no
    // kernel boundary lies on the chain of dependent half-steps.  (ovl == 2, D <= 6: pos_x / pos_w / pos_c are arrays of 64-byte
    // "mailbox" lines -- row, lnprob at [6], version at [7] -- and lp_x / lp_w / ver are not used: one load shows a partner's version
    // AND its row.)  Rows are double-buffered by the parity of the walker's update
    // count, so that a late reader of the half-step before never meets this half-step's store: own row read from pos_x / lp_x,
    // partner from pos_c, the row of this half-step (moved or not) written to pos_w / lp_w.  `timeout`: a wait gave up.
    int ovl, need, mine;
    int* ver;              // (W)
    const double* pos_c; const double* pos_x; double* pos_w;
    const double* lp_x; double* lp_w;
    int* timeout;
};

struct WalkerArgs {
    const double* theta;   // (W, D)
    const double* lb;      // (D) box prior (vfit_mcmc.py:291-295)
    const double* ub;
    double* lc;            // (W, L + NCm, LC_STRIDE) record workspace
    double* lnprob;        // (W)
    double sum_logw;       // sum of log inv_sigma2 of the instrument
    int D;
    int wave_lds;          // LDS doubles per wave (= per tile)
    int prio;              // != 0: the waves whose tiles hold line cores run at raised issue priority
    unsigned long long wperm;   // nibble k = the tile wave k evaluates (identity: 0xFEDCBA9876543210): the host's deal of the tiles to the
                           // waves by estimated cost (capi.hip: vp_add_instrument, walker_perm_for).  Tile sums still meet in LDS by
                           // TILE index, so results do not depend on it
    // Pre-armed launch (arm_slots != NULL; capi.hip: vp_lnprob_batch): the launch is ON the GPU before its batch exists.  Its waves
    // do everything that does not need theta (kernel arguments, pixel loads, table staging) and then wait for the HOST TO PUSH the
    // batch: arm_slots is fine-grained device memory the CPU writes through the PCIe BAR, one slot of arm_slot_doubles (a
    // multiple of 8: whole 64-byte lines) per workgroup -- the walker's theta row, then, in the slot's last 8 bytes and written
    // last, (arm_seq << 2) | code (code 1 = the row is in place, 2 = leave).  Every workgroup polls ITS OWN slot in local memory:
    // no read over PCIe, no word that a thousand waves share.  The wait is bounded (arm_ticks of the 100 MHz clock, looked at by
    // wave 0 of workgroup 0): on expiry that wave tells every slot to leave and the host (arm_host[ARM_EXPIRED_WORD], pinned host
    // memory); should the host's rows arrive at that very moment some workgroups run and some leave -- the host sees the expiry
    // word, lets the launch drain and starts the batch again the ordinary way (host_wait).
    const double* arm_slots;
    unsigned int* arm_host;
    unsigned int arm_seq;
    int arm_ticks;
    int arm_slot_doubles;
    // Flux form (walker_kernel<..., FLUX = 1>, vp_model_flux_batch*): `lnprob` is the (W, flux_stride) output, there is no prior box,
    // and a walker with a line outside the fast domain is not evaluated here: its flag is set (genflag, gen_any -- what
    // prep_lines_kernel does in the tile path) and tile_generic_kernel, launched behind, takes it from the records left in the
    // workspace.  genflag_clear / gen_any_clear: the other flag buffer, cleared for the launch after this one.
    int flux_stride;
    int* genflag;
    int* genflag_clear;
    int* gen_any;
    int* gen_any_clear;
    // Split form (walker_kernel<..., SPLIT = true>; batches that leave the GPU mostly idle: at most one walker per CU): a walker is
    // `split` workgroups, each with the waves of a contiguous run of ONE-pass tiles (192 evaluated pixels: the tile launches'
    // small-batch geometry, InstDev dev_s) -- half the chain per wave, and the walker's waves on several CUs.  Every workgroup forms
    // the walker's records for itself (rows blockIdx.x of the record workspace) and applies the prior; the tile sums meet in
    // `split_part` (W, tiles of the walker), and the workgroup that draws the walker's last ticket adds ALL of them in tile order --
    // the sum does not depend on `split`, and equals the one-pass tile launches' bit for bit.  split_row0: first walker row of
    // split_part / split_ticket this launch may use (two half-steps of the stretch sampler in flight at once).
    int split;
    int split_row0;
    double* split_part;
    unsigned int* split_ticket;
};

constexpr int ARM_GO = 1, ARM_LEAVE = 2;
constexpr int ARM_EXPIRED_WORD = 16, ARM_STUCK_WORD = 32;      // (in units of 4 bytes: separate cache lines of the host block)
// (wave-uniform) the decision for this workgroup; on ARM_GO `thv` holds the theta row across the lanes (lane >= D: the last parameter)
__device__ __forceinline__ int arm_wait(const WalkerArgs& A, int w, int wid, int lane, double& thv, bool first_group, int nslots) {
    const int n = A.arm_slot_doubles;
    const unsigned long long* slot = reinterpret_cast<const unsigned long long*>(A.arm_slots) + (size_t)w * n;
    const bool keeper = first_group && wid == 0;     // the one wave that looks at the clock (wave 0 of the launch's first workgroup)
    const long long t0 = keeper ? wall_clock64() : 0ll;
    const int dl = min(lane, A.D - 1);
    for (int spins = 0;; ++spins) {
        unsigned long long flag, row;
        if (n == 8) {
            // the whole slot is one 64-byte line: ONE request brings the row and the word behind it (lane 63 asks for the word)
            row = __hip_atomic_load(slot + (lane == 63 ? 7 : dl), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            flag = (unsigned long long)(unsigned int)__builtin_amdgcn_readlane((unsigned int)row, 63) |
                   ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((unsigned int)(row >> 32), 63) << 32);
        } else {
            flag = __hip_atomic_load(slot + n - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            flag = (unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((unsigned int)flag) |
                   ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((unsigned int)(flag >> 32)) << 32);
            row = 0ull;
        }
        bool mine = ((unsigned int)flag >> 2) == A.arm_seq;
        const int code = (int)(flag & 3ull);
        if (mine && code == ARM_GO && n == 8) {
            // one line, written by the host as ONE write-combined burst with no fence between row and word: should the burst be
            // cut (an interrupt, buffer pressure) x86 does not promise that its 8-byte pieces land in address order, so the word
            // vouches for the row itself -- its upper half is a fold of the row's bits (prearm_push); a line whose row does
            // not match yet is polled again (behind the keeper's look at the clock: a wait stays bounded whatever the slot holds)
            unsigned long long h = 0ull;
            for (int k = 0; k < A.D; ++k)
                h ^= (unsigned long long)(unsigned int)__builtin_amdgcn_readlane((unsigned int)row, k) |       // (readlane returns int: no sign extension)
                     ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((unsigned int)(row >> 32), k) << 32);
            mine = (unsigned int)(h ^ (h >> 32)) == (unsigned int)(flag >> 32);
        }
        if (mine) {
            if (code == ARM_GO) {
                // (several lines: the word was written behind a store fence, the row is read again now that it has been seen)
                if (n != 8) row = __hip_atomic_load(slot + dl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                else if (A.D < 64) {
                    const unsigned int lo = __builtin_amdgcn_readlane((unsigned int)row, 62), hi = __builtin_amdgcn_readlane((unsigned int)(row >> 32), 62);
                    if (lane == 63) row = lo | ((unsigned long long)hi << 32);       // (lane 63 held the word; lane 62 holds the last parameter)
                }
                thv = __longlong_as_double((long long)row);
            }
            return code;
        }
        if (keeper) {
            if (wall_clock64() - t0 > (long long)A.arm_ticks || spins > (1 << 22)) {     // (the count: should the clock ever stand still)
                const unsigned long long leave = ((unsigned long long)A.arm_seq << 2) | (unsigned long long)ARM_LEAVE;
                unsigned long long* all = reinterpret_cast<unsigned long long*>(const_cast<double*>(A.arm_slots));
                for (int i = lane; i < nslots; i += 64) __hip_atomic_store(all + (size_t)i * n + n - 1, leave, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (lane == 0) __hip_atomic_store(A.arm_host + ARM_EXPIRED_WORD, A.arm_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                return ARM_LEAVE;
            }
        } else if (spins > SYNC_SPIN_LIMIT) {           // (workgroup 0 speaks within arm_ticks: never met; the host starts the batch again)
            if (lane == 0) __hip_atomic_store(A.arm_host + ARM_STUCK_WORD, A.arm_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return ARM_LEAVE;
        }
        __builtin_amdgcn_s_sleep(2);
    }
}

// Where a walker's lnprob goes (one thread per workgroup): the batch's output vector, and -- direct-write gather of a
// multi-rank job, R.n > 0 -- this rank's block of the gathered vector of EVERY rank (R.lp[r], peer-mapped, 8 bytes per walker
// and rank); the next launch vouches for them (replicas_handshake).
__device__ __forceinline__ void walker_result(const WalkerArgs& A, const Replicas& R, int w, double lnp) {
    if (A.lnprob) A.lnprob[w] = lnp;
    if (R.n > 0) {
        // (this rank's own vector is read by later launches on this device: a plain store; the peers' get system-scope stores)
        for (int r = 0; r < R.n; ++r) {
            if (r == R.me) R.lp[r][w] = lnp;
            else __hip_atomic_store(R.lp[r] + w, lnp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// grid = W workgroups, block = 64 x ntiles threads (ntiles <= 16): wave t owns tile t of the walker and
// runs tile_work exactly as a single-wave tile_kernel workgroup would (own LDS block, no barrier with
// its siblings), so per-tile results -- and, with the same summation order, lnprob -- are bit-identical to
// the prep + tile + finalize launches.  What the workgroup adds:
//   1. the walker's records (prep_lines_kernel's work) are formed ONCE per walker by its first waves
//      while the others already fetch their pixels; they go to the record workspace in HBM/L2 with
//      plain vector stores, every storing wave drains them (s_waitcnt vmcnt(0): the write-through L1
//      has handed them to L2), then ONE workgroup barrier, after which all waves read them with scalar
//      loads.  The scalar cache cannot hold a stale copy: it is invalidated at every kernel start and
//      nothing reads a walker's record lines before this barrier (records are 512-B aligned, scalar
//      cache lines 64 B, so no neighbour's read touches them either);
//   2. the box prior (-inf without evaluating the model) by one wave, through an LDS flag;
//   3. the final reduction: per-tile sums meet in LDS behind a second barrier and thread 0 adds them in
//      tile order -- no ticket, no atomics, no finalize launch.
// Used for single-instrument contexts whose prior box keeps every line in the fast domain, for batches
// small enough that launch overheads matter (capi.hip: enqueue_lnprob).
// NI > 1: further instruments with the same line tables (the walker's records serve all): waves tb.t[k-1] ... are the
// tiles of instrument k, Ik its geometry and spectrum, tb.slw[k-1] its weight constant; the tile sums are added per
// instrument, in order, as finalize_kernel does.
struct WalkerMore { int t[3]; double slw[3]; };
template <int METHOD, bool CLUSTERS, bool SAMPLER, int NI, bool ARMED = false, int FLUX = 0, bool SPLIT = false>   // SPLIT: WalkerArgs::split.  FLUX: the convolved model flux instead of lnprob
                                       // (WalkerArgs::flux_stride ...).  ARMED: pre-armed launch (WalkerArgs::arm_*; an instance of
                                       // its own, so that the ordinary launch's entry is compiled without the wait).  CLUSTERS: the instrument has multipole cluster records (their
                                       // preparation needs more registers than the tile work and spills to scratch;
                                       // kept out of the plain instance).  SAMPLER: stretch-move half-step (StretchArgs)
#ifndef VP_WALKER_WPE
#define VP_WALKER_WPE 6
#endif
__device__ __forceinline__ void walker_body(InstDev I0, InstDev I1, InstDev I2, InstDev I3, WalkerMore tb, LinesDev T, WalkerArgs A, StretchArgs S) {
    extern __shared__ double smem[];
    if (VP_ABL(ABL_EMPTY)) { if (threadIdx.x == 0 && A.lnprob) A.lnprob[blockIdx.x] = 0.0; return; }
    VP_STAMP_DECL
    VP_STAMP(0);
#ifdef VP_STAMPS
    const long long rt_entry = wall_clock64();      // (written behind the barrier, and only by launches that run: a pre-armed launch
    long long rt_go = rt_entry;                     //  that leaves must not overwrite the stamps of the launch before it)
#endif
#ifdef VP_STAMPS
    {   // where the wave runs: HW_ID (wave slot [3:0], SIMD [5:4], CU [11:8], SH [12], SE [15:13]) | XCC_ID << 32
        unsigned int hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        if (g_stamp_w >= 0 && g_stamp_w < STAMP_W && (threadIdx.x & 63) == 0)
            g_stamps[(g_stamp_w * STAMP_WAVES + (int)(threadIdx.x >> 6)) * STAMP_STAGES + 7] = (long long)hwid | ((long long)xcc << 32);
    }
#endif
    // SPLIT: workgroup blockIdx.x is group sg of walker w; its rows of the record workspace are its own
    const int wb = blockIdx.x;
    const int w = SPLIT ? wb / A.split : wb, sg = SPLIT ? wb - w * A.split : 0;
    const int tid = threadIdx.x, lane = tid & 63, nw = blockDim.x >> 6;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform, and the compiler knows it: the tile
                                                                   // geometry stays in SGPRs as in tile_kernel
    double* __restrict__ fl = smem + (size_t)wid * A.wave_lds;
    double* __restrict__ red = smem + (size_t)nw * A.wave_lds;      // nw tile sums, then the prior flag
    const int nrec = T.L + T.NCm;
    double* __restrict__ lcw = A.lc + (size_t)wb * nrec * LC_STRIDE;
    // the theta row of this walker across the lanes of EVERY wave (D <= 64): read from the batch, or -- sampler
    // form -- formed here: z ~ g(z) on [1/a, a], partner j from the complementary half, Y = X_j - (X_j - X_k) z
    // (stretch_propose's arithmetic; every wave repeats the few instructions instead of waiting for one)
    double thv;
    int arm_code = ARM_GO;
    double* __restrict__ stash = red + nw + 2;     // sampler form: [0] (D-1) ln z, [1] ln u, [2] lnprob of X_k, [4 + lane] X_k, [68 + lane] Y
    if (SAMPLER) {
        replicas_wait(S.rep);                    // (sharded ensemble: the complementary half as the peers left it)
        const Philox4 r = draw(S.seed, S.step, S.half, S.s0 + w, 0u);
        const double t = (S.a - 1.0) * u01(r.v[0], r.v[1]) + 1.0;
        const double z = t * t / S.a;
        int j = (int)(u01(r.v[2], r.v[3]) * (double)S.nC);
        j = j < S.nC - 1 ? j : S.nC - 1;
        const int d = min(lane, A.D - 1);
        const double* __restrict__ posX = S.ovl ? S.pos_x : S.pos;
        const double* __restrict__ lpX = S.ovl ? S.lp_x : S.lp;
        double x = 0.0, c = 0.0;
        // (overlapped half-steps: only the waves that use the proposal before the workgroup's barrier -- prior, records, the
        //  accept step's operands -- wait for the partner; the others go on to their pixels and tables)
        const bool uses_theta = !S.ovl || wid < 1 + ((T.L + 3) >> 2) + (CLUSTERS ? ((T.NCm + 63) >> 6) : 0) || wid == nw - 1;
        if (uses_theta && S.ovl == 2) {
            // mailbox lines (D <= 6): a walker's row, its lnprob and its version in ONE 64-byte line per buffer -- the line that
            // shows the partner's version also holds its row (the row was written, and acknowledged, before the version)
            x = posX[(size_t)(S.s0 + w) * 8 + d];
            const unsigned long long* line = reinterpret_cast<const unsigned long long*>(S.pos_c) + (size_t)(S.c0 + j) * 8;
            for (int spins = 0;; ++spins) {
                const unsigned long long v = __hip_atomic_load(line + (lane == 63 ? 7 : d), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((int)__builtin_amdgcn_readlane((unsigned int)v, 63) >= S.need) {
                    c = __longlong_as_double((long long)v);
                    const unsigned int lo = __builtin_amdgcn_readlane((unsigned int)v, 62), hi = __builtin_amdgcn_readlane((unsigned int)(v >> 32), 62);
                    if (lane == 63) c = __longlong_as_double((long long)(lo | ((unsigned long long)hi << 32)));
                    break;
                }
                if (spins > SYNC_SPIN_LIMIT) {
                    if (lane == 0) __hip_atomic_store(S.timeout, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
        } else if (uses_theta) {
            x = posX[(size_t)(S.s0 + w) * A.D + d];
            if (S.ovl) {
                int spins = 0;
                while (__hip_atomic_load(S.ver + S.c0 + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < S.need) {
                    if (++spins > SYNC_SPIN_LIMIT) {
                        if (lane == 0) __hip_atomic_store(S.timeout, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
                c = __hip_atomic_load(S.pos_c + (size_t)(S.c0 + j) * A.D + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                c = S.pos[(size_t)(S.c0 + j) * A.D + d];
            }
        }
        thv = c - (c - x) * z;
        // everything the accept / reject step needs besides the proposal's lnprob, by the last wave while it would wait for
        // the records anyway: at the end of the workgroup's life it was ~2.5 us of Philox blocks, logarithms and a memory round
        // trip on wave 0 alone
        if (wid == nw - 1) {
            const double lp_old = S.ovl == 2 ? posX[(size_t)(S.s0 + w) * 8 + 6] : lpX[S.s0 + w];
            const Philox4 r1 = draw(S.seed, S.step, S.half, S.s0 + w, 1u);
            const double lz = (double)(A.D - 1) * log(z), lu = log(u01(r1.v[0], r1.v[1]));
            if (lane == 0) { stash[0] = lz; stash[1] = lu; stash[2] = lp_old; }
            stash[4 + lane] = x;
            stash[68 + lane] = thv;
        }
    } else {
        if (ARMED) thv = 0.0;          // (pre-armed launch: theta comes further down, behind everything that does not need it)
        else
        thv = A.theta[(size_t)w * A.D + min(lane, A.D - 1)];
        // direct-write gather (vp_gather_*): the batch before this one has arrived here from every rank before this one
        // computes -- the dependency of an ensemble step on the whole ensemble's lnprob, as a blocking all-gather states it
        if (S.rep.n > 0) replicas_handshake(S.rep);
    }
#ifdef VP_STAMPS
    VP_STAMP(8);
    asm volatile("s_waitcnt vmcnt(0)" :: "v"(thv) : "memory");
    VP_STAMP(9);
#endif
    // (wave-uniform) which instrument this wave's tile belongs to
    // the tile (over all instruments) this wave evaluates; SPLIT: tile sg * nw + wid of the walker, a wave past the walker's
    // last tile has none (it still takes its share of the entry's tasks and of the barriers)
    const int tw_all = SPLIT ? sg * nw + wid : (int)((A.wperm >> (4 * wid)) & 15ull);
    const bool has_tile = !SPLIT || tw_all < I0.ntiles;
    const int tw = SPLIT ? min(tw_all, I0.ntiles - 1) : tw_all;
    const int ki = (NI > 1 && tw >= tb.t[0]) ? ((NI > 2 && tw >= tb.t[1]) ? ((NI > 3 && tw >= tb.t[2]) ? 3 : 2) : 1) : 0;
    const InstDev& I = ki == 0 ? I0 : (ki == 1 ? I1 : (ki == 2 ? I2 : I3));
    const int lt = ki == 0 ? tw : tw - tb.t[ki - 1];              // tile of its instrument
    const int p0 = lt * I.TP, nout = min(p0 + I.TP, I.P) - p0;
    // the tile's hint ("met line cores before": stage the Dawson table while waiting for the records) is asked for ahead of
    // the wave's pixel loads: memory operations return in order, and a wave that waits for its hint behind its pixels
    // waits for those too -- in front of the workgroup's barrier
    const int hint_v = METHOD == 0 ? I.core_hint[lt] : 0;
    const int ngrp = (T.L + 3) >> 2, ncl = CLUSTERS ? ((T.NCm + 63) >> 6) : 0;
    const int ntask = 1 + ngrp + ncl;              // task 0: box prior; then line groups; then cluster records
    auto run_tasks = [&](double th, bool rehearsal) {
        for (int task = wid; task < ntask; task += nw) {
            if (task == 0) {
                if (FLUX) {                          // (no prior box; the other flag buffer is cleared for the launch after this one)
                    if (lane == 0) {
                        red[nw] = 0.0;
                        if (A.genflag_clear) A.genflag_clear[w] = 0;
                        if (A.gen_any_clear && w == 0) *A.gen_any_clear = 0;
                    }
                } else {
                const bool oob = lane < A.D && ((th < A.lb[min(lane, A.D - 1)]) || (th > A.ub[min(lane, A.D - 1)]));
                const bool any = __ballot(oob) != 0ull;
                if (lane == 0) red[nw] = any ? 1.0 : 0.0;
                }
            } else if (task <= ngrp) {
                prep_record_lanes(th, T, (task - 1) * 4, lcw, lane);
            } else if (CLUSTERS && !rehearsal) {
                const int k = (task - 1 - ngrp) * 64 + lane;
                const double* trow = ARMED ? A.arm_slots + (size_t)w * A.arm_slot_doubles : A.theta + (size_t)w * A.D;
                if (k < T.NCm) prep_cluster(trow, T, k, lcw + (size_t)(T.L + k) * LC_STRIDE);
            }
        }
    };
    TilePre pre;
    bool daw_ready;
    if (ARMED) {
        // Pre-armed launch: what does not need theta FIRST -- the wave's exp table, the Dawson table where the tile met line cores
        // before, the pixels of the waves that form no records -- then the waves that form the records wait for the host's push.  While they wait
        // they have rehearsed: the record code has run once on the middle of the prior box (same instructions, same tables,
        // same workspace rows, overwritten by the real pass), so that the pass that counts finds its instructions, its
        // constants and its translations in the caches.
        // (the record waves ask for their pixels behind their records, as in the ordinary launch: held across the record code
        //  the preloaded registers cost it spills)
        if (lane < EXP_LDS_DOUBLES) fl[I.span + FL_PAD + 4 + DAW_LDS_DOUBLES + lane] = exp2_eighth(lane);
        daw_ready = METHOD == 0 && __builtin_amdgcn_readfirstlane(hint_v) != 0;
        if (daw_ready) dawson_to_lds(fl + I.span + FL_PAD + 4, lane, 64);
        if (wid < ntask) {
#pragma nounroll
            for (int pass = 0; pass < 2; ++pass) {
                if (pass == 0) {
                    const int dl = min(lane, A.D - 1);
                    thv = 0.5 * (A.lb[dl] + A.ub[dl]);
                } else {
                    arm_code = arm_wait(A, w, wid, lane, thv, wb == 0, SPLIT ? (int)gridDim.x / A.split : (int)gridDim.x);
#ifdef VP_STAMPS
                    rt_go = wall_clock64();
#endif
                    if (arm_code != ARM_GO) break;
                }
                run_tasks(thv, pass == 0);
            }
            // every record wave polls the slot on its own, and the slot's word can change between two polls (the keeper's
            // "leave" against the host's "go" at the moment of expiry): each wave leaves ITS decision in a word of its own and
            // the workgroup goes on only if all of them saw "go" -- decided once, behind the barrier, by everybody alike
            if (lane == 0) stash[wid] = arm_code == ARM_GO ? 1.0 : 2.0;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        VP_STAMP(10);
        pre = tile_preload(I, p0, nout, lane);
    } else {
        run_tasks(thv, false);
        VP_STAMP(10);
        // only the waves that stored records drain (s_waitcnt vmcnt(0): the write-through L1 has handed the stores to L2), and
        // before they ask for their own pixels -- the other waves' loads stay in flight across the barrier (256 walkers 16.05 ->
        // 15.9 us, 512 unchanged: there the record wave itself is the last to arrive)
        if (wid < ntask) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        pre = tile_preload(I, p0, nout, lane);     // in flight while the stores drain
        if (lane < EXP_LDS_DOUBLES) fl[I.span + FL_PAD + 4 + DAW_LDS_DOUBLES + lane] = exp2_eighth(lane);   // the wave's exp table,
                                                                               // staged while it waits for the records anyway
        // ... and the Dawson table where the tile met line cores before (1.2 us between phase A and phase B otherwise)
        daw_ready = METHOD == 0 && __builtin_amdgcn_readfirstlane(hint_v) != 0;
        if (daw_ready) dawson_to_lds(fl + I.span + FL_PAD + 4, lane, 64);
    }
    VP_STAMP(11);
    __syncthreads();
    VP_STAMP(1);
#ifdef VP_STAMPS
    if ((!ARMED || arm_code == ARM_GO) && lane == 0 && w < STAMP_W) {
        long long* gs = g_stamps + (w * STAMP_WAVES + wid) * STAMP_STAGES;
        gs[14] = rt_entry; gs[12] = rt_go; gs[13] = wall_clock64();
    }
#endif
    // the tiles with line cores are the workgroup's critical path (twice the work of the others): their waves go first
    // on their SIMDs from the start where the hint says so, from phase B on otherwise
    // (only where two workgroups share the CU -- A.prio, set by the host for batches of more than one workgroup per CU: 512 walkers
    //  24.4 us without, 21.5 with; a workgroup that has its CU to itself measured 1 % slower with it: 256 walkers 15.68 / 15.50)
    if (!VP_ABL(ABL_NOPRIO) && daw_ready && A.prio) __builtin_amdgcn_s_setprio(VP_PRIO_LEVEL);
    if (ARMED) {                                   // (told to leave, by the host or by the clock: nobody evaluates or writes anything)
        bool leave = false;
        for (int k = 0; k < min(ntask, nw); ++k) leave = leave || stash[k] != 1.0;
        if (leave) return;
    }
    const bool oobw = red[nw] != 0.0;              // out-of-bounds walker: the model is not evaluated
    if (VP_ABL(ABL_ENTRY) && !SAMPLER && !FLUX) { if (tid == 0) walker_result(A, S.rep, w, 0.0); return; }
    if (oobw && !SAMPLER) {
        if (tid == 0 && sg == 0) walker_result(A, S.rep, w, -__builtin_inf());
        return;
    }
    if (SPLIT && oobw && sg != 0) return;          // (sampler form: the walker's first group alone rejects the proposal)
    double total = 0.0;
    if (FLUX) {
        // a line outside the fast domain (the records' mode words, a lane each, every wave for itself): the generic launch's walker
        unsigned long long pr = reinterpret_cast<unsigned long long>(lcw), pq;
        asm volatile("s_mov_b64 %0, %1" : "=s"(pq) : "s"(pr) : "memory");
        const double* __restrict__ recs = reinterpret_cast<const double*>(pq);
        bool bad = false;
        for (int l = lane; l < T.L; l += 64) bad = bad || reinterpret_cast<const int*>(recs + (size_t)l * LC_STRIDE + LC_MODE)[0] != 0;
        if (__ballot(bad) != 0ull) {
            // (the records formed above carry the fast domain's constants only: tile_generic_kernel forms whole ones for the
            //  walkers it takes over -- that code would cost THIS kernel registers it has no use for otherwise)
            if (tid == 0) {
                if (A.genflag) A.genflag[w] = 1;
                if (A.gen_any) *A.gen_any = 1;
            }
            return;
        }
        if (has_tile)
            (void)tile_work<METHOD, FLUX, false, true, true, !CLUSTERS>(I, (rec_t)pq, fl, p0, nout, w, lane, 64, pre, false, A.lnprob, A.flux_stride VP_STAMP_PASS, daw_ready);
        return;
    }
    if (!oobw) {
        // the record pointer is re-made behind the barrier through an opaque scalar move, so no record load
        // can be scheduled above it
        unsigned long long pr = reinterpret_cast<unsigned long long>(lcw), pq;
        asm volatile("s_mov_b64 %0, %1" : "=s"(pq) : "s"(pr) : "memory");
        const double wsum = has_tile ? wave_sum(tile_work<METHOD, 0, false, true, true, !CLUSTERS>(I, (rec_t)pq, fl, p0, nout, w, lane, 64, pre, false, nullptr, 0 VP_STAMP_PASS, daw_ready)) : 0.0;
        if (lane == 0) red[SPLIT ? wid : tw] = wsum;
        __syncthreads();
        VP_STAMP(5);
        VP_STAMP_RT(15);
        if (wid != 0) return;
        if (SPLIT) {
            // this group's tile sums into the walker's row, then the ticket: the group that draws the last one has all of them
            // (hand-off by agent-scope atomics on both sides, as publish_partial) and adds them in TILE order -- whatever `split` is
            const int nt = I0.ntiles;
            double* __restrict__ row = A.split_part + (size_t)(A.split_row0 + w) * nt;
            const int t0 = sg * nw;
            if (lane < nw && t0 + lane < nt) __hip_atomic_store(row + t0 + lane, red[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            unsigned int ticket = 0u;
            if (lane == 0) ticket = __hip_atomic_fetch_add(A.split_ticket + A.split_row0 + w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ticket = (unsigned int)__builtin_amdgcn_readfirstlane((int)ticket);
            if (ticket != (unsigned int)(A.split - 1)) return;
            double sk = 0.0;
            for (int b = 0; b < nt; b += 64) {
                const int t = b + lane;
                const double v = t < nt ? __hip_atomic_load(row + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;     // (its own among them: drained above)
                for (int k = 0; k < min(64, nt - b); ++k)
                    sk += __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), k), __builtin_amdgcn_readlane(__double2loint(v), k));
            }
            if (lane == 0) __hip_atomic_store(A.split_ticket + A.split_row0 + w, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            total += -0.5 * (sk - A.sum_logw);         // vfit_mcmc.py:309-311
        } else {
        double sk = 0.0;
        const int n0 = NI > 1 ? tb.t[0] : nw;
        for (int k = 0; k < n0; ++k) sk += red[k];
        total += -0.5 * (sk - A.sum_logw);         // vfit_mcmc.py:309-311
        for (int j = 1; j < NI; ++j) {
            double sj = 0.0;
            const int a0 = tb.t[j - 1], a1 = j + 1 < NI ? tb.t[j] : nw;
            for (int k = a0; k < a1; ++k) sj += red[k];
            total += -0.5 * (sj - tb.slw[j - 1]);
        }
        }
    } else if (wid != 0) {
        return;
    }
    const double lnp = oobw ? -__builtin_inf() : 0.0 + total;      // lp + lnlike (vfit_mcmc.py:353)
    if (!SAMPLER) {
        if (lane == 0) walker_result(A, S.rep, w, lnp);
        return;
    }
    // ---- sampler form: accept / reject by wave 0 (stretch_accept's arithmetic), the walker's chain entry --------
    {
        const int ws = S.s0 + w;
        const double x = stash[4 + lane], y = stash[68 + lane];     // (behind the workgroup's barriers)
        const double lp_old = stash[2];
        bool accept = false;
        if (lnp != lnp) {
            if (lane == 0) atomicExch(S.nanflag, 1);
        } else {
            const double lnq = stash[0] + lnp - lp_old;
            accept = stash[1] < lnq;
        }
        if (S.ovl == 2) {
            unsigned long long* line = reinterpret_cast<unsigned long long*>(S.pos_w) + (size_t)ws * 8;
            if (lane < A.D) __hip_atomic_store(line + lane, (unsigned long long)__double_as_longlong(accept ? y : x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (lane == 0) __hip_atomic_store(line + 6, (unsigned long long)__double_as_longlong(accept ? lnp : lp_old), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) {
                __hip_atomic_store(line + 7, (unsigned long long)S.mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (accept) atomicAdd(reinterpret_cast<unsigned long long*>(S.nacc + ws), 1ull);     // (behind the version: nobody waits for it)
            }
        } else if (S.ovl) {
            // this half-step's row -- moved or not -- into the buffer the next half-steps read, then the walker's version:
            // agent-scope stores, drained before the version (the partner polls it with agent-scope loads and reads the row
            // the same way)
            if (lane < A.D) __hip_atomic_store(S.pos_w + (size_t)ws * A.D + lane, accept ? y : x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (lane == 0) {
                __hip_atomic_store(S.lp_w + ws, accept ? lnp : lp_old, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (accept) atomicAdd(reinterpret_cast<unsigned long long*>(S.nacc + ws), 1ull);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(S.ver + ws, S.mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (accept) {
            if (S.rep.n > 0) {
                for (int rr = 0; rr < S.rep.n; ++rr) {
                    if (lane < A.D) S.rep.pos[rr][(size_t)ws * A.D + lane] = y;
                    if (lane == 0) S.rep.lp[rr][ws] = lnp;
                }
                if (lane == 0) atomicAdd(reinterpret_cast<unsigned long long*>(S.nacc + ws), 1ull);     // (no wait for the old count)
            } else {
                if (lane < A.D) S.pos[(size_t)ws * A.D + lane] = y;
                if (lane == 0) { S.lp[ws] = lnp; atomicAdd(reinterpret_cast<unsigned long long*>(S.nacc + ws), 1ull); }
            }
        }
        if (S.chain_pos) {
            if (lane < A.D) S.chain_pos[(size_t)ws * A.D + lane] = accept ? y : x;
            if (lane == 0) S.chain_lp[ws] = accept ? lnp : lp_old;
        }
        if (S.rep.n > 0 && S.rep.sync && lane == 0) replicas_publish(S.rep, SPLIT ? gridDim.x / A.split : gridDim.x, SPLIT ? w : blockIdx.x);
    }
}

template <int METHOD, bool CLUSTERS, bool SAMPLER, bool ARMED = false, int FLUX = 0, bool SPLIT = false>
__global__ __launch_bounds__(WALKER_THREADS_MAX, VP_WALKER_WPE) void walker_kernel(InstDev I, LinesDev T, WalkerArgs A, StretchArgs S) {
    walker_body<METHOD, CLUSTERS, SAMPLER, 1, ARMED, FLUX, SPLIT>(I, I, I, I, WalkerMore{}, T, A, S);
}
template <int METHOD, bool SAMPLER, bool ARMED = false>
__global__ __launch_bounds__(WALKER_THREADS_MAX, VP_WALKER_WPE) void walker_kernel2(InstDev I, InstDev I1, WalkerMore tb, LinesDev T, WalkerArgs A,
                                                                                   StretchArgs S) {
    walker_body<METHOD, false, SAMPLER, 2, ARMED>(I, I1, I1, I1, tb, T, A, S);
}
template <int METHOD, bool SAMPLER, bool ARMED = false>      // three or four instruments
__global__ __launch_bounds__(WALKER_THREADS_MAX, VP_WALKER_WPE) void walker_kernel4(InstDev I, InstDev I1, InstDev I2, InstDev I3, WalkerMore tb,
                                                                                   LinesDev T, WalkerArgs A, StretchArgs S) {
    walker_body<METHOD, false, SAMPLER, 4, ARMED>(I, I1, I2, I3, tb, T, A, S);
}

// Final reduction as a launch of its own (one lane per walker), used for batches so large that the two
// L2 round trips of the fused ticket at the end of every tile wave cost more than a launch (~6 % of the
// tile kernel at 8192 walkers).  Same summation order as the fused path: bit-identical lnprob.
constexpr int FIN_MAX_INST = 8;
struct FinalizeByValue {           // the per-instrument tables in the kernel arguments: no dependent scalar loads
    int n_inst;
    int tile_off[FIN_MAX_INST + 1];
    double sum_logw[FIN_MAX_INST];
};
template <bool BYVALUE>
__global__ __launch_bounds__(64) void finalize_kernel(double* __restrict__ partial, int stride, int W,
                                                      const int* __restrict__ flags, FinalizeArgs F, FinalizeByValue V, Replicas R) {
    const int w = blockIdx.x * 64 + threadIdx.x;
    if (w >= W) return;
    double* __restrict__ row = partial + (size_t)w * stride;
    const int oob = flags[w];
    const int n_inst = BYVALUE ? V.n_inst : F.n_inst;
    double total = 0.0;
    for (int k = 0; k < n_inst; ++k) {
        const int t0 = BYVALUE ? V.tile_off[k] : F.tile_off[k], t1 = BYVALUE ? V.tile_off[k + 1] : F.tile_off[k + 1];
        double sk = 0.0;
        int tt = t0;
        for (; tt + 8 <= t1; tt += 8) {        // eight loads in flight (one 64-byte line of the walker's row), added in tile
            double v[8];                       // order: with 182 tiles (C4) the one-load-at-a-time loop was a 34 us chain
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = row[tt + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) sk += v[u];
        }
        for (; tt < t1; ++tt) sk += row[tt];
        total += -0.5 * (sk - (BYVALUE ? V.sum_logw[k] : F.sum_logw[k]));   // vfit_mcmc.py:309-311
    }
    const double lnp = oob ? -__builtin_inf() : 0.0 + total;     // lp + lnlike (vfit_mcmc.py:353)
    if (F.lnprob && !oob) F.lnprob[w] = lnp;          // (out-of-bounds walkers keep the -inf written by prep)
    if (R.n > 0) {                                    // direct-write gather: this rank's block of every rank's vector
        for (int r = 0; r < R.n; ++r) {
            if (r == R.me) R.lp[r][w] = lnp;
            else __hip_atomic_store(R.lp[r] + w, lnp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// Test hook: H(a_i, x_j) with the production tier logic (wave = 64 consecutive x_j of one a_i).
__global__ __launch_bounds__(256) void voigt_h_kernel(const double* __restrict__ lc, const double* __restrict__ x,
                                                      int nx, double* __restrict__ out) {
    rec_t rec = as_rec(lc + (size_t)blockIdx.y * LC_STRIDE);
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= nx) return;
    DirectX xp{x[j]};
    out[(size_t)blockIdx.y * nx + j] = line_tau_wofz(xp, rec);
}

}  // namespace vp
