// Device-side arithmetic of the Voigt lnprob path for gfx950 (MI355X).  fp64 throughout.
//
// What is computed (reference: rbvfit core/voigt_model.py:100-159):
//     tau_lp = (N_l f_l constant_l) * H(a_l, x_lp),   H(a,x) = Re w(x + i a)
// scipy.special.wofz is replaced by a tiered evaluation chosen PER WAVEFRONT (64 consecutive
// pixels of one line), so the common case -- a line many Doppler widths away -- costs a handful
// of FMAs instead of a special-function call:
//
//   |x| >= 8, a <= 0.1 : real asymptotic series  H = (a/sqrt(pi)) s sum_m C_m(a^2) s^m,
//                        s = 1/x^2, C_m from the Gaussian moments of (t+ia)^(2m+1); the
//                        per-line coefficients are premultiplied by N f constant a/sqrt(pi)
//                        in the prep kernel, so a wing evaluation is one reciprocal + M FMAs
//                        with M in {2,3,4,6,9,14} picked from the wave's smallest |x|.
//   |x| <  8, a <= 0.1 : Taylor series of w(z) in the damping direction about the real axis,
//                          H = e^{a^2-x^2} cos(2ax) - a v_1 + a^3 v_3 - a^5 v_5 ...,
//                          v_0 = (2/sqrt(pi)) F(x), v_1 = (2/sqrt(pi)) (1 - 2xF(x)),
//                          v_{n+1} = -2 (x v_n + v_{n-1})/(n+1)     (from w' = -2zw + 2i/sqrt(pi)),
//                        with the Dawson function F and G = 1-2xF from piecewise degree-13
//                        polynomials (dawson_table.h); 1..7 odd terms depending on a.
//   0.1 < a < 7 (unphysical for UV/optical absorbers, kept for API completeness): the Gaussian-sum
//                        form of ACM TOMS Alg. 916 (Zaghloul & Ali 2011) near the axis, the
//                        Gautschi/Poppe-Wijers continued fraction elsewhere; a >= 7: continued
//                        fraction; a < 0: reflection formula.
//
// All approximations are held to <= ~1e-14 relative error in H (tests/test_faddeeva_gpu.py checks
// 1e-12 against the golden scipy grid), which bounds the flux error by 0.37 * relerr.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dawson_table.h"

namespace vp {

// ---- per-(walker,line) record written by the record preparation, read with scalar loads -------
// Records are READ through constant-address-space pointers (`rec_t`): a wave-uniform load from that
// address space is always selected as a scalar load (s_load_*, fields land in SGPRs), also in
// walker_kernel, whose own lanes wrote the records earlier in the same launch -- for a plain global
// pointer the compiler falls back to vector loads as soon as the kernel contains a store that may
// alias.  The hardware side of that hand-off (vector stores drained to L2, then scalar loads of lines
// the scalar cache cannot hold yet) is described at walker_kernel.
typedef const double __attribute__((address_space(4)))* rec_t;
typedef const int __attribute__((address_space(4)))* rec_int_t;
__device__ __forceinline__ rec_t as_rec(const double* p) { return (rec_t)reinterpret_cast<unsigned long long>(p); }
__device__ __forceinline__ int rec_int(rec_t rec, int field, int k) { return ((rec_int_t)(rec + field))[k]; }
constexpr int LC_STRIDE = 64;   // doubles per record (512 B, one record = 4 x 128-B lines)
constexpr int NWING = 14;       // longest wing series (valid for |x| >= 8)
constexpr int NCORE = 26;       // terms of the Gaussian sum (covers |x| < 7.2)
enum {
    // --- eager block: one 64-B scalar load feeds every tier with M <= 6 ---
    LC_A = 0,      // c_freq*(1+z_tot)/b_f           cheap x = fma(A, 1/wave, -B)
    LC_B = 1,      // freq0/b_f
    LC_K0 = 2,     // 14 wing coefficients  K_m = T*(a/sqrt(pi))*C_m(a^2)   (K0..K5 in the eager block)
    // --- on demand ---
    LC_D = 16,     // 1+z_tot  (as the reference rounds it)
    LC_RD = 17,    // RN(1/(1+z_tot))
    LC_CFD = 18,   // c_freq*(1+z_tot)
    LC_FREQ0 = 19, // c_freq/lambda0
    LC_IBF = 20,   // 1/b_f
    LC_T = 21,     // (N*f)*constant
    LC_Y = 22,     // a = gamma/(4 pi b_f)
    LC_EA2 = 23,   // exp(a^2)
    LC_MODE = 24,  // int[0]: 0: a<=0.1 ; 1: 0.1<a<7 ; 2: a>=7 or a<0 (generic path) ; 3: non-finite -> NaN
                   // int[1]: number of odd Taylor-in-a terms of the core series (mode 0)
    LC_CL = 25,    // int[0]: index of the multipole record if this line is the FIRST member of a cluster, else -1
                   // int[1]: one past the last line of its cluster   (static per instrument; copied into the record so
                   //         that it arrives with `mode` in one scalar load, prefetched a line ahead)
    LC_TBL0 = 26,  // 26 entries  0.5*c*a*exp(-h^2 n^2)/(h^2 n^2 + a^2)  (mode 1 only)
    LC_ACOS = 52,  // erfcx(a) - c*a*sum_n tbl_n                       (mode 1 only)
};

constexpr double C_FREQ = 2.99792458e18;          // core/voigt_model.py:130
constexpr double ATOMIC_CONSTANT = 4.48898479507e3;  // :131
constexpr double C_KMS = 299792.458;              // :197
constexpr double ALG916_H = 0.518321480430085929872;   // step of the Gaussian sum
constexpr double ALG916_C = 0.329973702884629072537;   // (2/pi) * h
constexpr double INV_SQRT_PI = 0.56418958354775628694807945156;
constexpr double X_CORE = 8.0;     // |x| below this: line-core evaluation

// a*b + c with the wave-uniform addend c taken from an SGPR pair.  For a Horner step with a uniform
// coefficient the compiler selects v_fmac_f64, whose addend must sit in the destination VGPRs, and
// pays two v_mov_b32 per step to put it there (measured: 28 v_mov for 18 FMAs in the 14-term tier);
// VOP3 v_fma_f64 takes the coefficient straight from SGPRs (scalar-loaded record field or s_mov'd
// literal), which moves that work from the VALU to the scalar unit.
__device__ __forceinline__ double fma_s(double a, double b, double c_uniform) {
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c_uniform));
    return d;
}

// 1/d to full double precision without the IEEE division sequence (no scaling/fixup needed:
// callers pass finite, normal, non-zero values; NaN/inf/0 propagate harmlessly).
__device__ __forceinline__ double fast_rcp(double d) {
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(e, r, r);
    e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(e, r, r);
    return r;
}

// 1/d to ~2e-15 relative (v_rcp_f64 delivers ~24 bits; one Newton step squares the error).  Enough
// for the wing series, whose terms only need ~1e-14.
__device__ __forceinline__ double fast_rcp1(double d) {
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    return __builtin_fma(e, r, r);
}

// exp(-t) for t >= 0 (flux = exp(-tau), Gaussian core exp(-x^2)) without the library's special-case
// selects: n = rint(-t log2 e), two-step Cody-Waite reduction, degree-13 Taylor polynomial on
// |r| <= ln2/2 (truncation 4e-18), scaling by v_ldexp (which underflows to 0 gracefully).  t is
// clamped at 800 (exp(-800) = 0 in double); NaN handling is done at the walker level (prep kernel),
// so a NaN here may come out as 0.
__device__ __forceinline__ double exp_neg(double t) {
    const double x = -fmin(t, 800.0);
    const double n = __builtin_rint(x * 1.4426950408889634074);
    double r = __builtin_fma(n, -6.93147180369123816490e-01, x);       // ln2 high part
    r = __builtin_fma(n, -1.90821492927058770002e-10, r);              // ln2 low part
    double p = 1.60590438368216145994e-10;                             // 1/13!
    p = __builtin_fma(p, r, 2.08767569878680989792e-09);
    p = __builtin_fma(p, r, 2.50521083854417187751e-08);
    p = __builtin_fma(p, r, 2.75573192239858906526e-07);
    p = __builtin_fma(p, r, 2.75573192239858906526e-06);
    p = __builtin_fma(p, r, 2.48015873015873015873e-05);
    p = __builtin_fma(p, r, 1.98412698412698412698e-04);
    p = __builtin_fma(p, r, 1.38888888888888888889e-03);
    p = __builtin_fma(p, r, 8.33333333333333333333e-03);
    p = __builtin_fma(p, r, 4.16666666666666666667e-02);
    p = __builtin_fma(p, r, 1.66666666666666666667e-01);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return __builtin_ldexp(p, (int)n);
}

// Table-driven exp(-t), t >= 0: n = rint(-t * 8/ln2), r = -t - n ln2/8 (|r| <= 0.0434), result =
// 2^(n>>3) * T[n&7] * (1 + r + ... + r^8/8!) (truncation 1.5e-18).  T = 2^(j/8) staged in LDS (per-lane gather).
// ~19 VALU instructions instead of ~45 for the polynomial version (whose 14 literal coefficients each cost extra
// v_mov's); relative error ~2e-16.  Eight entries, not 64 with a degree-5 polynomial: three more FMAs per call buy 448 B of
// LDS per tile, which (with the Dawson table below) is the difference between five and four allocation granules of
// 1280 B per single-wave tile workgroup -- 25 against 28 (register-limited) tiles per CU, and the tile launches run as fast
// as the CUs are full (profiles/r03_notes.md).
constexpr int EXP_LDS_DOUBLES = 8;
constexpr double LN2_8_HI = 8.0 * LN2_64_HI, LN2_8_LO = 8.0 * LN2_64_LO, INV_LN2_8 = INV_LN2_64 / 8.0;   // (exact scalings)
__device__ __forceinline__ double exp2_eighth(int j) { return g_exp2_64[8 * j]; }       // 2^(j/8), j < 8
__device__ __forceinline__ void exp_table_to_lds(double* __restrict__ et, int tid, int nthreads) {
    for (int j = tid; j < EXP_LDS_DOUBLES; j += nthreads) et[j] = exp2_eighth(j);
}
__device__ __forceinline__ double exp_neg_tab(double t, const double* __restrict__ et) {
    const double x = -fmin(t, 800.0);
    const double n = __builtin_rint(x * INV_LN2_8);
    double r = __builtin_fma(n, -LN2_8_HI, x);
    r = __builtin_fma(n, -LN2_8_LO, r);
    const int ni = (int)n;
    const double tj = et[ni & 7];
    // (Horner steps whose addend is a literal go through fma_s: the compiler's v_fmac form wants the addend in the destination
    //  VGPRs and pays two v_mov_b32 per step for it -- ten VALU instructions of this function's twenty-nine)
    double p = fma_s(r, 2.48015873015873015873e-05, 1.98412698412698412698e-04);   // 1/8!, 1/7!
    p = fma_s(p, r, 1.38888888888888888889e-03);           // 1/6!
    p = fma_s(p, r, 8.33333333333333333333e-03);           // 1/120
    p = fma_s(p, r, 4.16666666666666666667e-02);
    p = fma_s(p, r, 1.66666666666666666667e-01);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return __builtin_ldexp(tj * p, ni >> 3);
}

// exp(-t) for |t| < 2^-10 (most pixels of a spectrum lie in the far wings of every line): five Taylor terms, remainder
// t^6/720 < 1.3e-21.  tile_work takes it where a whole 64-pixel chunk is that shallow (wave-uniform test): 5 FMAs instead of the
// ~19 instructions of exp_neg_tab -- C1 at 512 walkers: 41 of a walker's 72 chunks.
constexpr double EXP_SMALL_MAX = 0x1p-10;
__device__ __forceinline__ double exp_neg_small(double t) {
    double p = fma_s(t, -1.0 / 120.0, 1.0 / 24.0);          // (literal addends as SGPR operands: fma_s)
    p = fma_s(p, t, -1.0 / 6.0);
    p = __builtin_fma(p, t, 0.5);
    p = __builtin_fma(p, t, -1.0);
    return __builtin_fma(p, t, 1.0);
}

// x as the reference rounds it (voigt_model.py:204,144,150) without divisions:
//   wr   = RN(wave / d)        via  q0 = wave*rd, exact residual, one correction
//   freq = RN(c_freq / wr)     via  f0 = (c_freq*d)*g  (g = RN(1/wave)), exact residual, correction
//   x    = (freq - freq0) * (1/b_f)
// Each quotient is the correctly rounded one except when the true quotient lies within ~1e-24
// (relative) of a rounding boundary.
__device__ __forceinline__ double faithful_x(double wave, double g, rec_t rec) {
    const double d = rec[LC_D], rd = rec[LC_RD], cfd = rec[LC_CFD];
    double q0 = wave * rd;
    double e = __builtin_fma(-q0, d, wave);
    double wr = __builtin_fma(e, rd, q0);
    double f0 = cfd * g;
    double y0 = d * g;                       // ~1/wr, only steers the correction
    double e2 = __builtin_fma(-f0, wr, C_FREQ);
    double freq = __builtin_fma(e2, y0, f0);
    return (freq - rec[LC_FREQ0]) * rec[LC_IBF];
}

// Wing optical depth of one line: s * Horner_M(K, s), K premultiplied by N f constant a/sqrt(pi).
template <int M>
__device__ __forceinline__ double wing_tau(double x, rec_t K) {
    const double s = fast_rcp1(x * x);
    double acc = K[M - 1];
#pragma unroll
    for (int m = M - 2; m >= 0; --m) acc = fma_s(acc, s, K[m]);
    return acc * s;
}

// sin(t), cos(t) for |t| <= 1.5 by Taylor series (error < 1e-17): the core path only ever sees
// t = |x| a <= 0.72 and 2t, so the library's full-range argument reduction is never needed.
__device__ __forceinline__ double sinc_small(double t) {   // sin(t)/t
    const double t2 = t * t;
    double p = -8.2206352466243297e-18;               // -1/19!
    p = __builtin_fma(p,t2, 2.8114572543455208e-15);   //  1/17!
    p = __builtin_fma(p,t2, -7.6471637318198165e-13);  // -1/15!
    p = __builtin_fma(p,t2, 1.6059043836821615e-10);   //  1/13!
    p = __builtin_fma(p,t2, -2.5052108385441719e-08);  // -1/11!
    p = __builtin_fma(p,t2, 2.7557319223985891e-06);   //  1/9!
    p = __builtin_fma(p,t2, -1.9841269841269841e-04);  // -1/7!
    p = __builtin_fma(p,t2, 8.3333333333333333e-03);   //  1/5!
    p = __builtin_fma(p,t2, -1.6666666666666667e-01);  // -1/3!
    return __builtin_fma(p, t2, 1.0);
}
__device__ __forceinline__ double cos_small(double t) {
    const double t2 = t * t;
    double p = fma_s(t2, 8.8967791632530450e-22, -4.1103176233121649e-19);   //  1/22!, -1/20!
    p = fma_s(p, t2, 1.5619206968586226e-16);          //  1/18!
    p = fma_s(p, t2, -4.7794773323873853e-14);         // -1/16!
    p = fma_s(p, t2, 1.1470745597729725e-11);          //  1/14!
    p = fma_s(p, t2, -2.0876756987868099e-09);         // -1/12!
    p = fma_s(p, t2, 2.7557319223985891e-07);          //  1/10!
    p = fma_s(p, t2, -2.4801587301587302e-05);         // -1/8!
    p = fma_s(p, t2, 1.3888888888888889e-03);          //  1/6!
    p = fma_s(p, t2, -4.1666666666666667e-02);         // -1/4!
    p = __builtin_fma(p, t2, 0.5);
    return __builtin_fma(-p, t2, 1.0);
}

// cos(t) for |t| <= 0.1: five terms (t^10/10! < 3e-18)
__device__ __forceinline__ double cos_tiny(double t) {
    const double t2 = t * t;
    double p = fma_s(t2, 2.4801587301587302e-05, -1.3888888888888889e-03);   //  1/8!, -1/6!
    p = fma_s(p, t2, 4.1666666666666667e-02);           //  1/4!
    p = __builtin_fma(p, t2, -0.5);
    return __builtin_fma(p, t2, 1.0);
}

__device__ __forceinline__ double sinc_safe(double t, double sint) {
    return fabs(t) < 1e-4 ? 1.0 - 0.1666666666666666666667 * t * t : sint / t;
}

// Number of odd terms (v_1, v_3, ...) of the core series needed for <= ~1e-15 relative error at
// |x| < 8 (validated against 50-digit mpmath at the band edges, scripts/gen_dawson_table.py).
__host__ __device__ __forceinline__ int core_terms(double a) {
    return a <= 1e-9 ? 1 : a <= 5e-4 ? 2 : a <= 5e-3 ? 3 : a <= 2e-2 ? 4 : a <= 5e-2 ? 5 : 7;
}

constexpr int DAW_GLO = 10;       // G = 1 - 2xF has its own polynomial from this interval (|x| >= 5) on

// Core H(a,x) for |x| < 8, 0 <= a <= 0.1 (see the header comment).  `nodd` and `ea2` = exp(a^2)
// are per-line constants.  Branch-free per lane; the only control flow is wave-uniform.
__device__ __forceinline__ double core_taylor_H(double x, double a, double ea2, int nodd) {
    const double ax = fabs(x);
    const int i = min((int)(ax * 2.0), DAW_NI - 1);
    const double t = __builtin_fma(ax, 4.0, -(double)(2 * i + 1));
    const double* __restrict__ cf = &g_dawson[i][0][0];
    double F = cf[DAW_DEG], G = cf[DAW_DEG + 1 + DAW_DEG];
#pragma unroll
    for (int k = DAW_DEG - 1; k >= 0; --k) {
        F = __builtin_fma(F, t, cf[k]);
        G = __builtin_fma(G, t, cf[DAW_DEG + 1 + k]);
    }
    G = (i >= DAW_GLO) ? G : __builtin_fma(-2.0 * ax, F, 1.0);   // (the LDS version below forms G from F everywhere)
    const double c = 1.1283791670955125739;          // 2/sqrt(pi)
    double vp = c * F, vc = c * G;                    // v_0, v_1
    const double E = exp_neg(ax * ax);
    const double a2 = a * a;
    double apow = -a;                                 // (-1)^k a^(2k-1) with alternating sign folded in
    double acc = apow * vc;
    // -2/(n+1), -2/(n+2) for n = 1, 3, 5, ...
    constexpr double R[12] = {-1.0, -2.0 / 3, -0.5, -0.4, -2.0 / 6, -2.0 / 7, -0.25, -2.0 / 9, -0.2, -2.0 / 11,
                              -2.0 / 12, -2.0 / 13};
#pragma unroll
    for (int k = 1; k < 7; ++k) {
        if (k < nodd) {                               // wave-uniform
            const double v1 = R[2 * k - 2] * __builtin_fma(ax, vc, vp);
            const double v2 = R[2 * k - 1] * __builtin_fma(ax, v1, vc);
            vp = v1; vc = v2;
            apow = -apow * a2;
            acc = __builtin_fma(apow, vc, acc);
        }
    }
    return __builtin_fma(E * ea2, cos_small(2.0 * a * ax), acc);
}

// Same series with the Dawson table staged in LDS; used by the tile kernel's hot loop.
// LDS copy: F for all 16 intervals; G = 1 - 2xF is formed from it.  Up to |x| = 5 its absolute error stays ~1.5e-16, which is
// what enters H (through a*(2/sqrt(pi))*G, next to a Gaussian term >= 1.4e-11): relative effect on H below 8e-15.  Beyond,
// where H is the a-term alone, the subtraction costs up to 2x^2 = 128 ulp of G: 1.4e-14 relative on H at |x| = 8, against a
// contract of 1e-12 (the global-memory form above keeps G's own polynomial there; an LDS copy of it was 84 doubles of every
// tile workgroup -- see exp_neg_tab on what LDS per tile costs).
constexpr int DAW_F_DOUBLES = DAW_NI * (DAW_DEG + 1);
constexpr int DAW_LDS_DOUBLES = DAW_F_DOUBLES;
__device__ __forceinline__ void dawson_to_lds(double* __restrict__ daw, int tid, int nthreads) {
    for (int idx = tid; idx < DAW_LDS_DOUBLES; idx += nthreads) daw[idx] = g_dawson[idx / (DAW_DEG + 1)][0][idx % (DAW_DEG + 1)];
}
// N independent pixels per lane for ONE line (a, ea2, nodd wave-uniform): the stages of the N evaluations sit side
// by side in one basic block, so that a wave that is alone on its SIMD overlaps their dependent chains.  Every
// pixel sees exactly the operations of the N = 1 form.
template <int N>
__device__ __forceinline__ void core_taylor_H_lds_n(const double (&x)[N], double a, double ea2, int nodd,
                                                    const double* __restrict__ daw, const double* __restrict__ et,
                                                    double (&H)[N]) {
    double ax[N], t[N], F[N], G[N];
    int i[N];
#pragma unroll
    for (int r = 0; r < N; ++r) {
        ax[r] = fabs(x[r]);
        i[r] = min((int)(ax[r] * 2.0), DAW_NI - 1);
        t[r] = __builtin_fma(ax[r], 4.0, -(double)(2 * i[r] + 1));
    }
    {
        const double* __restrict__ cF[N];
#pragma unroll
        for (int r = 0; r < N; ++r) { cF[r] = daw + i[r] * (DAW_DEG + 1); F[r] = cF[r][DAW_DEG]; }
#pragma unroll
        for (int k = DAW_DEG - 1; k >= 0; --k) {
#pragma unroll
            for (int r = 0; r < N; ++r) F[r] = __builtin_fma(F[r], t[r], cF[r][k]);
        }
    }
#pragma unroll
    for (int r = 0; r < N; ++r) G[r] = __builtin_fma(-2.0 * ax[r], F[r], 1.0);
    const double c = 1.1283791670955125739;          // 2/sqrt(pi)
    double vp[N], vc[N], E[N], acc[N];
    const double a2 = a * a;
    double apow = -a;
#pragma unroll
    for (int r = 0; r < N; ++r) {
        vp[r] = c * F[r]; vc[r] = c * G[r];           // v_0, v_1
        E[r] = exp_neg_tab(ax[r] * ax[r], et);
        acc[r] = apow * vc[r];
    }
    constexpr double R[12] = {-1.0, -2.0 / 3, -0.5, -0.4, -2.0 / 6, -2.0 / 7, -0.25, -2.0 / 9, -0.2, -2.0 / 11,
                              -2.0 / 12, -2.0 / 13};
#pragma unroll
    for (int k = 1; k < 7; ++k) {
        if (k < nodd) {                               // wave-uniform
            apow = -apow * a2;
#pragma unroll
            for (int r = 0; r < N; ++r) {
                const double v1 = R[2 * k - 2] * __builtin_fma(ax[r], vc[r], vp[r]);
                const double v2 = R[2 * k - 1] * __builtin_fma(ax[r], v1, vc[r]);
                vp[r] = v1; vc[r] = v2;
                acc[r] = __builtin_fma(apow, vc[r], acc[r]);
            }
        }
    }
    // 2 a |x| <= 16 a: a <= 5e-3 (nodd <= 3) keeps the argument below 0.08
    if (nodd <= 3) {
#pragma unroll
        for (int r = 0; r < N; ++r) H[r] = __builtin_fma(E[r] * ea2, cos_tiny(2.0 * a * ax[r]), acc[r]);
    } else {
#pragma unroll
        for (int r = 0; r < N; ++r) H[r] = __builtin_fma(E[r] * ea2, cos_small(2.0 * a * ax[r]), acc[r]);
    }
}
__device__ __forceinline__ double core_taylor_H_lds(double x, double a, double ea2, int nodd,
                                                    const double* __restrict__ daw,
                                                    const double* __restrict__ et) {
    const double xs[1] = {x};
    double H[1];
    core_taylor_H_lds_n<1>(xs, a, ea2, nodd, daw, et, H);
    return H[0];
}

// Core H(a,x), |x| < 7.2, 0 <= a < 7:  Alg. 916 real part
//   H = E [ (erfcx(a) - c a S1) cos(2xa) + c x sin(xa) sinc(xa) ] + E * sum_n tblc_n (e^{2hnx} + e^{-2hnx})
// with E = exp(-x^2), tblc_n = 0.5 c a exp(-h^2n^2)/(h^2n^2+a^2).
__device__ __forceinline__ double core_H(double x, rec_t rec) {
    const double ax = fabs(x);
    const double y = rec[LC_Y];
    const double E = exp(-ax * ax);
    const double e2 = exp((2.0 * ALG916_H) * ax);
    const double em2 = fast_rcp(e2);
    double p = 1.0, q = 1.0, s = 0.0;
#pragma unroll 2
    for (int n = 0; n < NCORE; ++n) {
        p *= e2;
        q *= em2;
        s = __builtin_fma(rec[LC_TBL0 + n], p + q, s);
    }
    const double t = ax * y;
    double sn, sc, cs;
    if (y <= 0.1) { sc = sinc_small(t); sn = sc * t; cs = cos_small(2.0 * t); }   // uniform per line
    else { sn = sin(t); sc = sinc_safe(t, sn); cs = cos(2.0 * t); }
    const double head = __builtin_fma(rec[LC_ACOS], cs, (ALG916_C * ax) * sn * sc);
    return E * (head + s);
}

// Generic Re w(x+iy) by the Gautschi / Poppe-Wijers continued fraction (large |z|), any y >= 0.
// Term count follows the fit published with the MIT Faddeeva package (nu = 3.9 + 11.398/(0.08254 x
// + 0.1421 y + 0.2023)), evaluated bottom-up in complex arithmetic.
__device__ inline double cf_rew(double x, double y) {
    const double ax = fabs(x);
    if (ax + y > 1e7) {   // w ~ i/(sqrt(pi) z), scaled against overflow
        if (ax > y) { double yax = y / ax; return INV_SQRT_PI / (ax + yax * y) * yax; }
        double xya = ax / y; return INV_SQRT_PI / (xya * ax + y);
    }
    double nu = floor(3.9 + 11.398 / (0.08254 * ax + 0.1421 * y + 0.2023));
    double wr = ax, wi = y;
    for (nu = 0.5 * (nu - 1.0); nu > 0.4; nu -= 0.5) {
        double denom = nu / (wr * wr + wi * wi);
        wr = ax - wr * denom;
        wi = y + wi * denom;
    }
    return INV_SQRT_PI / (wr * wr + wi * wi) * wi;
}

// Fully generic H for lines outside the fast domain (mode != 0): per-element branches, slow, rare.
__device__ inline double generic_H(double x, rec_t rec, int mode) {
    const double y = rec[LC_Y];
    const double ax = fabs(x);
    if (mode == 1) {
        if (ax < X_CORE) return core_H(x, rec);
        return cf_rew(ax, y);
    }
    if (mode == 3) return __builtin_nan("");       // non-finite line constants
    // mode 2: a >= 7 (continued fraction everywhere) or a < 0 (reflection)
    if (y >= 7.0) return cf_rew(ax, y);
    if (y < 0.0) {
        // w(z) for Im z < 0:  w(z) = 2 exp(-z^2) - w(-z)  =>  Re = 2 e^{y^2-x^2} cos(2xy) - H(|y|,x)
        const double ya = -y;
        double h;
        if (ya >= 7.0 || ax >= X_CORE) h = cf_rew(ax, ya);
        else {
            // small table-free evaluation of the Gaussian sum for |y| (rare path)
            const double E = exp(-ax * ax);
            double s1 = 0.0, s23 = 0.0;
            for (int n = 1; n <= NCORE; ++n) {
                double hn = ALG916_H * n;
                double tb = exp(-hn * hn) / (hn * hn + ya * ya);
                s1 += tb;
                s23 += tb * (exp(2.0 * hn * ax) + exp(-2.0 * hn * ax));
            }
            const double t = ax * ya;
            const double sn = sin(t);
            h = E * ((erfcx(ya) - ALG916_C * ya * s1) * cos(2.0 * t) + (ALG916_C * ax) * sn * sinc_safe(t, sn)
                     + 0.5 * ALG916_C * ya * s23);
        }
        return 2.0 * exp(ya * ya - ax * ax) * cos(2.0 * ax * ya) - h;
    }
    return __builtin_nan("");   // y is NaN
}

// 'fast' method: Tepper-Garcia form exactly as core/voigt_approx.py:69-86 (bug-compatible wings).
__device__ __forceinline__ double tepper_garcia_H(double x, double a, const double* __restrict__ et) {
    const double SQRT_PI = 1.7724538509055160273;   // numpy sqrt(pi)
    const double x2 = x * x;
    const double G = exp_neg_tab(x2, et);           // table-driven exp (2e-16), 0 beyond x^2 = 800 like exp()
    const double asp = a * 0.56418958354775628695;  // a / sqrt(pi) to 1 ulp
    const double eps = fmax(1e-2, 100.0 * fabs(a) / SQRT_PI);
    const double safe = fmax(x2, eps);
    const double numer = G * (4.0 * (safe * safe) + 7.0 * safe + 4.0) - 1.5;
    const double sp1 = safe + 1.0;
    const double denom = safe * (sp1 * sp1);
    const double H_tg = G - asp * numer * fast_rcp(denom);
    const double H_core = G * (1.0 - 2.0 * asp);
    const double h = (x2 < eps) ? H_core : H_tg;
    return (x != x) ? x : h;                        // exp_neg_tab's clamp would swallow a NaN
}

}  // namespace vp
