/*
 * voigt_oracle.c -- CPU ORACLE (plain C restatement) of the rbvfit lnprob hot path.
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  Parity status: PINNED by tests/test_c_oracle.py against the
 * golden vectors generated from the real reference (tests/golden/make_golden.py).
 *
 * Follows, one theta row at a time:
 *   _evaluate_compiled_model      src/rbvfit/core/voigt_model.py:162-261
 *   _vectorized_voigt_tau         src/rbvfit/core/voigt_model.py:100-159
 *   H_tepper_garcia               src/rbvfit/core/voigt_approx.py:69-86
 *   vfit.lnprior/lnlike/lnprob    src/rbvfit/vfit_mcmc.py:291-353
 * The Faddeeva function the reference obtains from scipy.special.wofz (un-vendored dependency,
 * scipy>=1.5, setup.cfg:32) is restated from its published algorithm: S. G. Johnson's Faddeeva
 * package = ACM TOMS Algorithm 916 (Zaghloul & Ali 2011) near the real axis, the Gautschi /
 * Poppe-Wijers continued fraction for large |z|.  Only Re w(z) is needed (voigt_model.py:156).
 * It deliberately shares no code with the HIP path (which uses a Taylor series in the damping
 * parameter + asymptotic wings), so the two check each other.
 *
 * Build: make -C oracle   (gcc -O2 -fopenmp -shared)
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define C_FREQ 2.99792458e18          /* voigt_model.py:130 */
#define ATOMIC_CONSTANT 4.48898479507e3   /* :131 */
#define C_KMS 299792.458              /* :197 */
#define ISPI 0.56418958354775628694807945156   /* 1/sqrt(pi) */
#define A916 0.518321480430085929872  /* step of the Gaussian sum for relerr = DBL_EPSILON */
#define C916 0.329973702884629072537  /* (2/pi) * A916 */

static double sinc_s(double x, double sinx) { return fabs(x) < 1e-4 ? 1 - 0.1666666666666666666667 * x * x : sinx / x; }

/* Re w(x + i y), y >= 0 */
static double rew_pos(double x, double ya)
{
    x = fabs(x);
    if (ya > 7 || (x > 6 && (ya > 0.1 || (x > 8 && ya > 1e-10) || x > 28))) {
        /* continued fraction; term count from the fit published with the package */
        if (x + ya > 4000) {
            if (x + ya > 1e7) {
                if (x > ya) { double yax = ya / x; return ISPI / (x + yax * ya) * yax; }
                if (isinf(ya)) return (isnan(x)) ? x : 0.0;
                { double xya = x / ya; return ISPI / (xya * x + ya); }
            } else {
                double dr = x * x - ya * ya - 0.5, di = 2 * x * ya;
                double denom = ISPI / (dr * dr + di * di);
                return denom * (x * di - ya * dr);
            }
        } else {
            double nu = floor(3.9 + 11.398 / (0.08254 * x + 0.1421 * ya + 0.2023));
            double wr = x, wi = ya;
            for (nu = 0.5 * (nu - 1); nu > 0.4; nu -= 0.5) {
                double denom = nu / (wr * wr + wi * wi);
                wr = x - wr * denom;
                wi = ya + wi * denom;
            }
            return ISPI / (wr * wr + wi * wi) * wi;
        }
    }
    {
        const double a = A916, c = C916, a2 = A916 * A916;
        double sum1 = 0, sum2 = 0, sum3 = 0, expx2;
        if (x < 10) {
            double prod2ax = 1, prodm2ax = 1;
            expx2 = exp(-x * x);
            {
                const double exp2ax = exp((2 * a) * x), expm2ax = 1 / exp2ax;
                int n;
                for (n = 1; n < 200; ++n) {
                    const double coef = exp(-a2 * (double)(n * n)) * expx2 / (a2 * (double)(n * n) + ya * ya);
                    prod2ax *= exp2ax;
                    prodm2ax *= expm2ax;
                    sum1 += coef;
                    sum2 += coef * prodm2ax;
                    sum3 += coef * prod2ax;
                    /* the package tests the slowest-decaying (imaginary) sum; same quantity here */
                    if ((coef * prod2ax) * (a * n) < 2.2204460492503131e-16 * (sum3 * a * n + 1e-300) && n > x / a + 1) break;
                }
            }
        } else {   /* x >= 10 and ya <= 1e-10: only sum3 matters; sum around n0 = x/a */
            double n0 = floor(x / a + 0.5), dx = a * n0 - x;
            double exp1 = exp(4 * a * dx), exp1dn = 1, tm, tp;
            int dn;
            expx2 = exp(-x * x);
            sum3 = exp(-dx * dx) / (a2 * (n0 * n0) + ya * ya);
            for (dn = 1; n0 - dn > 0; ++dn) {
                double np = n0 + dn, nm = n0 - dn;
                tp = exp(-(a * dn + dx) * (a * dn + dx));
                tm = exp(-(a * dn - dx) * (a * dn - dx));
                (void)exp1; (void)exp1dn;
                tp /= (a2 * (np * np) + ya * ya);
                tm /= (a2 * (nm * nm) + ya * ya);
                sum3 += tp + tm;
                if (tp + tm < 2.2204460492503131e-16 * sum3) break;
            }
            /* sum3 here already contains exp(-(an-x)^2) (no expx2 factor) */
            {
                const double sinxy = sin(x * ya);
                const double head = expx2 * (exp(ya * ya) * erfc(ya)) * cos(2 * x * ya) + (c * x * expx2) * sinxy * sinc_s(x * ya, sinxy);
                return head + (0.5 * c) * ya * sum3;
            }
        }
        {
            const double erfcx_y = exp(ya * ya) * erfc(ya);
            const double sinxy = sin(x * ya);
            const double coef1 = expx2 * erfcx_y - c * ya * sum1;
            const double coef2 = c * x * expx2;
            return coef1 * cos(2 * x * ya) + coef2 * sinxy * sinc_s(x * ya, sinxy) + (0.5 * c) * ya * (sum2 + sum3);
        }
    }
}

double vo_rew(double x, double y)
{
    if (isnan(x) || isnan(y)) return NAN;
    if (y >= 0) return rew_pos(x, y);
    /* w(z) = 2 exp(-z^2) - w(-z)  =>  Re = 2 exp(y^2-x^2) cos(2xy) - Re w(x + i|y|) */
    return 2 * exp(y * y - x * x) * cos(2 * x * y) - rew_pos(x, -y);
}

static double h_tepper_garcia(double x, double a)
{   /* voigt_approx.py:69-86 */
    const double sqrt_pi = sqrt(M_PI);
    double x2 = x * x, G = exp(-x2);
    double eps = fmax(1e-2, 100.0 * fabs(a) / sqrt_pi);
    double safe = fmax(x2, eps);
    double numer = G * (4.0 * (safe * safe) + 7.0 * safe + 4.0) - 1.5;
    double denom = safe * ((safe + 1.0) * (safe + 1.0));
    double H_tg = G - (a / sqrt_pi) * numer / denom;
    double H_core = G * (1.0 - 2.0 * a / sqrt_pi);
    return x2 < eps ? H_core : H_tg;
}

typedef struct {
    int P, L, K, lsf_mode, method;            /* lsf_mode: 0 none, 1 scipy nearest, 2 astropy extend */
    const double *wave, *flux, *inv_sigma2, *log_inv_sigma2;
    const double *lambda0, *gamma, *f, *zfac;
    const int *N_idx, *b_idx, *v_idx;
    const double *taps;
} vo_inst;

/* model flux of one theta row; work must hold 2*P doubles */
void vo_model_flux(const vo_inst* I, const double* theta, double* out, double* work, int convolved)
{
    const int P = I->P, L = I->L;
    double* tau = work;
    double* fl = work + P;
    int p, l, j;
    memset(tau, 0, sizeof(double) * P);
    for (l = 0; l < L; ++l) {
        const double N = pow(10.0, theta[I->N_idx[l]]);                /* voigt_model.py:192 */
        const double b = theta[I->b_idx[l]], v = theta[I->v_idx[l]];
        const double z_total = I->zfac[l] * (1 + v / C_KMS) - 1;      /* :200 */
        const double d = 1 + z_total;
        const double lam0 = I->lambda0[l];
        const double b_f = b / lam0 * 1e13;                            /* :142 */
        const double freq0 = C_FREQ / lam0;                            /* :143 */
        const double constant = ATOMIC_CONSTANT / (freq0 * b);         /* :146 */
        const double a = I->gamma[l] / (4 * M_PI * b_f);               /* :149 */
        const double T = N * I->f[l] * constant;                       /* :158 */
        for (p = 0; p < P; ++p) {
            const double wave_rest = I->wave[p] / d;                   /* :204 */
            const double freq = C_FREQ / wave_rest;                    /* :144 */
            const double x = (freq - freq0) / b_f;                     /* :150 */
            const double H = I->method == 1 ? h_tepper_garcia(x, a) : vo_rew(x, a);
            tau[p] += T * H;
        }
    }
    for (p = 0; p < P; ++p) fl[p] = exp(-tau[p]);                      /* :217 */
    if (!convolved || I->lsf_mode == 0 || I->K <= 0) { memcpy(out, fl, sizeof(double) * P); return; }
    {   /* true convolution, edge value replicated; astropy branch normalises the taps (:220-230) */
        const int K = I->K, c = K / 2;
        double norm = 1.0;
        int any_nan = 0;
        if (I->lsf_mode == 2) { norm = 0; for (j = 0; j < K; ++j) norm += I->taps[j]; }
        if (I->lsf_mode == 2) for (p = 0; p < P; ++p) any_nan |= fl[p] != fl[p];
        if (any_nan) {
            /* astropy convolve, nan_treatment='interpolate' (its default): top / bot over the window's non-NaN samples; a window
               of NaNs only keeps the input value (tests/golden/nan_semantics.npz, nan_wave_custom.npz) */
            for (p = 0; p < P; ++p) {
                double top = 0, bot = 0;
                for (j = K - 1; j >= 0; --j) {            /* ascending sample index, as the C loop of _convolveNd_c walks it */
                    int q = p + c - j;
                    q = q < 0 ? 0 : (q >= P ? P - 1 : q);
                    if (fl[q] == fl[q]) { top += fl[q] * I->taps[j]; bot += I->taps[j]; }
                }
                out[p] = bot == 0 ? fl[p] : top / bot;
            }
            return;
        }
        for (p = 0; p < P; ++p) {
            double s = 0;
            for (j = 0; j < K; ++j) {
                int q = p + c - j;
                q = q < 0 ? 0 : (q >= P ? P - 1 : q);
                s += (I->taps[j] / norm) * fl[q];
            }
            out[p] = s;
        }
    }
}

double vo_lnprob(const vo_inst* insts, int n_inst, int D, const double* theta, const double* lb, const double* ub,
                 double* work)
{
    int d, k, p;
    double total = 0;
    for (d = 0; d < D; ++d) if (theta[d] < lb[d] || theta[d] > ub[d]) return -INFINITY;   /* vfit_mcmc.py:293 */
    for (k = 0; k < n_inst; ++k) {
        const vo_inst* I = &insts[k];
        double* model = work;                 /* P */
        double s = 0;
        vo_model_flux(I, theta, model, work + I->P, 1);
        for (p = 0; p < I->P; ++p) {
            const double r = I->flux[p] - model[p];
            s += r * r * I->inv_sigma2[p] - I->log_inv_sigma2[p];          /* :309-311 */
        }
        total += -0.5 * s;
    }
    return 0.0 + total;
}

/* batch over walker rows; nthreads <= 1 is the serial map emcee does with pool=None */
void vo_lnprob_batch(const vo_inst* insts, int n_inst, int W, int D, const double* thetas, const double* lb,
                     const double* ub, double* out, int nthreads)
{
    int maxP = 0, k, w;
    for (k = 0; k < n_inst; ++k) if (insts[k].P > maxP) maxP = insts[k].P;
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads > 0 ? nthreads : 1)
#endif
    {
        double* work = (double*)malloc(sizeof(double) * 3 * (size_t)maxP);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (w = 0; w < W; ++w) out[w] = vo_lnprob(insts, n_inst, D, thetas + (size_t)w * D, lb, ub, work);
        free(work);
    }
}

void vo_rew_array(int n, const double* x, const double* y, double* out)
{
    int i;
    for (i = 0; i < n; ++i) out[i] = vo_rew(x[i], y[i]);
}
