// The scalar controller of one IVP: the per-system state of the reference's `Ida` / `IdaNLProblem` / `Newton` objects and the
// decisions taken on it. One source for both places that run it:
//   * the host stepper (ensemble_ida.cpp, g++, pow = the platform libm's), and
//   * the device-resident stepper (csrc/tiny_ida.hpp, hipcc, pow = glibc_pow::pow, which reproduces the same libm bit for bit),
// so that the two cannot drift apart. Mirrors, with the reference's names:
//   set_coeffs            /root/reference/src/lib.rs:722-782
//   nonlinear_solve       /root/reference/src/lib.rs:787-812       (lsetup decision, ss resets)
//   idaNlsConvTest        /root/reference/src/ida_nls.rs:218-266
//   test_error            /root/reference/src/lib.rs:967-1039      (decisions; the norms come from the vector code)
//   restore               /root/reference/src/lib.rs:1044-1083
//   handle_n_flag         /root/reference/src/lib.rs:1120-1244
//   complete_step         /root/reference/src/impl_complete_step.rs:22-147
//   get_solution          /root/reference/src/lib.rs:1274-1317      (coefficients)
// Every function is plain arithmetic on SysCore; -ffp-contract=off on both sides.
#pragma once
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#define IDA_HD __host__ __device__
#else
#define IDA_HD
#endif

// pow of the controller: the device side must not call the device math library (different bits)
#if defined(__HIP_DEVICE_COMPILE__)
#include "../csrc/glibc_pow.hpp"
#define IDA_POW(x, y) glibc_pow::pow((x), (y))
#define IDA_FABS(x) __builtin_fabs(x)
#define IDA_FMAX(a, b) __builtin_fmax((a), (b))
#define IDA_FMIN(a, b) __builtin_fmin((a), (b))
#else
#define IDA_POW(x, y) std::pow((x), (y))
#define IDA_FABS(x) std::fabs(x)
#define IDA_FMAX(a, b) std::fmax((a), (b))
#define IDA_FMIN(a, b) std::fmin((a), (b))
#endif

namespace idactl {

constexpr int MXORDP1 = 6;
constexpr int MAXORD_DEFAULT = 5;
constexpr long MXSTEP_DEFAULT = 500;
constexpr int MXNCF = 10, MXNEF = 10;
constexpr double EPCON = 0.33;
constexpr double XRATE = 0.25;
constexpr int MAXNLSIT = 4;
constexpr double RATEMAX = 0.9;

enum NlsCode { NLS_SUCCESS = 0, NLS_CONV_RECVR = 1, NLS_LSETUP_RECVR = 2 };
enum NFlag { NFLAG_NONE = 0, NFLAG_TEST_FAIL = 1, NFLAG_CONV_RECVR = 2, NFLAG_LSETUP_RECVR = 3 };
enum Phase { PH_IDLE = 0 /* between solve calls */, PH_LOOP_TOP = 1 /* needs the loop-top checks, then a new step */,
             PH_RETRY = 2 /* inside step()'s attempt loop */ };
// status codes of include/ida_ensemble.h that the controller itself produces
constexpr int ST_ERR_FAIL = -3, ST_CONV_FAIL = -4, ST_BAD_T = -26;

struct SysCore {
    // --- Ida scalars (src/lib.rs:89-244)
    double psi[MXORDP1] = {0}, alpha[MXORDP1] = {0}, beta[MXORDP1] = {0}, sigma[MXORDP1] = {0}, gamma[MXORDP1] = {0};
    double cvals[MXORDP1] = {0}, dvals[MAXORD_DEFAULT] = {0};
    int kk = 0, kused = 0, knew = 0, phase = 0, ns = 0;
    double hin = 0.0, h0u = 0.0, hh = 0.0, hused = 0.0, rr = 0.0;
    double tretlast = 0.0, cjlast = 0.0, eps_newt = 0.0, tolsf = 1.0;
    double tn = 0.0;
    long nst = 0, ncfn = 0, netf = 0;
    bool setup_done = false;
    // --- IdaNLProblem / IdaLProblem scalars (src/ida_nls.rs:27-59, src/ida_ls.rs:84-105)
    double cj = 0.0, cjold = 0.0, cjratio = 0.0, ss = 0.0, oldnrm = 0.0, toldel = 0.0;
    long nre = 0, nsetups = 0, nje = 0;
    long nli = 0, ncfl = 0;  // idaLsSolve's counters (ida_ls.rs:389-418): linear iterations, linear convergence failures
    // --- Newton (crates/nonlinear/src/newton.rs:14-32)
    bool jcur = false;
    int curiter = 0;
    long niters = 0, nconvfails = 0;
    // --- lock-step bookkeeping
    int ph = PH_IDLE;
    double saved_t = 0.0, ck = 0.0;
    long ncf = 0, nef = 0, nstloc = 0;
    bool call_lsetup = false;
    bool newton_retry = false;  // device lock-step stepper only: the Newton solve of the running attempt starts over (with a setup) in the next round
    int nls_ret = 0;
    double phi0nrm = 0.0;  // ||phi[0]||_wrms(ewt) for the next step's tolsf test
    bool ewt_bad = false;
    long n_attempts = 0;
    // how often this system took a path on which oracle and product follow C IDA instead of the reference's text:
    long nlufail = 0;      // Q2: the factorisation reported a zero pivot (the reference unwraps: panic)
    long nconv_jcur = 0;   // Q3/Q4: Newton gave up with a current Jacobian (reference: loops / treats it as fatal)
    long nfail_first = 0;  // Q5: a failed attempt while nst == 0 (reference's reset() rescales all of phi)
    int status = 0;
    double tret = 0.0;
    bool dead = false;  // a fatal IdaError was returned: later solve calls report it again
    double tout_cur = 0.0;  // the tout of the Ida::solve call this system is inside
    int sched_i = 0;        // index of that tout in the caller's schedule (idaens_solve: always 0)
    // --- root finding scalars (src/lib.rs:225-244); the per-function vectors live in the host stepper's Sys
    bool irfnd = false;
    double tlo = 0.0, thi = 0.0, trout = 0.0, ttol = 0.0, toutc = 0.0;
    int taskc = 0;
    long nge = 0;
};

IDA_HD inline double signum(double x) {  // f64::signum
    if (x != x) return x;
    return __builtin_signbit(x) ? -1.0 : 1.0;
}

// ---------------------------------------------------------------- set_coeffs scalars (lib.rs:722-766); returns ck
IDA_HD inline double set_coeffs(SysCore& s) {
    if (s.hh != s.hused || s.kk != s.kused) s.ns = 0;
    s.ns = (s.ns + 1 < s.kused + 2) ? s.ns + 1 : s.kused + 2;
    if (s.kk + 1 >= s.ns) {
        s.beta[0] = 1.0;
        s.alpha[0] = 1.0;
        double temp1 = s.hh;
        s.gamma[0] = 0.0;
        s.sigma[0] = 1.0;
        for (int i = 1; i <= s.kk; ++i) {
            const double scalar_i = (double)i;
            const double temp2 = s.psi[i - 1];
            s.psi[i - 1] = temp1;
            s.beta[i] = s.beta[i - 1] * s.psi[i - 1] / temp2;
            temp1 = temp2 + s.hh;
            s.alpha[i] = s.hh / temp1;
            s.sigma[i] = scalar_i * s.sigma[i - 1] * s.alpha[i];
            s.gamma[i] = s.gamma[i - 1] + s.alpha[i - 1] / s.hh;
        }
        s.psi[s.kk] = temp1;
    }
    double alphas = 0.0, alpha0 = 0.0;
    for (int i = 0; i < s.kk; ++i) {
        const double scalar_i = (double)(i + 1);
        alphas -= 1.0 / scalar_i;
        alpha0 -= s.alpha[i];
    }
    s.cjlast = s.cj;
    s.cj = -alphas / s.hh;
    double ck = IDA_FABS(s.alpha[s.kk] + alphas - alpha0);
    ck = IDA_FMAX(ck, s.alpha[s.kk]);
    return ck;  // the phi-star scaling phi[j] *= beta[j], j = ns..kk, is done with the prediction
}

// ---------------------------------------------------------------- one step attempt begins: step() prologue (lib.rs:619-653),
// set_coeffs, tn += hh, and the prologue of nonlinear_solve (lib.rs:792-812: lsetup decision, ss resets)
IDA_HD inline void begin_attempt(SysCore& s) {
    if (s.ph == PH_LOOP_TOP) {  // entering step()
        s.saved_t = s.tn;
        if (s.nst == 0) {
            s.kk = 1;
            s.kused = 0;
            s.hused = 0.0;
            s.psi[0] = s.hh;
            s.cj = 1.0 / s.hh;
            s.phase = 0;
            s.ns = 0;
        }
        s.ncf = 0;
        s.nef = 0;
        s.ph = PH_RETRY;
    }
    s.n_attempts += 1;
    s.ck = set_coeffs(s);
    s.tn += s.hh;
    s.call_lsetup = false;
    if (s.nst == 0) {
        s.cjold = s.cj;
        s.ss = 20.0;
        s.call_lsetup = true;
    }
    s.cjratio = s.cj / s.cjold;
    const double temp1 = (1.0 - XRATE) / (1.0 + XRATE);
    const double temp2 = 1.0 / temp1;
    if (s.cjratio < temp1 || s.cjratio > temp2) s.call_lsetup = true;
    if (s.cj != s.cjlast) s.ss = 100.0;
    s.nls_ret = NLS_SUCCESS;
}

// ---------------------------------------------------------------- idaNlsLSetup's bookkeeping (ida_nls.rs:168-179, ida_ls.rs:250)
IDA_HD inline void after_lsetup(SysCore& s, int info) {
    s.nsetups += 1;
    s.nje += 1;
    s.jcur = true;
    s.cjold = s.cj;
    s.cjratio = 1.0;
    s.ss = 20.0;
    s.nls_ret = info ? NLS_LSETUP_RECVR : NLS_SUCCESS;
}

// ---------------------------------------------------------------- idaLsSolve's bookkeeping around LSolver::solve (ida_ls.rs:316-418)
// ls_type: LSolverType (0 Direct, 1 Iterative, 2 MatrixIterative; include/ida_hip.h). Returns the tolerance the solver is to be
// called with: sqrt(N) * eplifac for an iterative solver, 0 for a direct one (:323-329).
constexpr double EPLIFAC = 0.05;  // ida_ls.rs:211 (pt05)
IDA_HD inline double lsolve_tol(int ls_type, double sqrt_n, double eplifac) { return ls_type == 0 ? 0.0 : sqrt_n * eplifac; }
// after the solve: nli += num_iters for an iterative solver (:389-400), ncfl += 1 when the solver failed (:413-415); returns
// whether the correction is to be scaled by 2 / (1 + cjratio) -- direct and matrix-iterative solvers only (:405-410)
IDA_HD inline bool after_lsolve(SysCore& s, int ls_type, long num_iters, bool failed) {
    if (ls_type != 0) s.nli += num_iters;
    if (failed) s.ncfl += 1;
    return (ls_type == 0 || ls_type == 2) && s.cjratio != 1.0;
}

// ---------------------------------------------------------------- idaNlsConvTest (ida_nls.rs:218-266) for iteration m = s.curiter
// returns NLS_SUCCESS (with *converged) or NLS_CONV_RECVR
IDA_HD inline int conv_test(SysCore& s, double delnrm, bool* converged) {
    const int m = s.curiter;
    *converged = false;
    int ret = NLS_SUCCESS;
    if (m == 0) {
        s.oldnrm = delnrm;
        if (delnrm <= 0.0001 * s.toldel) *converged = true;
    } else {
        const double base = delnrm / s.oldnrm;
        const double arg = 1.0 / (double)m;
        const double rate = IDA_POW(base, arg);
        if (rate > RATEMAX) ret = NLS_CONV_RECVR;
        else s.ss = rate / (1.0 - rate);
    }
    if (ret == NLS_SUCCESS && !*converged && s.ss * delnrm <= s.eps_newt) *converged = true;
    return ret;
}

// ---------------------------------------------------------------- test_error decisions (lib.rs:967-1039)
IDA_HD inline bool test_error(SysCore& s, double ck, const double* nrm /* enorm_k, enorm_km1, enorm_km2 */, double* err_k_out,
                              double* err_km1_out) {
    const double scalar_kk = (double)s.kk;
    const double enorm_k = nrm[0];
    const double err_k = s.sigma[s.kk] * enorm_k;
    const double terr_k = err_k * (scalar_kk + 1.0);
    double err_km1 = 0.0;
    int knew = s.kk;
    if (s.kk > 1) {
        const double enorm_km1 = nrm[1];
        err_km1 = s.sigma[s.kk - 1] * enorm_km1;
        const double terr_km1 = scalar_kk * err_km1;
        if (s.kk > 2) {
            const double enorm_km2 = nrm[2];
            const double err_km2 = s.sigma[s.kk - 2] * enorm_km2;
            const double terr_km2 = (scalar_kk - 1.0) * err_km2;
            if (IDA_FMAX(terr_km1, terr_km2) <= terr_k) knew = s.kk - 1;
        } else {
            if (terr_km1 <= terr_k * 0.5) knew = s.kk - 1;
        }
    }
    s.knew = knew;
    *err_k_out = err_k;
    *err_km1_out = err_km1;
    return (ck * enorm_k) <= 1.0;
}

// ---------------------------------------------------------------- restore scalars (lib.rs:1044-1083)
IDA_HD inline void restore_scalars(SysCore& s) {
    s.tn = s.saved_t;
    for (int j = 1; j < s.kk + 1; ++j) s.psi[j - 1] = s.psi[j] - s.hh;
    if (s.ns <= s.kk) {
        for (int j = s.ns; j <= s.kk; ++j) s.cvals[j - s.ns] = 1.0 / s.beta[j];
    }
}

// ---------------------------------------------------------------- handle_n_flag (lib.rs:1120-1244); 0 = predict again
IDA_HD inline int handle_n_flag(SysCore& s, int nflag, double err_k, double err_km1, long maxnef, long maxncf) {
    s.phase = 1;
    if (s.nst == 0) s.nfail_first += 1;
    if (nflag == NFLAG_LSETUP_RECVR) s.nlufail += 1;
    if (nflag == NFLAG_CONV_RECVR) s.nconv_jcur += 1;
    if (nflag == NFLAG_TEST_FAIL) {
        s.nef += 1;
        s.netf += 1;
        if (s.nef == 1) {
            const double err_knew = (s.kk == s.knew) ? err_k : err_km1;
            s.kk = s.knew;
            {
                const double base = 2.0 * err_knew + 0.0001;
                const double arg = 1.0 / (double)(s.kk + 1);
                s.rr = 0.9 * IDA_POW(base, -arg);
            }
            s.rr = IDA_FMAX(0.25, IDA_FMIN(0.9, s.rr));
            s.hh *= s.rr;
            return 0;
        } else if (s.nef == 2) {
            s.kk = s.knew;
            s.rr = 0.25;
            s.hh *= s.rr;
            return 0;
        } else if (s.nef < maxnef) {
            s.kk = 1;
            s.rr = 0.25;
            s.hh *= s.rr;
            return 0;
        }
        return ST_ERR_FAIL;
    }
    s.ncf += 1;
    s.ncfn += 1;
    s.rr = 0.25;
    s.hh *= s.rr;
    if (s.ncf < maxncf) return 0;
    return ST_CONV_FAIL;
}

// ---------------------------------------------------------------- complete_step scalars (impl_complete_step.rs:22-147)
IDA_HD inline void complete_step_scalars(SysCore& s, double err_k, double err_km1, double enorm_kp1, int maxord, double hmax_inv) {
    s.nst += 1;
    const int kdiff = s.kk - s.kused;
    s.kused = s.kk;
    s.hused = s.hh;
    if (s.knew == s.kk - 1 || s.kk == maxord) s.phase = 1;
    if (s.phase == 0) {
        if (s.nst > 1) {
            s.kk += 1;
            double hnew = 2.0 * s.hh;
            const double tmp = IDA_FABS(hnew) * hmax_inv;
            if (tmp > 1.0) hnew /= tmp;
            s.hh = hnew;
        }
    } else {
        enum { LOWER, MAINTAIN, RAISE } action;
        double err_kp1 = 0.0;
        if (s.knew == s.kk - 1) {
            action = LOWER;
        } else if (s.kk == maxord) {
            action = MAINTAIN;
        } else if (s.kk + 1 >= s.ns || kdiff == 1) {
            action = MAINTAIN;
        } else {
            const double enorm = enorm_kp1;  // ||ee - phi[kk+1]||
            err_kp1 = enorm / (double)(s.kk + 2);
            const double terr_k = (double)(s.kk + 1) * err_k;
            const double terr_kp1 = (double)(s.kk + 2) * err_kp1;
            if (s.kk == 1) {
                action = (terr_kp1 >= 0.5 * terr_k) ? MAINTAIN : RAISE;
            } else {
                const double terr_km1 = (double)s.kk * err_km1;
                if (terr_km1 <= IDA_FMIN(terr_k, terr_kp1)) action = LOWER;
                else if (terr_kp1 >= terr_k) action = MAINTAIN;
                else action = RAISE;
            }
        }
        double err_knew;
        if (action == RAISE) {
            s.kk += 1;
            err_knew = err_kp1;
        } else if (action == LOWER) {
            s.kk -= 1;
            err_knew = err_km1;
        } else {
            err_knew = err_k;
        }
        double hnew = s.hh;
        {
            const double base = 2.0 * err_knew + 0.0001;
            const double arg = -(1.0 / (double)(s.kk + 1));
            s.rr = IDA_POW(base, arg);
        }
        if (s.rr >= 2.0) {
            hnew = 2.0 * s.hh;
            const double tmp = IDA_FABS(hnew) * hmax_inv;
            if (tmp > 1.0) hnew /= tmp;
        } else if (s.rr <= 1.0) {
            s.rr = IDA_FMAX(0.5, IDA_FMIN(s.rr, 0.9));
            hnew = s.hh * s.rr;
        }
        s.hh = hnew;
    }
}

// ---------------------------------------------------------------- get_solution coefficients (lib.rs:1274-1317)
// returns 0 and fills kord/cvals/dvals, or ST_BAD_T
IDA_HD inline int get_solution_coeffs(SysCore& s, double t, int* kord_out) {
    const double eps = 2.220446049250313e-16;  // f64::EPSILON
    const double tfuzz = 100.0 * eps * (IDA_FABS(s.tn) + IDA_FABS(s.hh)) * signum(s.hh);
    const double tp = s.tn - s.hused - tfuzz;
    if ((t - tp) * s.hh < 0.0) return ST_BAD_T;
    const int kord = (s.kused == 0) ? 1 : s.kused;
    const double delt = t - s.tn;
    double c = 1.0, d = 0.0;
    double gam = delt / s.psi[0];
    s.cvals[0] = c;
    for (int j = 1; j <= kord; ++j) {
        d = d * gam + c / s.psi[j - 1];
        c = c * gam;
        gam = (delt + s.psi[j - 1]) / s.psi[j];
        s.cvals[j] = c;
        s.dvals[j - 1] = d;
    }
    *kord_out = kord;
    return 0;
}

}  // namespace idactl
