// ORACLE (test infrastructure, never shipped / never on the product path).
//
// CPU restatement of the reference's IDA stepper and its Newton/linear-solver glue, one IVP per object,
// sequential, fp64, no FMA contraction (build with -ffp-contract=off), glibc pow/sqrt.
//
//   Ida struct / new            <- /root/reference/src/lib.rs:89-244, 278-405
//   Ida::solve                  <- /root/reference/src/impl_solve.rs:69-376
//   Ida::step                   <- /root/reference/src/lib.rs:613-711
//   set_coeffs                  <- /root/reference/src/lib.rs:722-782
//   nonlinear_solve             <- /root/reference/src/lib.rs:787-890
//   predict                     <- /root/reference/src/lib.rs:894-959
//   test_error                  <- /root/reference/src/lib.rs:967-1039
//   restore                     <- /root/reference/src/lib.rs:1044-1083
//   handle_n_flag               <- /root/reference/src/lib.rs:1120-1244
//   reset                       <- /root/reference/src/lib.rs:1249-1252
//   get_solution                <- /root/reference/src/lib.rs:1274-1343
//   get_dky                     <- /root/reference/src/lib.rs:424-529 (quirk Q9: see the function)
//   complete_step               <- /root/reference/src/impl_complete_step.rs:22-177
//   stop_test1 / stop_test2     <- /root/reference/src/impl_stop_test.rs:36-125, 146-211
//   r_check1/2/3, root_find     <- /root/reference/src/impl_r_check.rs:32-576
//   IdaNLProblem sys/setup/solve/ctest <- /root/reference/src/ida_nls.rs:118-266
//   IdaLProblem setup/solve     <- /root/reference/src/ida_ls.rs:232-290, 298-455
//   TolControlSS / SV           <- /root/reference/src/tol_control.rs:36-44, 71-82
//   constants                   <- /root/reference/src/constants.rs
//
// Deliberate deviations from the reference text (none is exercised by a reference golden):
//   Q1  jac is evaluated at tn (reference passes 0.0, marked "TODO fix", ida_ls.rs:258-262).
//   Q2  LU failure is a recoverable lsetup failure (reference unwraps -> panic, ida_ls.rs:287).
//   Q4  a Newton ConvergenceRecover is recoverable at step level: ncf++, ncfn++, h *= 1/4, fail after maxncf
//       (reference's downcast makes it fatal, lib.rs:1133-1140).
//   Q5  reset() rescales phi[1] only (reference rescales all of phi incl. the solution, lib.rs:1249-1252).
// Kept as in the reference: Q7 (`ypnorm > 2/hh`, impl_solve.rs:127), Q8 (iroots = signum(glo)),
// Q11 (nni is the Newton counter), Q13 (t0 = 0 unless set explicitly).
//
// Pinned by the reference's state-injection goldens (src/tests/*.rs -> tests/golden/stepper_goldens.json)
// and by the end-to-end Roberts run (examples/roberts.rs:21-25 reference solution, 377 attempt frames in
// scripts/data_trace.ipynb) -- see tests/test_oracle_stepper.py, tests/test_oracle_roberts.py.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

#include "dense.hpp"
#include "newton.hpp"
#include "problems.hpp"

namespace oracle {

// constants.rs
constexpr double HMAX_INV_DEFAULT = 0.0;
constexpr int MAXORD_DEFAULT = 5;
constexpr int MXORDP1 = 6;
constexpr long MXSTEP_DEFAULT = 500;
constexpr int MXNCF = 10;
constexpr int MXNEF = 10;
constexpr double EPCON = 0.33;
constexpr double XRATE = 0.25;
constexpr int MAXNLSIT = 4;
constexpr double RATEMAX = 0.9;  // ida_nls.rs:15

enum IdaTask { IDA_NORMAL = 0, IDA_ONE_STEP = 1 };

// solve() return codes. >= 0: IdaSolveStatus (lib.rs:58-64); < 0: IdaError (error.rs), SUNDIALS numbering.
enum IdaStatus {
    IDA_SUCCESS = 0,
    IDA_TSTOP_RETURN = 1,
    IDA_ROOT_RETURN = 2,
    IDA_CONTINUE_STEPS = 99,
    IDA_TOO_MUCH_WORK = -1,
    IDA_TOO_MUCH_ACC = -2,
    IDA_ERR_FAIL = -3,
    IDA_CONV_FAIL = -4,
    IDA_LSETUP_FAIL = -6,
    IDA_REP_RES_ERR = -9,
    IDA_CONSTR_FAIL = -11,
    IDA_ILL_INPUT = -22,
    IDA_BAD_K = -25,
    IDA_BAD_T = -26,
    IDA_BAD_TSTOP = -27,
    IDA_CLOSE_ROOTS = -30,
};

// Failure classes fed to handle_n_flag (lib.rs:1120-1244)
enum NFlag { NFLAG_NONE = 0, NFLAG_TEST_FAIL = 1, NFLAG_CONV_RECVR = 2, NFLAG_LSETUP_RECVR = 3 };

// tol_control.rs
struct TolControl {
    double rtol = 0.0;
    double atol_s = 0.0;
    std::vector<double> atol_v;  // empty => scalar atol (TolControlSS), else TolControlSV
    void ewt_set(const double* ycur, double* ewt, int n) const {
        if (atol_v.empty()) {
            for (int i = 0; i < n; ++i) ewt[i] = 1.0 / (rtol * std::fabs(ycur[i]) + atol_s);
        } else {
            for (int i = 0; i < n; ++i) ewt[i] = 1.0 / (rtol * std::fabs(ycur[i]) + atol_v[i]);
        }
    }
};

// crates/linear/src/dense.rs:15-64 -- the `Dense` LSolver (owns only the pivots)
struct Dense {
    int n;
    std::vector<int64_t> pivots;
    explicit Dense(int n_) : n(n_), pivots(n_, 0) {}
    int setup(double* mat_a) { return dense_get_rf(mat_a, n, n, pivots.data()); }  // 0 | 1-based col
    void solve(const double* mat_a, double* x, const double* b) const {
        for (int i = 0; i < n; ++i) x[i] = b[i];  // x <- b (dense.rs:59)
        dense_get_rs(mat_a, n, pivots.data(), x);
    }
};

// ida_ls.rs
struct IdaLProblem {
    int n;
    Dense ls;
    std::vector<double> mat_j;  // column-major
    std::vector<double> x;
    long nje = 0, ncfl = 0, nre_dq = 0;
    double ida_cj = 0.0, ida_cjold = 0.0, ida_cjratio = 0.0;
    const Problem* problem;
    int last_lu_info = 0;

    IdaLProblem(const Problem* p) : n(p->model_size()), ls(n), mat_j((size_t)n * n, 0.0), x(n, 0.0), problem(p) {}

    // ida_ls.rs:232-290
    int setup(double tn, const double* y, const double* yp, const double* r) {
        nje += 1;
        std::fill(mat_j.begin(), mat_j.end(), 0.0);
        problem->jac(tn /* Q1 */, ida_cj, y, yp, r, mat_j.data());
        last_lu_info = ls.setup(mat_j.data());
        return last_lu_info;
    }
    // ida_ls.rs:298-455 (Direct solver branch)
    void solve(double* b) {
        std::fill(x.begin(), x.end(), 0.0);
        ls.solve(mat_j.data(), x.data(), b);
        for (int i = 0; i < n; ++i) b[i] = x[i];
        if (ida_cjratio != 1.0) {
            const double s = 2.0 / (1.0 + ida_cjratio);
            for (int i = 0; i < n; ++i) b[i] *= s;
        }
    }
};

// ida_nls.rs
struct IdaNLProblem : NLProblem {
    int n;
    std::vector<double> ida_yy, ida_yp, ida_yypredict, ida_yppredict, ida_ewt, ida_savres;
    double ida_tn = 0.0;
    double ida_ss = 0.0, ida_oldnrm = 0.0, ida_toldel = 0.0;
    long ida_nre = 0, ida_nsetups = 0;
    IdaLProblem lp;

    IdaNLProblem(const Problem* p, const double* yy0, const double* yp0)
        : n(p->model_size()), ida_yy(yy0, yy0 + n), ida_yp(yp0, yp0 + n), ida_yypredict(n, 0.0), ida_yppredict(n, 0.0),
          ida_ewt(n, 0.0), ida_savres(n, 0.0), lp(p) {}

    // idaNlsResidual, ida_nls.rs:118-153
    int sys(const double* ycor, double* res) override {
        for (int i = 0; i < n; ++i) ida_yy[i] = ida_yypredict[i] + ycor[i];
        for (int i = 0; i < n; ++i) ida_yp[i] = ida_yppredict[i] + lp.ida_cj * ycor[i];  // scaled_add: mul, add
        lp.problem->res(ida_tn, ida_yy.data(), ida_yp.data(), res);
        ida_nre += 1;
        for (int i = 0; i < n; ++i) ida_savres[i] = res[i];
        return NLS_SUCCESS;
    }
    // idaNlsLSetup, ida_nls.rs:156-187
    int setup(const double*, const double* res, bool, bool* jcur) override {
        ida_nsetups += 1;
        const int info = lp.setup(ida_tn, ida_yy.data(), ida_yp.data(), res);
        *jcur = true;
        lp.ida_cjold = lp.ida_cj;
        lp.ida_cjratio = 1.0;
        ida_ss = 20.0;
        return info == 0 ? NLS_SUCCESS : NLS_LSETUP_RECVR;  // Q2
    }
    // idaNlsLSolve, ida_nls.rs:190-215
    int solve(const double*, double* delta) override {
        lp.solve(delta);
        return NLS_SUCCESS;
    }
    // idaNlsConvTest, ida_nls.rs:218-266
    int ctest(const Newton& solver, const double*, const double* del, double tol, const double* ewt,
              bool* converged) override {
        const double delnrm = norm_wrms(del, ewt, n);
        const int m = solver.get_cur_iter();
        *converged = false;
        if (m == 0) {
            ida_oldnrm = delnrm;
            if (delnrm <= 0.0001 * ida_toldel) {
                *converged = true;
                return NLS_SUCCESS;
            }
        } else {
            const double base = delnrm / ida_oldnrm;
            const double arg = 1.0 / (double)m;
            const double rate = std::pow(base, arg);
            if (rate > RATEMAX) return NLS_CONV_RECVR;
            ida_ss = rate / (1.0 - rate);
        }
        if (ida_ss * delnrm <= tol) *converged = true;
        return NLS_SUCCESS;
    }
};

struct StepRecord {  // one accepted step, for parity traces
    double tn, hused;
    int kused;
    long nni, nsetups;
};

struct Ida {
    int n;
    const Problem* problem;
    TolControl tol_control;

    bool ida_setup_done = false;
    bool ida_suppressalg = false;

    std::vector<double> ida_phi;  // [MXORDP1][n]
    double ida_psi[MXORDP1] = {0}, ida_alpha[MXORDP1] = {0}, ida_beta[MXORDP1] = {0}, ida_sigma[MXORDP1] = {0},
           ida_gamma[MXORDP1] = {0};
    std::vector<double> ida_delta, ida_ee;
    std::vector<uint8_t> ida_id;

    bool has_tstop = false;
    double ida_tstop = 0.0;

    int ida_kk = 0, ida_kused = 0, ida_knew = 0, ida_phase = 0, ida_ns = 0;
    double ida_hin = 0.0, ida_h0u = 0.0, ida_hh = 0.0, ida_hused = 0.0, ida_rr = 0.0;
    double ida_tretlast = 0.0, ida_cjlast = 0.0;
    double ida_eps_newt = 0.0, ida_epcon = EPCON;
    long ida_maxncf = MXNCF, ida_maxnef = MXNEF;
    int ida_maxord = MAXORD_DEFAULT;
    long ida_mxstep = MXSTEP_DEFAULT;
    double ida_hmax_inv = HMAX_INV_DEFAULT;

    long ida_nst = 0, ida_ncfn = 0, ida_netf = 0;

    double ida_cvals[MXORDP1] = {0}, ida_dvals[MAXORD_DEFAULT] = {0};
    double ida_tolsf = 1.0;

    // rootfinding
    int ida_nrtfn;
    std::vector<double> ida_iroots, ida_glo, ida_ghi, ida_grout;
    std::vector<uint8_t> ida_rootdir, ida_gactive;
    double ida_tlo = 0.0, ida_thi = 0.0, ida_trout = 0.0, ida_toutc = 0.0, ida_ttol = 0.0;
    IdaTask ida_taskc = IDA_NORMAL;
    bool ida_irfnd = false;
    long ida_nge = 0;
    int ida_mxgnull = 1;

    std::vector<double> ida_zvec;  // zvecs[0]

    Newton nls;
    IdaNLProblem nlp;

    // instrumentation (not in the reference)
    long n_attempts = 0;
    bool record_steps = false;
    std::vector<StepRecord> steps;

    double* phi(int j) { return ida_phi.data() + (size_t)j * n; }
    const double* phi(int j) const { return ida_phi.data() + (size_t)j * n; }

    // lib.rs:278-405
    Ida(const Problem* p, const double* yy0, const double* yp0, const TolControl& tc, double t0 = 0.0)
        : n(p->model_size()), problem(p), tol_control(tc), ida_phi((size_t)MXORDP1 * n, 0.0), ida_delta(n, 0.0), ida_ee(n, 0.0),
          ida_id(n, 0), ida_nrtfn(p->num_roots()), ida_iroots(ida_nrtfn, 0.0), ida_glo(ida_nrtfn, 0.0), ida_ghi(ida_nrtfn, 0.0),
          ida_grout(ida_nrtfn, 0.0), ida_rootdir(ida_nrtfn, 0), ida_gactive(ida_nrtfn, 0 /* sic: false, lib.rs:373 */),
          ida_zvec(n, 0.0), nls(n, MAXNLSIT), nlp(p, yy0, yp0) {
        for (int i = 0; i < n; ++i) {
            phi(0)[i] = yy0[i];
            phi(1)[i] = yp0[i];
        }
        nlp.ida_tn = t0;  // Q13: the reference hard-wires 0 (ida_nls.rs:90)
        ida_tretlast = t0;
    }

    double wrms_norm(const double* x, const double* w, bool mask) const {  // lib.rs:1353-1370
        return mask ? norm_wrms_masked(x, w, ida_id.data(), n) : norm_wrms(x, w, n);
    }

    // ---------------------------------------------------------------- solve (impl_solve.rs:69-376)
    int solve(double tout, double* tret, IdaTask itask) {
        const double eps = std::numeric_limits<double>::epsilon();
        if (itask == IDA_NORMAL) ida_toutc = tout;
        ida_taskc = itask;

        if (ida_nst == 0) {
            if (!ida_setup_done) {
                tol_control.ewt_set(phi(0), nlp.ida_ewt.data(), n);  // initial_setup, lib.rs:537-545
                ida_setup_done = true;
            }
            const double tdist = std::fabs(tout - nlp.ida_tn);
            if (tdist == 0.0) return IDA_ILL_INPUT;
            const double troundoff = 2.0 * eps * (std::fabs(nlp.ida_tn) + std::fabs(tout));
            if (tdist < troundoff) return IDA_ILL_INPUT;

            ida_hh = ida_hin;
            if (ida_hh != 0.0 && (tout - nlp.ida_tn) * ida_hh < 0.0) return IDA_ILL_INPUT;

            if (ida_hh == 0.0) {
                ida_hh = 0.001 * tdist;
                const double ypnorm = wrms_norm(phi(1), nlp.ida_ewt.data(), ida_suppressalg);
                if (ypnorm > 2.0 / ida_hh) ida_hh = 0.5 / ypnorm;  // Q7 kept
                if (tout < nlp.ida_tn) ida_hh = -ida_hh;
            }
            const double rh = std::fabs(ida_hh) * ida_hmax_inv;
            if (rh > 1.0) ida_hh /= rh;

            if (has_tstop) {
                if ((ida_tstop - nlp.ida_tn) * ida_hh <= 0.0) return IDA_ILL_INPUT;
                if ((nlp.ida_tn + ida_hh - ida_tstop) * ida_hh > 0.0) ida_hh = (ida_tstop - nlp.ida_tn) * (1.0 - 4.0 * eps);
            }

            ida_h0u = ida_hh;
            ida_kk = 0;
            ida_kused = 0;

            if (ida_nrtfn > 0) r_check1();

            for (int i = 0; i < n; ++i) phi(1)[i] *= ida_hh;  // phi[1] = hh*y'

            ida_eps_newt = ida_epcon;
            nlp.ida_toldel = 0.0001 * ida_eps_newt;
        }

        long nstloc = 0;

        if (ida_nst > 0) {
            if (ida_nrtfn > 0) {
                const bool irfndp = ida_irfnd;
                int ier = r_check2();
                if (ier < 0) return ier;
                if (ier == IDA_ROOT_RETURN) {
                    ida_tretlast = ida_tlo;
                    *tret = ida_tlo;
                    return IDA_ROOT_RETURN;
                }
                const double troundoff = (std::fabs(nlp.ida_tn) + std::fabs(ida_hh)) * eps * 100.0;
                if (std::fabs(nlp.ida_tn - ida_tretlast) > troundoff) {
                    ier = r_check3();
                    if (ier < 0) return ier;
                    if (ier == IDA_CONTINUE_STEPS) {
                        ida_irfnd = false;
                        if (itask == IDA_ONE_STEP && irfndp) {
                            ida_tretlast = nlp.ida_tn;
                            *tret = nlp.ida_tn;
                            get_solution(nlp.ida_tn);
                            return IDA_SUCCESS;
                        }
                    } else {  // root found
                        ida_irfnd = true;
                        ida_tretlast = ida_tlo;
                        *tret = ida_tlo;
                        return IDA_ROOT_RETURN;
                    }
                }
            }
            const int istate = stop_test1(tout, tret, itask);
            if (istate != IDA_CONTINUE_STEPS) return istate;
        }

        for (;;) {
            if (ida_mxstep > 0 && nstloc >= ida_mxstep) {
                *tret = nlp.ida_tn;
                ida_tretlast = nlp.ida_tn;
                return IDA_TOO_MUCH_WORK;
            }
            if (ida_nst > 0) {
                tol_control.ewt_set(phi(0), nlp.ida_ewt.data(), n);
                for (int i = 0; i < n; ++i) {
                    if (nlp.ida_ewt[i] <= 0.0) {
                        get_solution(nlp.ida_tn);
                        *tret = nlp.ida_tn;
                        ida_tretlast = nlp.ida_tn;
                        return IDA_ILL_INPUT;
                    }
                }
            }
            const double nrm = wrms_norm(phi(0), nlp.ida_ewt.data(), ida_suppressalg);
            ida_tolsf = eps * nrm;
            if (ida_tolsf > 1.0) {
                ida_tolsf *= 10.0;
                *tret = nlp.ida_tn;
                ida_tretlast = nlp.ida_tn;
                if (ida_nst > 0) get_solution(nlp.ida_tn);
                return IDA_TOO_MUCH_ACC;
            }

            const int sflag = step();
            if (sflag != IDA_SUCCESS) {
                if (get_solution(nlp.ida_tn) == IDA_SUCCESS) {
                    *tret = nlp.ida_tn;
                    ida_tretlast = nlp.ida_tn;
                }
                return sflag;
            }
            nstloc += 1;

            if (ida_nrtfn > 0) {
                const int ier = r_check3();
                if (ier < 0) return ier;
                if (ier == IDA_ROOT_RETURN) {
                    ida_irfnd = true;
                    ida_tretlast = ida_tlo;
                    *tret = ida_tlo;
                    return IDA_ROOT_RETURN;
                }
            }
            const int istate = stop_test2(tout, tret, itask);
            if (istate != IDA_CONTINUE_STEPS) return istate;
        }
    }

    // ---------------------------------------------------------------- step (lib.rs:613-711)
    int step() {
        const double saved_t = nlp.ida_tn;
        if (ida_nst == 0) {
            ida_kk = 1;
            ida_kused = 0;
            ida_hused = 0.0;
            ida_psi[0] = ida_hh;
            nlp.lp.ida_cj = 1.0 / ida_hh;
            ida_phase = 0;
            ida_ns = 0;
        }
        long ncf = 0, nef = 0;
        double ck, err_k = 0.0, err_km1 = 0.0;
        for (;;) {
            n_attempts += 1;
            ck = set_coeffs();
            nlp.ida_tn += ida_hh;
            if (has_tstop) {
                if ((nlp.ida_tn - ida_tstop) * ida_hh > 1.0 /* Q6 kept */) nlp.ida_tn = ida_tstop;
            }
            predict();

            int nflag = NFLAG_NONE;
            err_k = 0.0;
            err_km1 = 0.0;
            const int nls_ret = nonlinear_solve();
            if (nls_ret == NLS_SUCCESS) {
                if (!test_error(ck, &err_k, &err_km1)) nflag = NFLAG_TEST_FAIL;
            } else if (nls_ret == NLS_CONV_RECVR) {
                nflag = NFLAG_CONV_RECVR;
            } else if (nls_ret == NLS_LSETUP_RECVR) {
                nflag = NFLAG_LSETUP_RECVR;
            } else {
                restore(saved_t);
                return IDA_LSETUP_FAIL;
            }
            if (nflag == NFLAG_NONE) break;

            restore(saved_t);
            const int kflag = handle_n_flag(nflag, err_k, err_km1, &ncf, &nef);
            if (kflag != IDA_SUCCESS) return kflag;
            if (ida_nst == 0) reset();
        }
        complete_step(err_k, err_km1);
        for (int i = 0; i < n; ++i) ida_ee[i] *= ck;  // lib.rs:708
        if (record_steps) steps.push_back({nlp.ida_tn, ida_hused, ida_kused, nls.niters, nlp.ida_nsetups});
        return IDA_SUCCESS;
    }

    // ---------------------------------------------------------------- set_coeffs (lib.rs:722-782)
    double set_coeffs() {
        if (ida_hh != ida_hused || ida_kk != ida_kused) ida_ns = 0;
        ida_ns = std::min(ida_ns + 1, ida_kused + 2);
        if (ida_kk + 1 >= ida_ns) {
            ida_beta[0] = 1.0;
            ida_alpha[0] = 1.0;
            double temp1 = ida_hh;
            ida_gamma[0] = 0.0;
            ida_sigma[0] = 1.0;
            for (int i = 1; i <= ida_kk; ++i) {
                const double scalar_i = (double)i;
                const double temp2 = ida_psi[i - 1];
                ida_psi[i - 1] = temp1;
                ida_beta[i] = ida_beta[i - 1] * ida_psi[i - 1] / temp2;
                temp1 = temp2 + ida_hh;
                ida_alpha[i] = ida_hh / temp1;
                ida_sigma[i] = scalar_i * ida_sigma[i - 1] * ida_alpha[i];
                ida_gamma[i] = ida_gamma[i - 1] + ida_alpha[i - 1] / ida_hh;
            }
            ida_psi[ida_kk] = temp1;
        }
        double alphas = 0.0, alpha0 = 0.0;
        for (int i = 0; i < ida_kk; ++i) {
            const double scalar_i = (double)(i + 1);
            alphas -= 1.0 / scalar_i;
            alpha0 -= ida_alpha[i];
        }
        ida_cjlast = nlp.lp.ida_cj;
        nlp.lp.ida_cj = -alphas / ida_hh;

        double ck = std::fabs(ida_alpha[ida_kk] + alphas - alpha0);
        ck = std::fmax(ck, ida_alpha[ida_kk]);

        if (ida_ns <= ida_kk) {
            for (int j = ida_ns; j <= ida_kk; ++j) {
                double* p = phi(j);
                const double b = ida_beta[j];
                for (int i = 0; i < n; ++i) p[i] *= b;
            }
        }
        return ck;
    }

    // ---------------------------------------------------------------- nonlinear_solve (lib.rs:787-890)
    int nonlinear_solve() {
        bool call_lsetup = false;
        if (ida_nst == 0) {
            nlp.lp.ida_cjold = nlp.lp.ida_cj;
            nlp.ida_ss = 20.0;
            call_lsetup = true;
        }
        nlp.lp.ida_cjratio = nlp.lp.ida_cj / nlp.lp.ida_cjold;
        const double temp1 = (1.0 - XRATE) / (1.0 + XRATE);
        const double temp2 = 1.0 / temp1;
        if (nlp.lp.ida_cjratio < temp1 || nlp.lp.ida_cjratio > temp2) call_lsetup = true;
        if (nlp.lp.ida_cj != ida_cjlast) nlp.ida_ss = 100.0;

        std::fill(ida_delta.begin(), ida_delta.end(), 0.0);
        const std::vector<double> w = nlp.ida_ewt;  // lib.rs:828

        const int retval = nls.solve(nlp, ida_delta.data(), ida_ee.data(), w.data(), ida_eps_newt, call_lsetup);

        // lib.rs:845-849 (always, even on failure)
        for (int i = 0; i < n; ++i) nlp.ida_yy[i] = nlp.ida_yypredict[i] + ida_ee[i];
        for (int i = 0; i < n; ++i) nlp.ida_yp[i] = nlp.ida_yppredict[i] + nlp.lp.ida_cj * ida_ee[i];
        return retval;
    }

    // ---------------------------------------------------------------- predict (lib.rs:894-959)
    void predict() {
        double* yyp = nlp.ida_yypredict.data();
        double* ypp = nlp.ida_yppredict.data();
        for (int i = 0; i < n; ++i) yyp[i] = 0.0;
        for (int j = 0; j <= ida_kk; ++j) {
            const double* p = phi(j);
            for (int i = 0; i < n; ++i) yyp[i] += p[i];
        }
        for (int i = 0; i < n; ++i) ypp[i] = 0.0;
        for (int j = 1; j <= ida_kk; ++j) {
            const double* p = phi(j);
            const double g = ida_gamma[j];
            for (int i = 0; i < n; ++i) ypp[i] += g * p[i];  // scaled_add: mul then add
        }
    }

    // ---------------------------------------------------------------- test_error (lib.rs:967-1039)
    bool test_error(double ck, double* err_k_out, double* err_km1_out) {
        const double scalar_kk = (double)ida_kk;
        const double* ewt = nlp.ida_ewt.data();
        const double enorm_k = wrms_norm(ida_ee.data(), ewt, ida_suppressalg);
        const double err_k = ida_sigma[ida_kk] * enorm_k;
        const double terr_k = err_k * (scalar_kk + 1.0);
        double err_km1 = 0.0;
        int knew = ida_kk;
        if (ida_kk > 1) {
            const double* pk = phi(ida_kk);
            for (int i = 0; i < n; ++i) ida_delta[i] = pk[i] + ida_ee[i];
            const double enorm_km1 = wrms_norm(ida_delta.data(), ewt, ida_suppressalg);
            err_km1 = ida_sigma[ida_kk - 1] * enorm_km1;
            const double terr_km1 = scalar_kk * err_km1;
            if (ida_kk > 2) {
                const double* pkm1 = phi(ida_kk - 1);
                for (int i = 0; i < n; ++i) ida_delta[i] += pkm1[i];
                const double enorm_km2 = wrms_norm(ida_delta.data(), ewt, ida_suppressalg);
                const double err_km2 = ida_sigma[ida_kk - 2] * enorm_km2;
                const double terr_km2 = (scalar_kk - 1.0) * err_km2;
                if (std::fmax(terr_km1, terr_km2) <= terr_k) knew = ida_kk - 1;
            } else {
                if (terr_km1 <= terr_k * 0.5) knew = ida_kk - 1;
            }
        }
        ida_knew = knew;
        *err_k_out = err_k;
        *err_km1_out = err_km1;
        return (ck * enorm_k) <= 1.0;
    }

    // ---------------------------------------------------------------- restore (lib.rs:1044-1083)
    void restore(double saved_t) {
        nlp.ida_tn = saved_t;
        for (int j = 1; j < ida_kk + 1; ++j) ida_psi[j - 1] = ida_psi[j] - ida_hh;
        if (ida_ns <= ida_kk) {
            for (int j = ida_ns; j <= ida_kk; ++j) ida_cvals[j - ida_ns] = 1.0 / ida_beta[j];
            for (int j = ida_ns; j <= ida_kk; ++j) {
                double* p = phi(j);
                const double c = ida_cvals[j - ida_ns];
                for (int i = 0; i < n; ++i) p[i] *= c;
            }
        }
    }

    // ---------------------------------------------------------------- handle_n_flag (lib.rs:1120-1244)
    int handle_n_flag(int nflag, double err_k, double err_km1, long* ncf, long* nef) {
        ida_phase = 1;
        if (nflag == NFLAG_TEST_FAIL) {
            *nef += 1;
            ida_netf += 1;
            if (*nef == 1) {
                const double err_knew = (ida_kk == ida_knew) ? err_k : err_km1;
                ida_kk = ida_knew;
                {
                    const double base = 2.0 * err_knew + 0.0001;
                    const double arg = 1.0 / (double)(ida_kk + 1);
                    ida_rr = 0.9 * std::pow(base, -arg);
                }
                ida_rr = std::fmax(0.25, std::fmin(0.9, ida_rr));
                ida_hh *= ida_rr;
                return IDA_SUCCESS;
            } else if (*nef == 2) {
                ida_kk = ida_knew;
                ida_rr = 0.25;
                ida_hh *= ida_rr;
                return IDA_SUCCESS;
            } else if (*nef < ida_maxnef) {
                ida_kk = 1;
                ida_rr = 0.25;
                ida_hh *= ida_rr;
                return IDA_SUCCESS;
            }
            return IDA_ERR_FAIL;
        }
        // recoverable convergence-type failure (Q4: C semantics)
        *ncf += 1;
        ida_ncfn += 1;
        ida_rr = 0.25;
        ida_hh *= ida_rr;
        if (*ncf < ida_maxncf) return IDA_SUCCESS;
        return IDA_CONV_FAIL;
    }

    // ---------------------------------------------------------------- reset (lib.rs:1249-1252, Q5)
    void reset() {
        ida_psi[0] = ida_hh;
        for (int i = 0; i < n; ++i) phi(1)[i] *= ida_rr;
    }

    // ---------------------------------------------------------------- complete_step (impl_complete_step.rs:22-177)
    void complete_step(double err_k, double err_km1) {
        ida_nst += 1;
        const int kdiff = ida_kk - ida_kused;
        ida_kused = ida_kk;
        ida_hused = ida_hh;

        if (ida_knew == ida_kk - 1 || ida_kk == ida_maxord) ida_phase = 1;

        if (ida_phase == 0) {
            if (ida_nst > 1) {
                ida_kk += 1;
                double hnew = 2.0 * ida_hh;
                const double tmp = std::fabs(hnew) * ida_hmax_inv;
                if (tmp > 1.0) hnew /= tmp;
                ida_hh = hnew;
            }
        } else {
            enum { LOWER, MAINTAIN, RAISE } action;
            double err_kp1 = 0.0;
            if (ida_knew == ida_kk - 1) {
                action = LOWER;
            } else if (ida_kk == ida_maxord) {
                action = MAINTAIN;
            } else if (ida_kk + 1 >= ida_ns || kdiff == 1) {
                action = MAINTAIN;
            } else {
                const double* pk1 = phi(ida_kk + 1);
                std::vector<double> temp(n);
                for (int i = 0; i < n; ++i) temp[i] = ida_ee[i] - pk1[i];
                const double enorm = wrms_norm(temp.data(), nlp.ida_ewt.data(), ida_suppressalg);
                err_kp1 = enorm / (double)(ida_kk + 2);
                const double terr_k = (double)(ida_kk + 1) * err_k;
                const double terr_kp1 = (double)(ida_kk + 2) * err_kp1;
                if (ida_kk == 1) {
                    action = (terr_kp1 >= 0.5 * terr_k) ? MAINTAIN : RAISE;
                } else {
                    const double terr_km1 = (double)ida_kk * err_km1;
                    if (terr_km1 <= std::fmin(terr_k, terr_kp1)) action = LOWER;
                    else if (terr_kp1 >= terr_k) action = MAINTAIN;
                    else action = RAISE;
                }
            }
            double err_knew;
            if (action == RAISE) {
                ida_kk += 1;
                err_knew = err_kp1;
            } else if (action == LOWER) {
                ida_kk -= 1;
                err_knew = err_km1;
            } else {
                err_knew = err_k;
            }
            double hnew = ida_hh;
            {
                const double base = 2.0 * err_knew + 0.0001;
                const double arg = -(1.0 / (double)(ida_kk + 1));
                ida_rr = std::pow(base, arg);
            }
            if (ida_rr >= 2.0) {
                hnew = 2.0 * ida_hh;
                const double tmp = std::fabs(hnew) * ida_hmax_inv;
                if (tmp > 1.0) hnew /= tmp;
            } else if (ida_rr <= 1.0) {
                ida_rr = std::fmax(0.5, std::fmin(ida_rr, 0.9));
                hnew = ida_hh * ida_rr;
            }
            ida_hh = hnew;
        }

        if (ida_kused < ida_maxord) {
            double* p = phi(ida_kused + 1);
            for (int i = 0; i < n; ++i) p[i] = ida_ee[i];
        }
        // recurrence: tmp = ee; for j = kused..0: tmp += phi[j]; phi[j] = tmp
        for (int i = 0; i < n; ++i) ida_zvec[i] = ida_ee[i];
        for (int j = ida_kused; j >= 0; --j) {
            double* p = phi(j);
            for (int i = 0; i < n; ++i) {
                ida_zvec[i] += p[i];
                p[i] = ida_zvec[i];
            }
        }
    }

    // ---------------------------------------------------------------- get_solution (lib.rs:1274-1343)
    int get_solution(double t) {
        const double eps = std::numeric_limits<double>::epsilon();
        const double sgn = (ida_hh > 0.0 || (ida_hh == 0.0 && !std::signbit(ida_hh))) ? 1.0 : -1.0;  // f64::signum
        const double tfuzz = 100.0 * eps * (std::fabs(nlp.ida_tn) + std::fabs(ida_hh)) * sgn;
        const double tp = nlp.ida_tn - ida_hused - tfuzz;
        if ((t - tp) * ida_hh < 0.0) return IDA_BAD_T;

        const int kord = (ida_kused == 0) ? 1 : ida_kused;
        const double delt = t - nlp.ida_tn;
        double c = 1.0, d = 0.0;
        double gam = delt / ida_psi[0];
        ida_cvals[0] = c;
        for (int j = 1; j <= kord; ++j) {
            d = d * gam + c / ida_psi[j - 1];
            c = c * gam;
            gam = (delt + ida_psi[j - 1]) / ida_psi[j];
            ida_cvals[j] = c;
            ida_dvals[j - 1] = d;
        }
        double* yy = nlp.ida_yy.data();
        double* yp = nlp.ida_yp.data();
        for (int i = 0; i < n; ++i) yy[i] = 0.0;
        for (int j = 0; j <= kord; ++j) {
            const double* p = phi(j);
            const double cj_ = ida_cvals[j];
            for (int i = 0; i < n; ++i) yy[i] += cj_ * p[i];
        }
        for (int i = 0; i < n; ++i) yp[i] = 0.0;
        for (int j = 1; j <= kord; ++j) {
            const double* p = phi(j);
            const double dj = ida_dvals[j - 1];
            for (int i = 0; i < n; ++i) yp[i] += dj * p[i];
        }
        return IDA_SUCCESS;
    }

    // ---------------------------------------------------------------- get_dky (lib.rs:424-529, IDAGetDky)
    // Coefficients c_j^(k)(t) of the k-th derivative of the interpolating polynomial (recurrence of lib.rs:464-508), then
    // dky = sum_{j=k..kused} c_j^(k) phi_j accumulated from zero in ascending j, product first (lib.rs:517-526: an array of
    // products, then ndarray's row-by-row sum_axis).
    // Quirk Q9: the reference bounds the inner loops by `kused - k + 1` where C IDA has `kused - k + i`
    // (the C line survives as a comment, lib.rs:498,506). For k <= 1 the coefficients that enter the sum are the same; for
    // k >= 2 the reference leaves c_j^(k) for j > kused - k + 1 at zero and so drops terms of the derivative, and for k = 0
    // with kused = 5 it indexes cjk[6] out of bounds (a panic). `literal_q9` = true restates the reference loop bound as it is
    // (clamped to the array) for the cross-check in tests/; the default, which the product mirrors, is the C IDA bound.
    int get_dky_coeffs(double t, int k, double* cjk, bool literal_q9 = false) const {
        if (k < 0 || k > ida_kused) return IDA_BAD_K;
        const double eps = std::numeric_limits<double>::epsilon();
        const double sgn = (ida_hh > 0.0 || (ida_hh == 0.0 && !std::signbit(ida_hh))) ? 1.0 : -1.0;  // f64::signum
        const double tfuzz = 100.0 * eps * (std::fabs(nlp.ida_tn) + std::fabs(ida_hh)) * sgn;
        const double tp = nlp.ida_tn - ida_hused - tfuzz;
        if ((t - tp) * ida_hh < 0.0) return IDA_BAD_T;
        double cjk_1[MXORDP1 + 1] = {0.0};
        double work[MXORDP1 + 1] = {0.0};  // one spare slot: the literal bound writes c[kused + 1] when k = 0
        const double delt = t - nlp.ida_tn;
        double psij_1 = 0.0;
        for (int i = 0; i <= k; ++i) {
            const double scalar_i = (double)i;
            if (i == 0) {
                work[i] = 1.0;
            } else {
                work[i] = work[i - 1] * scalar_i / ida_psi[i - 1];
                psij_1 = ida_psi[i - 1];
            }
            int jlast = literal_q9 ? ida_kused - k + 1 : ida_kused - k + i;
            if (jlast > MXORDP1 - 1) jlast = MXORDP1 - 1;  // (literal bound only: psi has MXORDP1 entries)
            for (int j = i + 1; j <= jlast; ++j) {
                work[j] = (scalar_i * cjk_1[j - 1] + work[j - 1] * (delt + psij_1)) / ida_psi[j - 1];
                psij_1 = ida_psi[j - 1];
            }
            for (int j = i + 1; j <= jlast; ++j) cjk_1[j] = work[j];
        }
        for (int j = 0; j < MXORDP1; ++j) cjk[j] = work[j];
        return IDA_SUCCESS;
    }
    int get_dky(double t, int k, double* dky, bool literal_q9 = false) const {
        double cjk[MXORDP1];
        const int rc = get_dky_coeffs(t, k, cjk, literal_q9);
        if (rc != IDA_SUCCESS) return rc;
        for (int i = 0; i < n; ++i) dky[i] = 0.0;
        for (int j = k; j <= ida_kused; ++j) {
            const double* p = phi(j);
            for (int i = 0; i < n; ++i) dky[i] = dky[i] + p[i] * cjk[j];
        }
        return IDA_SUCCESS;
    }

    // ---------------------------------------------------------------- stop tests (impl_stop_test.rs)
    int stop_test1(double tout, double* tret, IdaTask itask) {
        const double eps = std::numeric_limits<double>::epsilon();
        if (has_tstop) {
            if ((nlp.ida_tn - ida_tstop) * ida_hh > 0.0) return IDA_BAD_TSTOP;
        }
        if (itask == IDA_NORMAL) {
            if (tout == ida_tretlast) {
                ida_tretlast = tout;
                *tret = tout;
                return IDA_SUCCESS;
            }
            if ((nlp.ida_tn - tout) * ida_hh >= 0.0) {
                const int ier = get_solution(tout);
                if (ier != IDA_SUCCESS) return ier;
                ida_tretlast = tout;
                *tret = tout;
                return IDA_SUCCESS;
            }
            if (has_tstop) {
                const double troundoff = 100.0 * eps * (std::fabs(nlp.ida_tn) + std::fabs(ida_hh));
                if (std::fabs(nlp.ida_tn - ida_tstop) <= troundoff) {
                    if (get_solution(ida_tstop) != IDA_SUCCESS) return IDA_BAD_TSTOP;
                    ida_tretlast = ida_tstop;
                    *tret = ida_tstop;
                    has_tstop = false;
                    return IDA_TSTOP_RETURN;
                }
                if ((nlp.ida_tn + ida_hh - ida_tstop) * ida_hh > 0.0) ida_hh = (ida_tstop - nlp.ida_tn) * (1.0 - 4.0 * eps);
            }
            return IDA_CONTINUE_STEPS;
        }
        // OneStep
        if ((nlp.ida_tn - ida_tretlast) * ida_hh > 0.0) {
            get_solution(nlp.ida_tn);
            ida_tretlast = nlp.ida_tn;
            *tret = nlp.ida_tn;
            return IDA_SUCCESS;
        }
        if (has_tstop) {
            const double troundoff = 100.0 * eps * (std::fabs(nlp.ida_tn) + std::fabs(ida_hh));
            if (std::fabs(nlp.ida_tn - ida_tstop) <= troundoff) {
                const int ier = get_solution(ida_tstop);
                if (ier != IDA_SUCCESS) return ier;
                ida_tretlast = ida_tstop;
                *tret = ida_tstop;
                return IDA_TSTOP_RETURN;
            }
            if ((nlp.ida_tn + ida_hh - ida_tstop) * ida_hh > 0.0) ida_hh = (ida_tstop - nlp.ida_tn) * (1.0 - 4.0 * eps);
        }
        return IDA_CONTINUE_STEPS;
    }

    int stop_test2(double tout, double* tret, IdaTask itask) {
        const double eps = std::numeric_limits<double>::epsilon();
        if (itask == IDA_NORMAL) {
            if ((nlp.ida_tn - tout) * ida_hh >= 0.0) {
                *tret = tout;
                ida_tretlast = tout;
                get_solution(tout);
                return IDA_SUCCESS;
            }
            if (has_tstop) {
                const double troundoff = 100.0 * eps * (std::fabs(nlp.ida_tn) + std::fabs(ida_hh));
                if (std::fabs(nlp.ida_tn - ida_tstop) <= troundoff) {
                    get_solution(ida_tstop);
                    *tret = ida_tstop;
                    ida_tretlast = ida_tstop;
                    has_tstop = false;
                    return IDA_TSTOP_RETURN;
                }
                if ((nlp.ida_tn + ida_hh - ida_tstop) * ida_hh > 0.0) ida_hh = (ida_tstop - nlp.ida_tn) * (1.0 - 4.0 * eps);
            }
            return IDA_CONTINUE_STEPS;
        }
        if (has_tstop) {
            const double troundoff = 100.0 * eps * (std::fabs(nlp.ida_tn) + std::fabs(ida_hh));
            if (std::fabs(nlp.ida_tn - ida_tstop) <= troundoff) {
                get_solution(ida_tstop);
                *tret = ida_tstop;
                ida_tretlast = ida_tstop;
                has_tstop = false;
                return IDA_TSTOP_RETURN;
            }
            if ((nlp.ida_tn + ida_hh - ida_tstop) * ida_hh > 0.0) ida_hh = (ida_tstop - nlp.ida_tn) * (1.0 - 4.0 * eps);
        }
        *tret = nlp.ida_tn;
        ida_tretlast = nlp.ida_tn;
        return IDA_SUCCESS;
    }

    // ---------------------------------------------------------------- rootfinding (impl_r_check.rs)
    static double signum(double x) {  // f64::signum: +-1 (by sign bit), NaN for NaN
        if (std::isnan(x)) return x;
        return std::signbit(x) ? -1.0 : 1.0;
    }

    void r_check1() {  // impl_r_check.rs:32-115
        const double eps = std::numeric_limits<double>::epsilon();
        std::fill(ida_iroots.begin(), ida_iroots.end(), 0.0);
        ida_tlo = nlp.ida_tn;
        ida_ttol = (std::fabs(nlp.ida_tn) + std::fabs(ida_hh)) * eps * 100.0;
        problem->root(ida_tlo, phi(0), phi(1), ida_glo.data());
        ida_nge = 1;
        bool zroot = false;
        for (int i = 0; i < ida_nrtfn; ++i) {
            if (std::fabs(ida_glo[i]) == 0.0) {
                ida_gactive[i] = 0;
                zroot = true;
            }
        }
        if (zroot) {
            const double hratio = std::fmax(ida_ttol / std::fabs(ida_hh), 0.1);
            const double smallh = hratio * ida_hh;
            const double tplus = ida_tlo + smallh;
            for (int i = 0; i < n; ++i) nlp.ida_yy[i] = phi(0)[i];
            for (int i = 0; i < n; ++i) nlp.ida_yy[i] += smallh * phi(1)[i];
            problem->root(tplus, nlp.ida_yy.data(), phi(1), ida_ghi.data());
            ida_nge += 1;
            for (int i = 0; i < ida_nrtfn; ++i) {
                if (!ida_gactive[i] && std::fabs(ida_ghi[i]) != 0.0) {
                    ida_gactive[i] = 1;
                    ida_glo[i] = ida_ghi[i];
                }
            }
        }
    }

    int r_check2() {  // impl_r_check.rs:117-219
        const double eps = std::numeric_limits<double>::epsilon();
        if (!ida_irfnd) return IDA_CONTINUE_STEPS;
        get_solution(ida_tlo);
        problem->root(ida_tlo, nlp.ida_yy.data(), nlp.ida_yp.data(), ida_glo.data());
        ida_nge += 1;
        std::fill(ida_iroots.begin(), ida_iroots.end(), 0.0);
        bool zroot = false;
        for (int i = 0; i < ida_nrtfn; ++i) {
            if (ida_gactive[i] && std::fabs(ida_glo[i]) == 0.0) {
                zroot = true;
                ida_iroots[i] = 1.0;
            }
        }
        if (zroot) {
            ida_ttol = (std::fabs(nlp.ida_tn) + std::fabs(ida_hh)) * eps * 100.0;
            const double smallh = ida_ttol * signum(ida_hh);
            const double tplus = ida_tlo + smallh;
            if ((tplus - nlp.ida_tn) * ida_hh >= 0.0) {
                const double hratio = smallh / ida_hh;
                for (int i = 0; i < n; ++i) nlp.ida_yy[i] += hratio * phi(1)[i];
            } else {
                get_solution(tplus);
            }
            problem->root(tplus, nlp.ida_yy.data(), nlp.ida_yp.data(), ida_ghi.data());
            ida_nge += 1;
            bool zroot2 = false;
            for (int i = 0; i < ida_nrtfn; ++i) {
                if (ida_gactive[i]) {
                    if (std::fabs(ida_ghi[i]) == 0.0) {
                        if (ida_iroots[i] > 0.0) return IDA_CLOSE_ROOTS;
                        zroot2 = true;
                        ida_iroots[i] = 1.0;
                    } else {
                        if (ida_iroots[i] > 0.0) ida_glo[i] = ida_ghi[i];
                    }
                }
            }
            if (zroot2) return IDA_ROOT_RETURN;
        }
        return IDA_CONTINUE_STEPS;
    }

    int r_check3() {  // impl_r_check.rs:221-280
        const double eps = std::numeric_limits<double>::epsilon();
        if (ida_taskc == IDA_ONE_STEP) {
            ida_thi = nlp.ida_tn;
        } else {
            ida_thi = ((ida_toutc - nlp.ida_tn) * ida_hh >= 0.0) ? nlp.ida_tn : ida_toutc;
        }
        get_solution(ida_thi);
        problem->root(ida_thi, nlp.ida_yy.data(), nlp.ida_yp.data(), ida_ghi.data());
        ida_nge += 1;
        ida_ttol = (std::fabs(nlp.ida_tn) + std::fabs(ida_hh)) * eps * 100.0;
        const int ier = root_find();
        for (int i = 0; i < ida_nrtfn; ++i) {
            if (!ida_gactive[i] && ida_grout[i] != 0.0) ida_gactive[i] = 1;
        }
        ida_tlo = ida_trout;
        ida_glo = ida_grout;
        if (ier == IDA_ROOT_RETURN) get_solution(ida_trout);
        return ier;
    }

    void scan_roots(const std::vector<double>& gval, bool first, bool* zroot, bool* sgnchg, int* imax) const {
        double maxfrac = 0.0;
        *zroot = false;
        *sgnchg = false;
        for (int i = 0; i < ida_nrtfn; ++i) {
            if (!ida_gactive[i]) continue;
            const bool rootdir_glo_neg = (double)ida_rootdir[i] * ida_glo[i] <= 0.0;
            if (first) {  // impl_r_check.rs:361-383
                if (std::fabs(gval[i]) == 0.0) {
                    if (rootdir_glo_neg) *zroot = true;
                    continue;
                }
            } else {  // impl_r_check.rs:486-504
                if (std::fabs(gval[i]) == 0.0 && rootdir_glo_neg) {
                    *zroot = true;
                    continue;
                }
            }
            if (ida_glo[i] * gval[i] < 0.0 && rootdir_glo_neg) {
                const double gfrac = std::fabs(gval[i] / (gval[i] - ida_glo[i]));
                if (gfrac > maxfrac) {
                    *sgnchg = true;
                    maxfrac = gfrac;
                    *imax = i;
                }
            }
        }
    }

    int root_find() {  // impl_r_check.rs:343-576
        int imax = 0;
        bool zroot, sgnchg;
        scan_roots(ida_ghi, true, &zroot, &sgnchg, &imax);

        if (!sgnchg) {
            ida_trout = ida_thi;
            ida_grout = ida_ghi;
            if (!zroot) return IDA_CONTINUE_STEPS;
            for (int i = 0; i < ida_nrtfn; ++i) {
                ida_iroots[i] = 0.0;
                if (ida_gactive[i]) {
                    const bool rootdir_glo_neg = (double)ida_rootdir[i] * ida_glo[i] <= 0.0;
                    if (std::fabs(ida_ghi[i]) == 0.0 && rootdir_glo_neg) ida_iroots[i] = signum(ida_glo[i]);
                }
            }
            return IDA_ROOT_RETURN;
        }

        double alph = 1.0;
        int side = 0, sideprev = -1;
        for (;;) {
            if (std::fabs(ida_thi - ida_tlo) <= ida_ttol) break;
            if (sideprev == side) {
                alph = (side == 2) ? alph * 2.0 : alph * 0.5;
            } else {
                alph = 1.0;
            }
            double tmid = ida_thi - (ida_thi - ida_tlo) * ida_ghi[imax] / (ida_ghi[imax] - alph * ida_glo[imax]);
            if (std::fabs(tmid - ida_tlo) < 0.5 * ida_ttol) {
                const double fracint = std::fabs(ida_thi - ida_tlo) / ida_ttol;
                const double fracsub = (fracint > 5.0) ? 0.1 : 0.5 / fracint;
                tmid = ida_tlo + fracsub * (ida_thi - ida_tlo);
            }
            if (std::fabs(ida_thi - tmid) < 0.5 * ida_ttol) {
                const double fracint = std::fabs(ida_thi - ida_tlo) / ida_ttol;
                const double fracsub = (fracint > 5.0) ? 0.1 : 0.5 / fracint;
                tmid = ida_thi - fracsub * (ida_thi - ida_tlo);
            }
            get_solution(tmid);
            problem->root(tmid, nlp.ida_yy.data(), nlp.ida_yp.data(), ida_grout.data());
            ida_nge += 1;
            sideprev = side;
            scan_roots(ida_grout, false, &zroot, &sgnchg, &imax);
            if (sgnchg) {
                ida_thi = tmid;
                ida_ghi = ida_grout;
                side = 1;
                if (std::fabs(ida_thi - ida_tlo) <= ida_ttol) break;
                continue;
            }
            if (zroot) {
                ida_thi = tmid;
                ida_ghi = ida_grout;
                break;
            }
            ida_tlo = tmid;
            ida_glo = ida_grout;
            side = 2;
            if (std::fabs(ida_thi - ida_tlo) <= ida_ttol) break;
        }
        ida_trout = ida_thi;
        ida_grout = ida_ghi;
        for (int i = 0; i < ida_nrtfn; ++i) {
            ida_iroots[i] = 0.0;
            if (ida_gactive[i]) {
                const bool rootdir_glo_neg = (double)ida_rootdir[i] * ida_glo[i] <= 0.0;
                if (rootdir_glo_neg && (std::fabs(ida_ghi[i]) == 0.0 || ida_glo[i] * ida_ghi[i] < 0.0))
                    ida_iroots[i] = signum(ida_glo[i]);
            }
        }
        return IDA_ROOT_RETURN;
    }
};

}  // namespace oracle
