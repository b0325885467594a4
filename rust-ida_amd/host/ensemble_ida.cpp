// libidaens.so -- host-side BDF stepper for an ensemble of independent IVPs (implementation of
// include/ida_ensemble.h). Mirrors, per system and with the same names, the scalar logic of the reference:
//   Ida::solve            /root/reference/src/impl_solve.rs:69-376
//   Ida::step             /root/reference/src/lib.rs:613-711
//   set_coeffs            /root/reference/src/lib.rs:722-782
//   nonlinear_solve       /root/reference/src/lib.rs:787-890      (lsetup decision, ss resets)
//   Newton::solve         /root/reference/crates/nonlinear/src/newton.rs:51-167   (as a batched state machine)
//   idaNlsConvTest        /root/reference/src/ida_nls.rs:218-266   (host libm pow)
//   test_error            /root/reference/src/lib.rs:967-1039      (decisions; the norms come from the device)
//   restore               /root/reference/src/lib.rs:1044-1083
//   handle_n_flag         /root/reference/src/lib.rs:1120-1244
//   complete_step         /root/reference/src/impl_complete_step.rs:22-177
//   get_solution          /root/reference/src/lib.rs:1274-1343      (coefficients; the sums run on the device)
//   get_dky               /root/reference/src/lib.rs:424-529        (coefficients; quirk Q9: C IDA's loop bound)
//   stop_test1/2          /root/reference/src/impl_stop_test.rs:36-211 (no tstop: the reference has no setter)
//   r_check1/2/3, root_finding  /root/reference/src/impl_r_check.rs:32-576 (scalar bracketing here, y(t) interpolated on the device)
// Vectors live on the device; this file talks to it only through include/ida_hip.h. The oracle is NOT used here.
//
// Lock-step execution: one "round" is one step attempt (set_coeffs -> predict -> Newton -> error test -> accept or
// restore) for every system that still has to reach tout; systems diverge freely in h, order, Newton count and
// lsetup timing, the device calls act on compacted index lists.
//
// Deviations from the reference text, identical to the oracle's: Q1 (jac at tn), Q2 (LU failure is a
// recoverable lsetup failure), Q3 (Newton breaks out on ConvergenceRecover with a current Jacobian), Q4 (Newton
// ConvergenceRecover is recoverable at step level), Q5 (reset() rescales phi[1] only). Constraints and tstop are out of
// scope (the reference has no setter for either).
//
// A system whose solve call ended in a fatal IdaError (too much work, error-test or convergence failures, ...) is `dead`:
// the reference's Ida::solve could be called again after such an error and would try to continue; here the status is
// sticky -- later calls skip the system and report the same status (include/ida_ensemble.h says so).
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <limits>
#include <chrono>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ida_ensemble.h"

#include "ida_controller.hpp"

namespace {

using namespace idactl;  // constants, SysCore and the scalar controller shared with the device-resident stepper

struct Sys : SysCore {
    // --- root finding (src/lib.rs:225-244, src/impl_r_check.rs); the vectors have nrtfn entries when roots are enabled
    std::vector<double> glo, ghi, grout, iroots;
    std::vector<uint8_t> gactive;
};

}  // namespace

struct idaens {
    idahip_ctx* ctx = nullptr;
    int n = 0, batch = 0;
    std::vector<Sys> sys;
    std::string err;
    long mxstep = MXSTEP_DEFAULT;
    int maxord = MAXORD_DEFAULT;
    long maxncf = MXNCF, maxnef = MXNEF;
    double epcon = EPCON, hmax_inv = 0.0;
    int64_t total_rounds = 0;
    int trace_sys = -1;
    std::vector<double> trace;
    bool sched_unfinished = false;  // the last idaens_solve_schedule call stopped at its round limit
    double t0 = 0.0;                // every system starts at tn = t0 (Sys default)
    int64_t retired_iters = 0, passes = 0;  // idaens_stream: Newton iterations / integrations of systems already restarted
    bool have_ic = false, streaming = false;
    bool pow_mismatch = false; // glibc_pow.hpp != this host's std::pow (checked at create): the device steppers stay off
    bool device_ctl = true;    // small device problems: the whole of Ida::solve in one launch (idahip_tiny_solve), no lock-step rounds
    bool fused_newton = true;  // first two Newton iterations and their convergence tests in one device call (idahip_newton_iter2)
    std::vector<int64_t> start_round;  // idaens_stream with a stagger: the round at which each system first enters
    // root functions g_i(t, y, y') = y[rt_comp[i]] - rt_thr[i] (the form of the reference's Roberts example,
    // src/sample_problems/roberts.rs: g0 = y0 - 1e-4, g1 = y2 - 0.01); nrtfn == 0: no root finding
    int nrtfn = 0;
    std::vector<int32_t> rt_comp;
    std::vector<double> rt_thr;
    idaens_root_fn rt_fn = nullptr;  // or any host function (idaens_set_root_fn)
    void* rt_user = nullptr;
    std::vector<double> hy, hyp;  // host copies of one system's y, y' for the root functions
    // scratch for list calls
    std::vector<int32_t> idx, ia, ib;
    std::vector<double> da, db, dc, dd;
};

namespace {

int efail(idaens* e, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    e->err = buf;
    return code;
}

// IDAENS_PROFILE=1 in the environment: wall time inside device-library calls vs in this file's own host logic, printed
// by idaens_destroy (development aid for the host/device balance of a round)
static double g_prof_dev = 0.0, g_prof_t0 = -1.0;
static const bool g_prof = std::getenv("IDAENS_PROFILE") != nullptr;
static double prof_now() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}
#define ENS_CALL(e, call)                                                                                   \
    do {                                                                                                    \
        const double t0__ = g_prof ? prof_now() : 0.0;                                                      \
        int rc__ = (call);                                                                                  \
        if (g_prof) g_prof_dev += prof_now() - t0__;                                                        \
        if (rc__ < 0) return efail((e), rc__, "%s failed (%d): %s", #call, rc__, idahip_last_error((e)->ctx)); \
    } while (0)

struct SolList {  // systems whose yy/yp must be interpolated by the device
    std::vector<int32_t> idx, kord;
    std::vector<double> cvals, dvals;
    void add(int b, const Sys& s, int kord_) {
        idx.push_back(b);
        kord.push_back(kord_);
        cvals.insert(cvals.end(), s.cvals, s.cvals + MXORDP1);
        dvals.insert(dvals.end(), s.dvals, s.dvals + MAXORD_DEFAULT);
    }
};

// get_solution(t) for system b: coefficients now, device sums deferred to the list. Returns 0 or IDAENS_BAD_T.
int queue_solution(Sys& s, int b, double t, SolList& sl) {
    int kord = 1;
    const int rc = get_solution_coeffs(s, t, &kord);
    if (rc) return rc;
    sl.add(b, s, kord);
    return 0;
}

// ---------------------------------------------------------------- stop tests (impl_stop_test.rs), tstop == None
int stop_test1(Sys& s, int b, double tout, int itask, SolList& sl) {
    if (itask == IDAENS_NORMAL) {
        if (tout == s.tretlast) {
            s.tretlast = tout;
            s.tret = tout;
            return IDAENS_SUCCESS;
        }
        if ((s.tn - tout) * s.hh >= 0.0) {
            const int ier = queue_solution(s, b, tout, sl);
            if (ier) return ier;
            s.tretlast = tout;
            s.tret = tout;
            return IDAENS_SUCCESS;
        }
        return IDAENS_UNFINISHED;  // ContinueSteps
    }
    if ((s.tn - s.tretlast) * s.hh > 0.0) {
        queue_solution(s, b, s.tn, sl);
        s.tretlast = s.tn;
        s.tret = s.tn;
        return IDAENS_SUCCESS;
    }
    return IDAENS_UNFINISHED;
}

int stop_test2(Sys& s, int b, double tout, int itask, SolList& sl) {
    if (itask == IDAENS_NORMAL) {
        if ((s.tn - tout) * s.hh >= 0.0) {
            s.tret = tout;
            s.tretlast = tout;
            queue_solution(s, b, tout, sl);
            return IDAENS_SUCCESS;
        }
        return IDAENS_UNFINISHED;
    }
    s.tret = s.tn;  // OneStep: yy/yp already hold y(tn)
    s.tretlast = s.tn;
    return IDAENS_SUCCESS;
}

int flush_solutions(idaens* e, SolList& sl) {
    if (sl.idx.empty()) return 0;
    ENS_CALL(e, idahip_get_solution(e->ctx, sl.kord.data(), sl.cvals.data(), sl.dvals.data(), sl.idx.data(), (int)sl.idx.size()));
    sl = SolList();
    return 0;
}

// ---------------------------------------------------------------- root finding (src/impl_r_check.rs:32-576)
// Roots are rare events of single systems: the bracketing runs system by system on the host, with the device
// interpolating y(t), y'(t) (idahip_get_solution) and the two vectors coming back for the root functions.
int root_fn(const idaens* e, int b, double t, const double* y, const double* yp, double* g) {
    if (e->rt_fn) return e->rt_fn(e->rt_user, b, t, y, yp, e->nrtfn, g) == 0 ? 0 : IDAENS_RTFUNC_FAIL;
    for (int i = 0; i < e->nrtfn; ++i) g[i] = y[e->rt_comp[i]] - e->rt_thr[i];
    return 0;
}

// get_solution(t) of system b on the device (yy, yp of that system become y(t), y'(t), lib.rs:1274), copy to e->hy, e->hyp
int interp_now(idaens* e, int b, double t) {
    Sys& s = e->sys[b];
    int kord = 1;
    const int rc = get_solution_coeffs(s, t, &kord);
    if (rc) return rc;
    const int32_t ib = b, ko = kord;
    ENS_CALL(e, idahip_get_solution(e->ctx, &ko, s.cvals, s.dvals, &ib, 1));
    ENS_CALL(e, idahip_download(e->ctx, IDAHIP_F_YY, b, 1, e->hy.data()));
    ENS_CALL(e, idahip_download(e->ctx, IDAHIP_F_YP, b, 1, e->hyp.data()));
    return 0;
}

// impl_r_check.rs:32-115 -- at the first call, before phi[1] is scaled by hh
int r_check1(idaens* e, int b) {
    const double eps = std::numeric_limits<double>::epsilon();
    Sys& s = e->sys[b];
    const int n = e->n, nr = e->nrtfn;
    std::fill(s.iroots.begin(), s.iroots.end(), 0.0);
    s.tlo = s.tn;
    s.ttol = (std::fabs(s.tn) + std::fabs(s.hh)) * eps * 100.0;
    ENS_CALL(e, idahip_download(e->ctx, IDAHIP_F_PHI0, b, 1, e->hy.data()));
    ENS_CALL(e, idahip_download(e->ctx, (idahip_field)(IDAHIP_F_PHI0 + 1), b, 1, e->hyp.data()));
    if (int rf = root_fn(e, b, s.tlo, e->hy.data(), e->hyp.data(), s.glo.data())) return rf;
    s.nge = 1;
    bool zroot = false;
    for (int i = 0; i < nr; ++i)
        if (std::fabs(s.glo[i]) == 0.0) {
            s.gactive[i] = 0;
            zroot = true;
        }
    if (zroot) {
        const double hratio = std::fmax(s.ttol / std::fabs(s.hh), 0.1);
        const double smallh = hratio * s.hh;
        for (int i = 0; i < n; ++i) e->hy[i] = e->hy[i] + smallh * e->hyp[i];  // yy = phi[0] + smallh * phi[1]
        ENS_CALL(e, idahip_upload(e->ctx, IDAHIP_F_YY, b, 1, e->hy.data()));
        if (int rf = root_fn(e, b, s.tlo + smallh, e->hy.data(), e->hyp.data(), s.ghi.data())) return rf;
        s.nge += 1;
        for (int i = 0; i < nr; ++i)
            if (!s.gactive[i] && std::fabs(s.ghi[i]) != 0.0) {
                s.gactive[i] = 1;
                s.glo[i] = s.ghi[i];
            }
    }
    return 0;
}

// impl_r_check.rs:117-219 -- on re-entry after a root return. Returns IDAENS_UNFINISHED (continue), ROOT_RETURN or < 0.
int r_check2(idaens* e, int b) {
    const double eps = std::numeric_limits<double>::epsilon();
    Sys& s = e->sys[b];
    const int n = e->n, nr = e->nrtfn;
    if (!s.irfnd) return IDAENS_UNFINISHED;
    int rc = interp_now(e, b, s.tlo);
    if (rc) return rc;
    if (int rf = root_fn(e, b, s.tlo, e->hy.data(), e->hyp.data(), s.glo.data())) return rf;
    s.nge += 1;
    std::fill(s.iroots.begin(), s.iroots.end(), 0.0);
    bool zroot = false;
    for (int i = 0; i < nr; ++i)
        if (s.gactive[i] && std::fabs(s.glo[i]) == 0.0) {
            zroot = true;
            s.iroots[i] = 1.0;
        }
    if (zroot) {
        s.ttol = (std::fabs(s.tn) + std::fabs(s.hh)) * eps * 100.0;
        const double smallh = s.ttol * signum(s.hh);
        const double tplus = s.tlo + smallh;
        if ((tplus - s.tn) * s.hh >= 0.0) {
            const double hratio = smallh / s.hh;
            std::vector<double> p1(n);
            ENS_CALL(e, idahip_download(e->ctx, (idahip_field)(IDAHIP_F_PHI0 + 1), b, 1, p1.data()));
            for (int i = 0; i < n; ++i) e->hy[i] = e->hy[i] + hratio * p1[i];  // yy += hratio * phi[1]
            ENS_CALL(e, idahip_upload(e->ctx, IDAHIP_F_YY, b, 1, e->hy.data()));
        } else {
            rc = interp_now(e, b, tplus);
            if (rc) return rc;
        }
        if (int rf = root_fn(e, b, tplus, e->hy.data(), e->hyp.data(), s.ghi.data())) return rf;
        s.nge += 1;
        bool zroot2 = false;
        for (int i = 0; i < nr; ++i) {
            if (!s.gactive[i]) continue;
            if (std::fabs(s.ghi[i]) == 0.0) {
                if (s.iroots[i] > 0.0) return IDAENS_CLOSE_ROOTS;
                zroot2 = true;
                s.iroots[i] = 1.0;
            } else if (s.iroots[i] > 0.0) {
                s.glo[i] = s.ghi[i];
            }
        }
        if (zroot2) return IDAENS_ROOT_RETURN;
    }
    return IDAENS_UNFINISHED;
}

void scan_roots(const idaens* e, const Sys& s, const std::vector<double>& gval, bool first, bool* zroot, bool* sgnchg, int* imax) {
    double maxfrac = 0.0;
    *zroot = false;
    *sgnchg = false;
    for (int i = 0; i < e->nrtfn; ++i) {
        if (!s.gactive[i]) continue;
        const bool rootdir_glo_neg = 0.0 * s.glo[i] <= 0.0;  // rootdir is 0 (no setter in the reference, lib.rs:372)
        if (first) {  // impl_r_check.rs:361-383
            if (std::fabs(gval[i]) == 0.0) {
                if (rootdir_glo_neg) *zroot = true;
                continue;
            }
        } else if (std::fabs(gval[i]) == 0.0 && rootdir_glo_neg) {  // impl_r_check.rs:486-504
            *zroot = true;
            continue;
        }
        if (s.glo[i] * gval[i] < 0.0 && rootdir_glo_neg) {
            const double gfrac = std::fabs(gval[i] / (gval[i] - s.glo[i]));
            if (gfrac > maxfrac) {
                *sgnchg = true;
                maxfrac = gfrac;
                *imax = i;
            }
        }
    }
}

// impl_r_check.rs:343-576 (modified secant / Illinois). Returns IDAENS_UNFINISHED (no root), ROOT_RETURN or < 0.
int root_find(idaens* e, int b) {
    Sys& s = e->sys[b];
    const int nr = e->nrtfn;
    int imax = 0;
    bool zroot, sgnchg;
    scan_roots(e, s, s.ghi, true, &zroot, &sgnchg, &imax);
    if (!sgnchg) {
        s.trout = s.thi;
        s.grout = s.ghi;
        if (!zroot) return IDAENS_UNFINISHED;
        for (int i = 0; i < nr; ++i) {
            s.iroots[i] = 0.0;
            if (s.gactive[i] && std::fabs(s.ghi[i]) == 0.0 && 0.0 * s.glo[i] <= 0.0) s.iroots[i] = signum(s.glo[i]);
        }
        return IDAENS_ROOT_RETURN;
    }
    double alph = 1.0;
    int side = 0, sideprev = -1;
    for (;;) {
        if (std::fabs(s.thi - s.tlo) <= s.ttol) break;
        if (sideprev == side) alph = (side == 2) ? alph * 2.0 : alph * 0.5;
        else alph = 1.0;
        double tmid = s.thi - (s.thi - s.tlo) * s.ghi[imax] / (s.ghi[imax] - alph * s.glo[imax]);
        if (std::fabs(tmid - s.tlo) < 0.5 * s.ttol) {
            const double fracint = std::fabs(s.thi - s.tlo) / s.ttol;
            const double fracsub = (fracint > 5.0) ? 0.1 : 0.5 / fracint;
            tmid = s.tlo + fracsub * (s.thi - s.tlo);
        }
        if (std::fabs(s.thi - tmid) < 0.5 * s.ttol) {
            const double fracint = std::fabs(s.thi - s.tlo) / s.ttol;
            const double fracsub = (fracint > 5.0) ? 0.1 : 0.5 / fracint;
            tmid = s.thi - fracsub * (s.thi - s.tlo);
        }
        const int rc = interp_now(e, b, tmid);
        if (rc) return rc;
        if (int rf = root_fn(e, b, tmid, e->hy.data(), e->hyp.data(), s.grout.data())) return rf;
        s.nge += 1;
        sideprev = side;
        scan_roots(e, s, s.grout, false, &zroot, &sgnchg, &imax);
        if (sgnchg) {
            s.thi = tmid;
            s.ghi = s.grout;
            side = 1;
            if (std::fabs(s.thi - s.tlo) <= s.ttol) break;
            continue;
        }
        if (zroot) {
            s.thi = tmid;
            s.ghi = s.grout;
            break;
        }
        s.tlo = tmid;
        s.glo = s.grout;
        side = 2;
        if (std::fabs(s.thi - s.tlo) <= s.ttol) break;
    }
    s.trout = s.thi;
    s.grout = s.ghi;
    for (int i = 0; i < nr; ++i) {
        s.iroots[i] = 0.0;
        if (s.gactive[i] && 0.0 * s.glo[i] <= 0.0 && (std::fabs(s.ghi[i]) == 0.0 || s.glo[i] * s.ghi[i] < 0.0))
            s.iroots[i] = signum(s.glo[i]);
    }
    return IDAENS_ROOT_RETURN;
}

// impl_r_check.rs:221-280 -- after a successful step. Returns IDAENS_UNFINISHED (no root), ROOT_RETURN or < 0.
int r_check3(idaens* e, int b) {
    const double eps = std::numeric_limits<double>::epsilon();
    Sys& s = e->sys[b];
    if (s.taskc == IDAENS_ONE_STEP) s.thi = s.tn;
    else s.thi = ((s.toutc - s.tn) * s.hh >= 0.0) ? s.tn : s.toutc;
    int rc = interp_now(e, b, s.thi);
    if (rc) return rc;
    if (int rf = root_fn(e, b, s.thi, e->hy.data(), e->hyp.data(), s.ghi.data())) return rf;
    s.nge += 1;
    s.ttol = (std::fabs(s.tn) + std::fabs(s.hh)) * eps * 100.0;
    const int ier = root_find(e, b);
    if (ier < 0) return ier;
    for (int i = 0; i < e->nrtfn; ++i)
        if (!s.gactive[i] && s.grout[i] != 0.0) s.gactive[i] = 1;
    s.tlo = s.trout;
    s.glo = s.grout;
    if (ier == IDAENS_ROOT_RETURN) {
        rc = interp_now(e, b, s.trout);
        if (rc) return rc;
    }
    return ier;
}

// ---------------------------------------------------------------- the batched Newton solve (newton.rs:51-167)
// `act`: systems taking a step attempt this round, with s.call_lsetup decided. Sets s.nls_ret.
int newton_solve_batched(idaens* e, const std::vector<int32_t>& act) {
    std::vector<Sys>& S = e->sys;
    std::vector<int32_t> R(act), I, C, L, P;
    std::vector<uint8_t> jbad(e->batch, 0);
    std::vector<double> tn, cj, sc, nrm;
    std::vector<int32_t> info;
    while (!R.empty() || !I.empty()) {
        if (!R.empty()) {
            // sys(y0), y <- y0 = 0 (newton.rs:73); the systems whose Newton solve then calls setup (call_lsetup) get both
            // in one device call, the others sys alone
            L.clear(); P.clear();
            for (int b : R) (S[b].call_lsetup ? L : P).push_back(b);
            if (!P.empty()) {
                tn.clear(); cj.clear();
                for (int b : P) { tn.push_back(S[b].tn); cj.push_back(S[b].cj); }
                ENS_CALL(e, idahip_nls_sys(e->ctx, tn.data(), cj.data(), 1, P.data(), (int)P.size()));
            }
            for (int b : R) S[b].nre += 1;
            if (!L.empty()) {
                tn.clear(); cj.clear();
                for (int b : L) { tn.push_back(S[b].tn); cj.push_back(S[b].cj); }
                info.assign(L.size(), 0);
                ENS_CALL(e, idahip_nls_sys_setup(e->ctx, tn.data(), cj.data(), 1, info.data(), L.data(), (int)L.size()));
                for (size_t q = 0; q < L.size(); ++q) {
                    Sys& s = S[L[q]];
                    after_lsetup(s, info[q]);  // idaNlsLSetup / idaLsSetup bookkeeping (ida_nls.rs:168-179, ida_ls.rs:250)
                }
            }
            for (int b : R) {
                Sys& s = S[b];
                if (s.call_lsetup && s.nls_ret == NLS_LSETUP_RECVR) {
                    s.nconvfails += 1;  // jcur is true: no retry (newton.rs:146-153 with Q3)
                    continue;
                }
                s.curiter = 0;
                I.push_back(b);
            }
            R.clear();
        }
        if (I.empty()) break;
        sc.clear();
        {   // idaLsSolve's bookkeeping around LSolver::solve (ida_ls.rs:316-418): the solver's type decides the tolerance it is
            // given (0 for a direct solver: idahip_ls_solve ignores it), the nli / ncfl counters and whether the correction is
            // scaled by 2 / (1 + cjratio) (:405-410). The library's solver is Direct with no iterations and no failures.
            const int lst = idahip_ls_type(e->ctx);
            const long nli_inc = idahip_ls_num_iters(e->ctx);
            // tol = sqrt(N) * eplifac for an iterative solver, 0 for a direct one (:323-329; eplifac = 0.05, :211). The fused
            // iteration kernel is LSolver::solve of the DIRECT solver and takes no tolerance: anything else has no kernel here.
            const double tol = lsolve_tol(lst, std::sqrt((double)e->n), EPLIFAC);
            if (lst != IDAHIP_LS_DIRECT || tol != 0.0) return efail(e, -3, "LSolverType %d (tol %g): only the dense direct solver is implemented", lst, tol);
            for (int b : I) sc.push_back(after_lsolve(S[b], lst, nli_inc, false) ? 2.0 / (1.0 + S[b].cjratio) : 1.0);
        }
        C.clear();
        // what follows a convergence test (newton.rs:109-153): converged / iterate again / ConvergenceRecover
        auto after_ctest = [&](int b, int ret, bool converged) {
            Sys& s = S[b];
            if (ret == NLS_SUCCESS && converged) {
                s.jcur = false;
                s.nls_ret = NLS_SUCCESS;
                return;
            }
            if (ret == NLS_SUCCESS) {
                s.curiter += 1;
                if (s.curiter >= MAXNLSIT) ret = NLS_CONV_RECVR;
            }
            if (ret == NLS_SUCCESS) {
                C.push_back(b);  // sys(y) then iterate again
                return;
            }
            // ConvergenceRecover
            if (!s.jcur) {
                s.nconvfails += 1;
                s.call_lsetup = true;
                jbad[b] = 1;
                R.push_back(b);
            } else {
                s.nconvfails += 1;
                s.nls_ret = NLS_CONV_RECVR;
            }
        };
        bool all_fresh = e->fused_newton;
        for (int b : I) all_fresh = all_fresh && S[b].curiter == 0;
        if (all_fresh) {
            // the first two iterations in one device call: both convergence tests are decided there (no powf needed for
            // m <= 1, ida_nls.rs:243-262); the scalar state is brought up to date here from the two norms it returns
            tn.clear(); cj.clear();
            std::vector<double> told, ssv, epsv;
            for (int b : I) {
                tn.push_back(S[b].tn); cj.push_back(S[b].cj);
                told.push_back(S[b].toldel); ssv.push_back(S[b].ss); epsv.push_back(S[b].eps_newt);
            }
            nrm.assign(2 * I.size(), 0.0);
            std::vector<int32_t> conv(I.size(), 0);
            ENS_CALL(e, idahip_newton_iter2(e->ctx, sc.data(), tn.data(), cj.data(), told.data(), ssv.data(), epsv.data(), nrm.data(),
                                            conv.data(), I.data(), (int)I.size()));
            for (size_t q = 0; q < I.size(); ++q) {
                const int b = I[q];
                Sys& s = S[b];
                s.niters += 1;
                s.oldnrm = nrm[2 * q];           // m = 0 (ida_nls.rs:244)
                if (conv[q] == 1) {
                    after_ctest(b, NLS_SUCCESS, true);
                    continue;
                }
                s.curiter = 1;                   // the device went on: sys(y), second iteration, test with m = 1
                s.nre += 1;
                s.niters += 1;
                const double rate = nrm[2 * q + 1] / s.oldnrm;  // powf(base, 1/1)
                if (conv[q] == 3) {
                    after_ctest(b, NLS_CONV_RECVR, false);
                    continue;
                }
                s.ss = rate / (1.0 - rate);
                after_ctest(b, NLS_SUCCESS, conv[q] == 2);
            }
        } else {
        nrm.assign(I.size(), 0.0);
        ENS_CALL(e, idahip_newton_iter(e->ctx, sc.data(), nrm.data(), I.data(), (int)I.size()));
        for (size_t q = 0; q < I.size(); ++q) {
            const int b = I[q];
            Sys& s = S[b];
            s.niters += 1;
            bool converged = false;
            const int ret = conv_test(s, nrm[q], &converged);  // idaNlsConvTest (ida_nls.rs:218-266)
            after_ctest(b, ret, converged);
        }
        }
        if (!C.empty()) {
            tn.clear(); cj.clear();
            for (int b : C) { tn.push_back(S[b].tn); cj.push_back(S[b].cj); }
            ENS_CALL(e, idahip_nls_sys(e->ctx, tn.data(), cj.data(), 0, C.data(), (int)C.size()));
            for (int b : C) S[b].nre += 1;
        }
        I.swap(C);
    }
    return 0;
}

// ---------------------------------------------------------------- one idaens_solve / idaens_solve_schedule call
// A schedule touts[0..ntout) means: for every system, Ida::solve(touts[0]), then Ida::solve(touts[1]), ... -- but a
// system that returns from one call enters the next at once instead of waiting for the slowest system of the batch, so
// the lock-step rounds stay full. Each return is processed exactly as the reference does (stop tests, interpolation
// to tout).
struct SolveCall {
    const double* touts = nullptr;
    int ntout = 1;
    int itask = IDAENS_NORMAL;
    SolList sl;                                 // interpolations queued for the device
    std::vector<std::pair<int, int>> reached;   // (system, schedule index) whose output is in sl
    double *hYout = nullptr, *hYPout = nullptr; // optional [ntout][batch][n]
    int32_t* hReached = nullptr;                // optional [batch]: returns with IDAENS_SUCCESS so far
    bool recycle = false;                       // idaens_stream: a system that finished the schedule starts over at once
};

// run the queued interpolations, hand the outputs of the touts reached since the last call to the caller
int emit_outputs(idaens* e, SolveCall& C) {
    int rc = flush_solutions(e, C.sl);
    if (rc) return rc;
    if (C.hYout || C.hYPout) {
        const size_t n = e->n, bn = (size_t)e->batch * n;
        for (const auto& r : C.reached) {
            if (C.hYout) ENS_CALL(e, idahip_download(e->ctx, IDAHIP_F_YY, r.first, 1, C.hYout + r.second * bn + r.first * n));
            if (C.hYPout) ENS_CALL(e, idahip_download(e->ctx, IDAHIP_F_YP, r.first, 1, C.hYPout + r.second * bn + r.first * n));
        }
    }
    C.reached.clear();
    return 0;
}

// ---------------------------------------------------------------- one lock-step round over `act`
int continue_schedule(idaens* e, SolveCall& C, int b);

int attempt_round(idaens* e, std::vector<int32_t>& act, SolveCall& C) {
    SolList& sl = C.sl;
    const int itask = C.itask;
    std::vector<Sys>& S = e->sys;
    const int na = (int)act.size();
    // --- step() prologue + set_coeffs + tn += hh (lib.rs:619-653)
    // A system the device lock-step stepper left inside an attempt whose Newton solve must start over with a linear setup
    // (newton_retry: round_ida.hpp's round_newton_ctl_kernel, "in the next round") has had its begin_attempt and its prediction:
    // here it only joins the Newton solve, with call_lsetup set (a round-limited device call followed by a host-stepper call).
    std::vector<int32_t> pred, kkns;
    std::vector<double> beta, gamma;
    for (int q = 0; q < na; ++q) {
        Sys& s = S[act[q]];
        if (s.newton_retry) {
            s.newton_retry = false;
            s.call_lsetup = true;
            continue;
        }
        begin_attempt(s);  // step() prologue, set_coeffs, tn += hh, lsetup decision (ida_controller.hpp)
        pred.push_back(act[q]);
        kkns.push_back(s.kk);
        kkns.push_back(s.ns);
        beta.insert(beta.end(), s.beta, s.beta + MXORDP1);
        gamma.insert(gamma.end(), s.gamma, s.gamma + MXORDP1);
    }
    if (!pred.empty()) ENS_CALL(e, idahip_predict(e->ctx, kkns.data(), beta.data(), gamma.data(), pred.data(), (int)pred.size()));

    // --- Newton
    int rc = newton_solve_batched(e, act);
    if (rc) return rc;

    // --- final yy/yp + error-test norms (lib.rs:845-849, 983-1004; impl_complete_step.rs:74-77)
    std::vector<double> cj(na), norms(4 * (size_t)na);
    std::vector<int32_t> kk(na);
    for (int q = 0; q < na; ++q) {
        cj[q] = S[act[q]].cj;
        kk[q] = S[act[q]].kk;
    }
    ENS_CALL(e, idahip_post_newton(e->ctx, cj.data(), kk.data(), norms.data(), act.data(), na));

    // --- decisions
    std::vector<int32_t> rest_idx, rest_kkns, done_idx, done_kused, reset_idx;
    std::vector<double> rest_cvals, done_ck, reset_fac;
    std::vector<int32_t> next;
    for (int q = 0; q < na; ++q) {
        const int b = act[q];
        Sys& s = S[b];
        int nflag = NFLAG_NONE;
        double err_k = 0.0, err_km1 = 0.0;
        if (s.nls_ret == NLS_SUCCESS) {
            if (!test_error(s, s.ck, &norms[4 * q], &err_k, &err_km1)) nflag = NFLAG_TEST_FAIL;
        } else if (s.nls_ret == NLS_CONV_RECVR) {
            nflag = NFLAG_CONV_RECVR;
        } else {
            nflag = NFLAG_LSETUP_RECVR;
        }
        if (nflag != NFLAG_NONE) {
            // restore (with the kk/ns/beta of this attempt), then handle_n_flag
            const int kk_att = s.kk, ns_att = s.ns;
            restore_scalars(s);
            if (ns_att <= kk_att) {
                rest_idx.push_back(b);
                rest_kkns.push_back(kk_att);
                rest_kkns.push_back(ns_att);
                rest_cvals.insert(rest_cvals.end(), s.cvals, s.cvals + MXORDP1);
            }
            const int kflag = handle_n_flag(s, nflag, err_k, err_km1, e->maxnef, e->maxncf);
            if (kflag != 0) {  // step failed for good: Ida::solve's failed-step path (impl_solve.rs:300-313)
                if (queue_solution(s, b, s.tn, sl) == 0) {
                    s.tret = s.tn;
                    s.tretlast = s.tn;
                }
                s.status = kflag;
                s.dead = true;
                s.ph = PH_IDLE;
                continue;
            }
            if (s.nst == 0) {  // reset(): psi[0] = hh; phi[1] *= rr  (Q5)
                s.psi[0] = s.hh;
                reset_idx.push_back(b);
                reset_fac.push_back(s.rr);
            }
            next.push_back(b);  // predict again
            continue;
        }
        // accepted: complete_step (scalars now, vectors below), ee *= ck
        complete_step_scalars(s, err_k, err_km1, norms[4 * q + 3], e->maxord, e->hmax_inv);
        done_idx.push_back(b);
        done_kused.push_back(s.kused);
        done_ck.push_back(s.ck);
        if (b == e->trace_sys) {
            e->trace.push_back(s.tn);
            e->trace.push_back(s.hused);
            e->trace.push_back((double)s.kused);
        }
    }
    if (!rest_idx.empty())
        ENS_CALL(e, idahip_restore(e->ctx, rest_kkns.data(), rest_cvals.data(), rest_idx.data(), (int)rest_idx.size()));
    if (!reset_idx.empty()) ENS_CALL(e, idahip_scale_phi1(e->ctx, reset_fac.data(), reset_idx.data(), (int)reset_idx.size()));
    if (!done_idx.empty()) {
        std::vector<double> p0(done_idx.size());
        std::vector<int32_t> bad(done_idx.size());
        ENS_CALL(e, idahip_complete_step(e->ctx, done_kused.data(), done_ck.data(), e->maxord, p0.data(), bad.data(), done_idx.data(),
                                         (int)done_idx.size()));
        for (size_t q = 0; q < done_idx.size(); ++q) {
            const int b = done_idx[q];
            Sys& s = S[b];
            s.phi0nrm = p0[q];
            s.ewt_bad = bad[q] != 0;
            s.nstloc += 1;
            s.ph = PH_LOOP_TOP;
            if (e->nrtfn > 0) {  // impl_solve.rs:343-356
                const int ier = r_check3(e, b);
                if (ier < 0) {
                    s.status = ier;
                    s.dead = true;
                    s.ph = PH_IDLE;
                    continue;
                }
                if (ier == IDAENS_ROOT_RETURN) {
                    s.irfnd = true;
                    s.tretlast = s.tlo;
                    s.tret = s.tlo;
                    s.status = IDAENS_ROOT_RETURN;
                    s.ph = PH_IDLE;
                    continue;
                }
            }
            const int istate = stop_test2(s, b, s.tout_cur, itask, sl);
            if (istate != IDAENS_UNFINISHED) {
                s.status = istate;
                s.ph = PH_IDLE;
                const int cs = continue_schedule(e, C, b);  // the next tout of the schedule, if there is one
                if (cs < 0) return cs;
                if (cs == 1) next.push_back(b);
            } else {
                next.push_back(b);
            }
        }
    }
    act.swap(next);
    return 0;
}

}  // namespace

extern "C" {

int idaens_create(idaens** out, idahip_ctx* ctx, const double* hYY0, const double* hYP0) {
    if (!out || !ctx || !hYY0 || !hYP0) return -1;
    idaens* e = new idaens();
    if (g_prof && g_prof_t0 < 0.0) g_prof_t0 = prof_now();
    e->ctx = ctx;
    e->n = idahip_n(ctx);
    e->batch = idahip_batch(ctx);
    e->fused_newton = idahip_kind(ctx) != IDAHIP_HOST_CALLBACK;  // a host residual cannot run between two device iterations
    e->sys.resize(e->batch);
    // Ida::new (lib.rs:291-293): phi[0] = yy0, phi[1] = yp0; yy/yp start as yy0/yp0 (ida_nls.rs:83-84)
    int rc = idahip_upload(ctx, IDAHIP_F_PHI0, 0, e->batch, hYY0);
    if (!rc) rc = idahip_upload(ctx, (idahip_field)(IDAHIP_F_PHI0 + 1), 0, e->batch, hYP0);
    if (!rc) rc = idahip_upload(ctx, IDAHIP_F_YY, 0, e->batch, hYY0);
    if (!rc) rc = idahip_upload(ctx, IDAHIP_F_YP, 0, e->batch, hYP0);
    if (rc) {
        delete e;
        return rc;
    }
    e->have_ic = idahip_snapshot_initial(ctx) == 0;  // for idaens_stream's restarts; costs two batch-sized vectors
    // The device steppers decide step sizes and orders with glibc_pow.hpp, a restatement of ONE libm's pow; the host stepper
    // (and the reference's f64::powf) use this host's std::pow. The two agree bit for bit on the hosts this was built for
    // (tests/test_glibc_pow.py); on another libm they need not, and the two steppers would then take different step
    // sequences without a sign. Checked once per process on the controller's argument ranges: on a mismatch the ensemble
    // stays on the host stepper and says so in its error text.
    // (the verdict is process-wide and guarded: ensembles may be created from several host threads. A self-check that could not
    // RUN -- idahip_pow_batch failed -- is not a verdict: this ensemble stays on the host stepper, says why, and the next
    // idaens_create tries again.)
    static std::mutex g_pow_mu;
    static int g_pow_ok = -1;  // -1: not known yet, 0: bits differ, 1: bit-identical
    bool could_not_run = false;
    int pow_ok;
    {
        std::lock_guard<std::mutex> lk(g_pow_mu);
        if (g_pow_ok < 0) {
            std::vector<double> x, y, r;
            unsigned long long z = 0x9E3779B97F4A7C15ull;
            auto u01 = [&]() { z ^= z << 13; z ^= z >> 7; z ^= z << 17; return (double)(z >> 11) / 9007199254740992.0; };
            for (int i = 0; i < 384; ++i) {
                const int k = 1 + i % 6;
                x.push_back(1.0e-4 + 3.0 * u01());                      // 2 err + 0.0001 (handle_n_flag, complete_step), delnrm / oldnrm (ctest)
                y.push_back((i & 1) ? -1.0 / (double)(k + 1) : 1.0 / (double)k);
            }
            r.resize(x.size());
            if (idahip_pow_batch(ctx, x.data(), y.data(), r.data(), x.size()) != 0) {
                could_not_run = true;
            } else {
                g_pow_ok = 1;
                for (size_t i = 0; i < x.size(); ++i) {
                    const double h = std::pow(x[i], y[i]);
                    if (std::memcmp(&h, &r[i], sizeof h) != 0) g_pow_ok = 0;
                }
            }
        }
        pow_ok = g_pow_ok;
    }
    if (could_not_run) {
        e->device_ctl = false;
        e->pow_mismatch = true;
        e->err = std::string("the device pow self-check could not run (idahip_pow_batch failed: ") + idahip_last_error(ctx) +
                 "): the device steppers are off for this ensemble (host stepper in use)";
    } else if (pow_ok == 0) {
        e->device_ctl = false;
        e->pow_mismatch = true;
        e->err = "the device pow does not reproduce this host's std::pow bit for bit: the device steppers are off (host stepper in use)";
    }
    *out = e;
    return 0;
}

int idaens_destroy(idaens* e) {
    if (g_prof && e) {
        const double tot = prof_now() - g_prof_t0;
        std::fprintf(stderr, "[idaens profile] rounds %lld: in device-library calls %.3f s, host logic + idle %.3f s (since the first create)\n",
                     (long long)e->total_rounds, g_prof_dev, tot - g_prof_dev);
    }
    delete e;
    return 0;
}

const char* idaens_last_error(const idaens* e) { return e ? e->err.c_str() : "null"; }

int idaens_set_max_num_steps(idaens* e, long mxstep) {
    if (!e || mxstep < 0) return -1;
    e->mxstep = mxstep;
    return 0;
}
int idaens_set_device_controller(idaens* e, int on) {
    if (!e) return -1;
    e->device_ctl = on != 0 && !e->pow_mismatch;
    return (on != 0 && e->pow_mismatch) ? 1 : 0;  // 1: refused (see idaens_last_error)
}
int idaens_set_fused_newton(idaens* e, int on) {
    if (!e) return -1;
    e->fused_newton = on != 0 && idahip_kind(e->ctx) != IDAHIP_HOST_CALLBACK;
    return 0;
}

int idaens_set_max_ord(idaens* e, int maxord) {
    if (!e || maxord < 1 || maxord > MAXORD_DEFAULT) return -1;
    e->maxord = maxord;
    return 0;
}

}  // extern "C"

namespace {

// Entry of one Ida::solve(s.tout_cur) call for a system that is between calls (impl_solve.rs:179-241): root checks and
// stop tests. Returns IDAENS_UNFINISHED when the system has to step, else the status this call returns with (tret set).
int enter_call(idaens* e, SolveCall& C, int b) {
    Sys& s = e->sys[b];
    SolList& sl = C.sl;
    const int itask = C.itask;
    const double tout = s.tout_cur;
    s.nstloc = 0;
    if (itask == IDAENS_NORMAL) s.toutc = tout;
    s.taskc = itask;
    if (s.nst > 0 && e->nrtfn > 0) {
        const double eps_ = std::numeric_limits<double>::epsilon();
        const bool irfndp = s.irfnd;
        int ier = r_check2(e, b);
        if (ier < 0) {
            s.dead = true;
            return ier;
        }
        if (ier == IDAENS_ROOT_RETURN) {
            s.tretlast = s.tlo;
            s.tret = s.tlo;
            return IDAENS_ROOT_RETURN;
        }
        const double troundoff = (std::fabs(s.tn) + std::fabs(s.hh)) * eps_ * 100.0;
        if (std::fabs(s.tn - s.tretlast) > troundoff) {
            ier = r_check3(e, b);
            if (ier < 0) {
                s.dead = true;
                return ier;
            }
            if (ier == IDAENS_UNFINISHED) {
                s.irfnd = false;
                if (itask == IDAENS_ONE_STEP && irfndp) {
                    s.tretlast = s.tn;
                    s.tret = s.tn;
                    queue_solution(s, b, s.tn, sl);
                    return IDAENS_SUCCESS;
                }
            } else {  // root found
                s.irfnd = true;
                s.tretlast = s.tlo;
                s.tret = s.tlo;
                return IDAENS_ROOT_RETURN;
            }
        }
    }
    if (s.nst > 0) {
        const int istate = stop_test1(s, b, tout, itask, sl);
        if (istate != IDAENS_UNFINISHED) {
            if (istate < 0) s.dead = true;
            return istate;
        }
    }
    return IDAENS_UNFINISHED;
}

// A system's call has just returned (s.status set, phase idle). With IDAENS_SUCCESS and touts left in the schedule it
// enters the next call at once. Returns 1 when the system is stepping again, 0 when it is done for this call, < 0 on a
// device failure.
int continue_schedule(idaens* e, SolveCall& C, int b) {
    Sys& s = e->sys[b];
    for (;;) {
        if (s.status == IDAENS_SUCCESS) C.reached.push_back({b, s.sched_i});
        if (s.status != IDAENS_SUCCESS || s.sched_i + 1 >= C.ntout) return 0;
        s.sched_i += 1;
        s.tout_cur = C.touts[s.sched_i];
        // the next call may interpolate again for this system (tn already past the next tout), or bracket a root through
        // yy/yp: the output of the call that just returned has to be out of the way first
        if (e->nrtfn > 0 || (s.tn - s.tout_cur) * s.hh >= 0.0) {
            const int rc = emit_outputs(e, C);
            if (rc) return rc;
        }
        const int ist = enter_call(e, C, b);
        if (ist == IDAENS_UNFINISHED) {
            s.ph = PH_LOOP_TOP;
            return 1;
        }
        s.status = ist;
    }
}

// Small systems with a device residual: the whole call runs on the device, one thread per IVP with its own time loop and the
// controller of ida_controller.hpp compiled for the device (idahip_tiny_solve). The host only moves the controller states.
// 1 = one thread per system (idahip_tiny_solve), 2 = lock-step rounds driven from the device (idahip_round_solve), 0 = host stepper
int device_ctl_applies(const idaens* e, const SolveCall& C) {
    const int k = idahip_kind(e->ctx);
    if (!e->device_ctl || C.itask != IDAENS_NORMAL || e->trace_sys >= 0) return 0;
    // root finding on the device: the function family of idaens_set_roots (not a user callback), not in idaens_stream
    if (e->nrtfn != 0 && (e->rt_fn != nullptr || e->nrtfn > IDAHIP_MAX_ROOTS || C.recycle)) return 0;
    if (e->n <= 8 && (k == IDAHIP_ROBERTS || k == IDAHIP_LORENZ63)) return 1;
    if (e->n > 8 && e->n <= 4096 && (k == IDAHIP_LINEAR_DENSE || k == IDAHIP_HEAT1D) && idahip_lu_variant(e->ctx) >= 4) return 2;
    return 0;
}

int solve_core_device(idaens* e, SolveCall& C, double* hTret, int32_t* hStatus, long max_rounds, int mode) {
    std::vector<Sys>& S = e->sys;
    const int batch = e->batch;
    const size_t n = e->n;
    const bool resume = C.ntout > 1 && e->sched_unfinished;
    auto reached_of = [&](const Sys& s) { return s.sched_i + ((s.ph == PH_IDLE && s.status == IDAENS_SUCCESS) ? 1 : 0); };
    std::vector<SysCore> st(batch);
    std::vector<int> prev(batch, 0);
    for (int b = 0; b < batch; ++b) {
        st[b] = S[b];
        if (resume) prev[b] = reached_of(S[b]);
    }
    idahip_tiny_call call;
    call.touts = C.touts;
    call.ntout = C.ntout;
    call.recycle = C.recycle ? 1 : 0;
    call.resume = resume ? 1 : 0;
    call.max_rounds = max_rounds;
    call.mxstep = e->mxstep;
    call.maxord = e->maxord;
    call.maxnef = e->maxnef;
    call.maxncf = e->maxncf;
    call.epcon = e->epcon;
    call.hmax_inv = e->hmax_inv;
    call.t0 = e->t0;
    call.start_round = (C.recycle && !e->start_round.empty()) ? e->start_round.data() : nullptr;
    call.round_base = e->total_rounds;
    // the systems' root state (Ida's ida_glo / ghi / grout / iroots / gactive) goes with the controller records
    std::vector<idahip_root_state> rstate;
    std::vector<int32_t> rcomp(e->rt_comp.begin(), e->rt_comp.end());
    call.nroots = e->nrtfn;
    call.root_comps = rcomp.data();
    call.root_thresholds = e->rt_thr.data();
    call.root_states = nullptr;
    if (e->nrtfn > 0) {
        rstate.resize(batch);
        for (int b = 0; b < batch; ++b)
            for (int i = 0; i < e->nrtfn; ++i) {
                rstate[b].glo[i] = S[b].glo[i]; rstate[b].ghi[i] = S[b].ghi[i]; rstate[b].grout[i] = S[b].grout[i];
                rstate[b].iroots[i] = S[b].iroots[i]; rstate[b].gactive[i] = S[b].gactive[i];
            }
        call.root_states = rstate.data();
    }
    std::vector<int64_t> rounds(batch, 0);
    uint64_t acc[2] = {0, 0};
    std::vector<double> yo, ypo;
    if (C.hYout) yo.resize((size_t)C.ntout * batch * n);
    if (C.hYPout) ypo.resize((size_t)C.ntout * batch * n);
    int64_t rounds_run = -1;
    if (mode == 1)
        ENS_CALL(e, idahip_tiny_solve(e->ctx, st.data(), sizeof(SysCore), &call, rounds.data(), acc, C.hYout ? yo.data() : nullptr,
                                      C.hYPout ? ypo.data() : nullptr));
    else
        ENS_CALL(e, idahip_round_solve(e->ctx, st.data(), sizeof(SysCore), &call, rounds.data(), acc, C.hYout ? yo.data() : nullptr,
                                       C.hYPout ? ypo.data() : nullptr, &rounds_run));
    int64_t rmax = 0;
    bool unfinished = false;
    for (int b = 0; b < batch; ++b) {
        static_cast<SysCore&>(S[b]) = st[b];
        for (int i = 0; i < e->nrtfn; ++i) {
            S[b].glo[i] = rstate[b].glo[i]; S[b].ghi[i] = rstate[b].ghi[i]; S[b].grout[i] = rstate[b].grout[i];
            S[b].iroots[i] = rstate[b].iroots[i]; S[b].gactive[i] = (uint8_t)rstate[b].gactive[i];
        }
        rmax = std::max(rmax, rounds[b]);
        Sys& s = S[b];
        const int now = reached_of(s);
        for (int i = prev[b]; i < now; ++i) {  // the outputs of the touts reached in this call
            const size_t off = ((size_t)i * batch + b) * n;
            if (C.hYout) std::memcpy(C.hYout + off, yo.data() + off, sizeof(double) * n);
            if (C.hYPout) std::memcpy(C.hYPout + off, ypo.data() + off, sizeof(double) * n);
        }
        if (C.hReached) C.hReached[b] = now;
        if (s.ph != PH_IDLE) {
            hStatus[b] = IDAENS_UNFINISHED;
            hTret[b] = s.tn;
            unfinished = true;
        } else {
            hStatus[b] = s.status;
            hTret[b] = s.tret;
        }
    }
    e->total_rounds += rounds_run >= 0 ? rounds_run : rmax;
    e->retired_iters += (int64_t)acc[0];
    e->passes += (int64_t)acc[1];
    e->sched_unfinished = C.ntout > 1 && unfinished;
    return 0;
}

int solve_core(idaens* e, SolveCall& C, double* hTret, int32_t* hStatus, long max_rounds) {
    if (const int mode = device_ctl_applies(e, C)) return solve_core_device(e, C, hTret, hStatus, max_rounds, mode);
    const double eps = std::numeric_limits<double>::epsilon();
    std::vector<Sys>& S = e->sys;
    SolList& sl = C.sl;
    const double tout = C.touts[0];  // the first call's tout sizes the initial step (impl_solve.rs:104-131)
    std::vector<int32_t> act;
    const bool resume = C.ntout > 1 && e->sched_unfinished;  // continuing a round-limited schedule call

    // (re)enter the schedule with the idle systems `cand`: first-call block for those that have not started, then the
    // entry of their first Ida::solve call; whoever has to step is appended to `act`
    auto start_systems = [&](const std::vector<int32_t>& cand) -> int {
    // ---- first-call block for systems that have not started (impl_solve.rs:84-173)
        {
            std::vector<int32_t> fresh;
            for (int b : cand)
                if (S[b].ph == PH_IDLE && S[b].nst == 0 && !S[b].setup_done && !S[b].dead) fresh.push_back(b);
            if (!fresh.empty()) {
                // initial_setup's ewt_set(phi[0]) (lib.rs:537-545), ||phi[1]|| for the h0 heuristic (impl_solve.rs:122-126)
                // and ||phi[0]|| for the first tolsf test (impl_solve.rs:289-295)
                std::vector<double> ypnorm(fresh.size()), p0nrm(fresh.size());
                ENS_CALL(e, idahip_init_first(e->ctx, ypnorm.data(), p0nrm.data(), fresh.data(), (int)fresh.size()));
                std::vector<int32_t> ok;
                std::vector<double> fac;
                for (size_t q = 0; q < fresh.size(); ++q) {
                    Sys& s = S[fresh[q]];
                    const double tdist = std::fabs(tout - s.tn);
                    const double troundoff = 2.0 * eps * (std::fabs(s.tn) + std::fabs(tout));
                    if (tdist == 0.0 || tdist < troundoff) {
                        s.status = IDAENS_ILL_INPUT;  // "tout too close to t0 to start integration"
                        s.tret = s.tn;
                        continue;
                    }
                    s.setup_done = true;
                    s.hh = s.hin;
                    if (s.hh == 0.0) {
                        s.hh = 0.001 * tdist;
                        if (ypnorm[q] > 2.0 / s.hh) s.hh = 0.5 / ypnorm[q];  // Q7 kept (impl_solve.rs:127)
                        if (tout < s.tn) s.hh = -s.hh;
                    }
                    const double rh = std::fabs(s.hh) * e->hmax_inv;
                    if (rh > 1.0) s.hh /= rh;
                    s.h0u = s.hh;
                    s.kk = 0;
                    s.kused = 0;
                    s.eps_newt = e->epcon;
                    s.toldel = 0.0001 * s.eps_newt;
                    s.phi0nrm = p0nrm[q];
                    ok.push_back(fresh[q]);
                    fac.push_back(s.hh);
                }
                if (e->nrtfn > 0)
                    for (size_t q = 0; q < ok.size();) {  // impl_solve.rs:157-159
                        const int rc1 = r_check1(e, ok[q]);
                        if (rc1 == IDAENS_RTFUNC_FAIL) {  // the user's root function failed for this system: it never starts
                            S[ok[q]].status = rc1;
                            S[ok[q]].tret = S[ok[q]].tn;
                            S[ok[q]].dead = true;
                            ok.erase(ok.begin() + q);
                            fac.erase(fac.begin() + q);
                            continue;
                        }
                        if (rc1) return rc1;
                        ++q;
                    }
                if (!ok.empty()) ENS_CALL(e, idahip_scale_phi1(e->ctx, fac.data(), ok.data(), (int)ok.size()));  // phi[1] = hh*y'
            }
        }

        // ---- per-system entry: stop tests for started systems, then collect who steps (impl_solve.rs:179-241)
        for (int b : cand) {
            Sys& s = S[b];
            if (s.dead || !s.setup_done) continue;  // earlier fatal error / ILL_INPUT at the first call: status is sticky
            s.sched_i = 0;
            s.tout_cur = C.touts[0];
            const int ist = enter_call(e, C, b);
            if (ist == IDAENS_UNFINISHED) {
                s.ph = PH_LOOP_TOP;
                act.push_back(b);
                continue;
            }
            s.status = ist;
            const int cs = continue_schedule(e, C, b);
            if (cs < 0) return cs;
            if (cs == 1) act.push_back(b);
        }
        return 0;
    };
    {
        std::vector<int32_t> cand;
        for (int b = 0; b < e->batch; ++b) {
            if (S[b].ph != PH_IDLE) act.push_back(b);  // left mid-flight by a round limit: resume
            else if (C.recycle && !e->start_round.empty() && e->start_round[b] > e->total_rounds) continue;  // staggered start
            else if (!resume) cand.push_back(b);       // (resume: idle systems finished in an earlier slice of the call)
        }
        const int rc0 = start_systems(cand);
        if (rc0) return rc0;
    }

    // ---- main loop
    long rounds = 0;
    while (!act.empty()) {
        if (max_rounds > 0 && rounds >= max_rounds) break;
        // loop-top checks for systems starting a new step (impl_solve.rs:246-297)
        std::vector<int32_t> go;
        for (int b : act) {
            Sys& s = S[b];
            if (s.ph == PH_LOOP_TOP) {
                if (e->mxstep > 0 && s.nstloc >= e->mxstep) {
                    s.tret = s.tn;
                    s.tretlast = s.tn;
                    s.status = IDAENS_TOO_MUCH_WORK;  // recoverable for the caller: the next solve call continues
                    s.ph = PH_IDLE;
                    continue;
                }
                if (s.nst > 0 && s.ewt_bad) {
                    queue_solution(s, b, s.tn, sl);
                    s.tret = s.tn;
                    s.tretlast = s.tn;
                    s.status = IDAENS_ILL_INPUT;
                    s.dead = true;
                    s.ph = PH_IDLE;
                    continue;
                }
                s.tolsf = eps * s.phi0nrm;
                if (s.tolsf > 1.0) {
                    s.tolsf *= 10.0;
                    s.tret = s.tn;
                    s.tretlast = s.tn;
                    if (s.nst > 0) queue_solution(s, b, s.tn, sl);
                    s.status = IDAENS_TOO_MUCH_ACC;
                    s.dead = true;
                    s.ph = PH_IDLE;
                    continue;
                }
            }
            go.push_back(b);
        }
        act.swap(go);
        if (act.empty()) break;
        int rc = attempt_round(e, act, C);
        if (rc) return rc;
        rounds += 1;
        e->total_rounds += 1;
        if (C.ntout > 1 || C.recycle) {  // outputs of the touts reached in this round, before the next round steps on
            rc = emit_outputs(e, C);
            if (rc) return rc;
        }
        if (C.recycle) {  // Ida::new again for the systems that finished their schedule in this round
            std::vector<int32_t> again;
            for (int b = 0; b < e->batch; ++b) {
                Sys& s = S[b];
                if (s.ph == PH_IDLE && !s.dead && s.setup_done && s.status == IDAENS_SUCCESS && s.sched_i == C.ntout - 1 && s.nst > 0) {
                    e->retired_iters += s.niters;
                    e->passes += 1;
                    s = Sys();
                    s.tn = e->t0;
                    again.push_back(b);
                }
            }
            if (!again.empty()) {
                ENS_CALL(e, idahip_restore_initial(e->ctx, again.data(), (int)again.size()));
                rc = start_systems(again);
                if (rc) return rc;
            }
            if (!e->start_round.empty()) {  // staggered start: the systems whose turn it is now
                std::vector<int32_t> late;
                for (int b = 0; b < e->batch; ++b)
                    if (e->start_round[b] == e->total_rounds && S[b].ph == PH_IDLE && S[b].nst == 0 && !S[b].setup_done) late.push_back(b);
                if (!late.empty()) {
                    rc = start_systems(late);
                    if (rc) return rc;
                }
            }
        }
    }
    int rc = emit_outputs(e, C);
    if (rc) return rc;
    bool unfinished = false;
    for (int b = 0; b < e->batch; ++b) {
        Sys& s = S[b];
        if (C.hReached) C.hReached[b] = s.sched_i + ((s.ph == PH_IDLE && s.status == IDAENS_SUCCESS) ? 1 : 0);
        if (s.ph != PH_IDLE) {
            hStatus[b] = IDAENS_UNFINISHED;
            hTret[b] = s.tn;
            unfinished = true;
        } else {
            hStatus[b] = s.status;
            hTret[b] = s.tret;
        }
    }
    e->sched_unfinished = C.ntout > 1 && unfinished;
    return 0;
}

}  // namespace

extern "C" {

int idaens_device_controller_active(const idaens* e) {
    if (!e) return -1;
    SolveCall C;
    C.itask = IDAENS_NORMAL;
    return device_ctl_applies(e, C);
}

int idaens_solve(idaens* e, double tout, int itask, double* hTret, int32_t* hStatus, long max_rounds) {
    if (!e || !hTret || !hStatus) return -1;
    SolveCall C;
    C.touts = &tout;
    C.ntout = 1;
    C.itask = itask;
    return solve_core(e, C, hTret, hStatus, max_rounds);
}

int idaens_stream(idaens* e, const double* touts, int ntout, long max_rounds, long stagger_rounds, int64_t* passes_done) {
    if (!e || !touts || ntout < 1 || max_rounds < 1 || stagger_rounds < 0) return -1;
    if (!e->streaming && stagger_rounds > 0) {  // first call: system b enters at round b * stagger / batch
        e->start_round.resize(e->batch);
        for (int b = 0; b < e->batch; ++b) e->start_round[b] = e->total_rounds + (int64_t)b * stagger_rounds / e->batch;
    }
    if (e->nrtfn > 0) return efail(e, -2, "idaens_stream does not combine with root finding");
    if (!e->have_ic) return efail(e, -2, "no snapshot of the initial conditions (idaens_create failed to take it)");
    SolveCall C;
    C.touts = touts;
    C.ntout = ntout;
    C.itask = IDAENS_NORMAL;
    C.recycle = true;
    std::vector<double> tret(e->batch);
    std::vector<int32_t> status(e->batch);
    e->sched_unfinished = e->total_rounds > 0 && e->streaming;  // later slices continue the systems in flight
    e->streaming = true;
    const int rc = solve_core(e, C, tret.data(), status.data(), max_rounds);
    if (passes_done) *passes_done = e->passes;
    for (int b = 0; b < e->batch; ++b)
        if (status[b] < 0) return efail(e, -5, "system %d failed with status %d while streaming", b, status[b]);
    return rc;
}

int idaens_solve_schedule(idaens* e, const double* touts, int ntout, double* hTret, int32_t* hStatus, int32_t* hReached,
                          double* hYout, double* hYPout, long max_rounds) {
    if (!e || !touts || ntout < 1 || !hTret || !hStatus) return -1;
    SolveCall C;
    C.touts = touts;
    C.ntout = ntout;
    C.itask = IDAENS_NORMAL;
    C.hReached = hReached;
    C.hYout = hYout;
    C.hYPout = hYPout;
    return solve_core(e, C, hTret, hStatus, max_rounds);
}

// ---- several ensembles side by side on one device (include/ida_ensemble.h): one host thread per ensemble, each on its own
// context and HIP stream. The ensembles share nothing; what they gain is the device's own scheduling -- while one group's
// round is in a part that leaves most of the chip idle (the serial chain of the panel kernels, the later Newton passes and
// their residuals for a few hundred systems, round begin / end), the other groups' launches fill it.
namespace {
int check_group(idaens* const* ens, int ngroups) {
    if (!ens || ngroups < 1) return -1;
    for (int g = 0; g < ngroups; ++g)
        if (!ens[g]) return -1;
    for (int g = 0; g < ngroups; ++g)
        for (int h = g + 1; h < ngroups; ++h)
            if (ens[g] == ens[h] || ens[g]->ctx == ens[h]->ctx)
                return efail(ens[0], -2, "the ensembles of a group need a context (and HIP stream) each: groups %d and %d share one", g, h);
    return 0;
}
int run_group(int ngroups, long offset_us, const std::function<int(int)>& one) {
    std::vector<int> rc(ngroups, 0);
    std::vector<std::thread> th;
    th.reserve(ngroups > 0 ? ngroups - 1 : 0);
    for (int g = 1; g < ngroups; ++g)
        th.emplace_back([&, g]() {
            if (offset_us > 0) std::this_thread::sleep_for(std::chrono::microseconds((long long)g * offset_us));
            rc[g] = one(g);
        });
    rc[0] = one(0);  // group 0 on the calling thread
    for (auto& t : th) t.join();
    int worst = 0;
    for (int g = 0; g < ngroups; ++g)
        if (rc[g] < worst || (worst == 0 && rc[g] != 0)) worst = rc[g];
    return worst;
}
}  // namespace

int idaens_stream_group(idaens* const* ens, int ngroups, const double* touts, int ntout, long max_rounds, long stagger_rounds, long offset_us,
                        int64_t* passes_done) {
    const int rc = check_group(ens, ngroups);
    if (rc) return rc;
    return run_group(ngroups, offset_us, [&](int g) {
        return idaens_stream(ens[g], touts, ntout, max_rounds, stagger_rounds, passes_done ? passes_done + g : nullptr);
    });
}

int idaens_solve_schedule_group(idaens* const* ens, int ngroups, const double* touts, int ntout, double* const* hTret, int32_t* const* hStatus,
                                int32_t* const* hReached, long max_rounds) {
    const int rc = check_group(ens, ngroups);
    if (rc) return rc;
    if (!hTret || !hStatus) return -1;
    for (int g = 0; g < ngroups; ++g)
        if (!hTret[g] || !hStatus[g]) return -1;
    return run_group(ngroups, 0, [&](int g) {
        return idaens_solve_schedule(ens[g], touts, ntout, hTret[g], hStatus[g], hReached ? hReached[g] : nullptr, nullptr, nullptr, max_rounds);
    });
}

namespace {
int install_roots(idaens* e, int nroots) {
    for (const Sys& s : e->sys)
        if (s.nst > 0 || s.setup_done) return efail(e, -2, "root functions must be set before the first solve call");
    e->nrtfn = nroots;
    e->hy.assign(e->n, 0.0);
    e->hyp.assign(e->n, 0.0);
    for (Sys& s : e->sys) {
        s.glo.assign(nroots, 0.0);
        s.ghi.assign(nroots, 0.0);
        s.grout.assign(nroots, 0.0);
        s.iroots.assign(nroots, 0.0);
        s.gactive.assign(nroots, 0);  // sic: false (lib.rs:373); r_check1/3 switch them on
    }
    return 0;
}
}  // namespace

int idaens_set_roots(idaens* e, int nroots, const int32_t* comps, const double* thresholds) {
    if (!e || nroots < 0 || (nroots > 0 && (!comps || !thresholds))) return -1;
    for (int i = 0; i < nroots; ++i)
        if (comps[i] < 0 || comps[i] >= e->n) return efail(e, -2, "root function %d: component %d outside 0..%d", i, comps[i], e->n - 1);
    const int rc = install_roots(e, nroots);
    if (rc) return rc;
    e->rt_fn = nullptr;
    e->rt_comp.assign(comps, comps + nroots);
    e->rt_thr.assign(thresholds, thresholds + nroots);
    return 0;
}

int idaens_set_root_fn(idaens* e, int nroots, idaens_root_fn fn, void* user) {
    if (!e || nroots < 0 || (nroots > 0 && !fn)) return -1;
    const int rc = install_roots(e, nroots);
    if (rc) return rc;
    e->rt_fn = nroots > 0 ? fn : nullptr;
    e->rt_user = user;
    e->rt_comp.clear();
    e->rt_thr.clear();
    return 0;
}

int idaens_get_roots(const idaens* e, int32_t* out) {
    if (!e || !out) return -1;
    for (int b = 0; b < e->batch; ++b)
        for (int i = 0; i < e->nrtfn; ++i) out[(size_t)b * e->nrtfn + i] = (int32_t)e->sys[b].iroots[i];
    return 0;
}

int idaens_get_counter(const idaens* e, int which, int64_t* out) {
    if (!e || !out) return -1;
    for (int b = 0; b < e->batch; ++b) {
        const Sys& s = e->sys[b];
        int64_t v = 0;
        switch (which) {
            case IDAENS_C_NST: v = s.nst; break;
            case IDAENS_C_NRE: v = s.nre; break;
            case IDAENS_C_NJE: v = s.nje; break;
            case IDAENS_C_NSETUPS: v = s.nsetups; break;
            case IDAENS_C_NNI: v = s.niters; break;  // Q11: nni is the Newton counter (ida_io.rs:84-88)
            case IDAENS_C_NETF: v = s.netf; break;
            case IDAENS_C_NCFN: v = s.ncfn; break;
            case IDAENS_C_NATTEMPTS: v = s.n_attempts; break;
            case IDAENS_C_NLS_NCONVFAILS: v = s.nconvfails; break;
            case IDAENS_C_KUSED: v = s.kused; break;
            case IDAENS_C_KK: v = s.kk; break;
            case IDAENS_C_NGE: v = s.nge; break;
            case IDAENS_C_NLUFAIL: v = s.nlufail; break;
            case IDAENS_C_NCONV_JCUR: v = s.nconv_jcur; break;
            case IDAENS_C_NFAIL_FIRST: v = s.nfail_first; break;
            case IDAENS_C_NLI: v = s.nli; break;
            case IDAENS_C_NCFL: v = s.ncfl; break;
            default: return -2;
        }
        out[b] = v;
    }
    return 0;
}

int idaens_get_real(const idaens* e, int which, double* out) {
    if (!e || !out) return -1;
    for (int b = 0; b < e->batch; ++b) {
        const Sys& s = e->sys[b];
        switch (which) {
            case IDAENS_R_TN: out[b] = s.tn; break;
            case IDAENS_R_HUSED: out[b] = s.hused; break;
            case IDAENS_R_HH: out[b] = s.hh; break;
            case IDAENS_R_H0U: out[b] = s.h0u; break;
            case IDAENS_R_TOLSF: out[b] = s.tolsf; break;
            default: return -2;
        }
    }
    return 0;
}

int idaens_get_yy(idaens* e, double* hYY) {
    if (!e) return -1;
    ENS_CALL(e, idahip_download(e->ctx, IDAHIP_F_YY, 0, e->batch, hYY));
    return 0;
}
int idaens_get_yp(idaens* e, double* hYP) {
    if (!e) return -1;
    ENS_CALL(e, idahip_download(e->ctx, IDAHIP_F_YP, 0, e->batch, hYP));
    return 0;
}

// IDAGetDky coefficients c_j^(k)(t) (lib.rs:464-508, recurrence; C IDA's loop bound, see ida_ensemble.h)
static int get_dky_coeffs(const Sys& s, double t, int k, double* cjk) {
    if (k < 0 || k > s.kused) return IDAENS_BAD_K;
    const double eps = std::numeric_limits<double>::epsilon();
    const double tfuzz = 100.0 * eps * (std::fabs(s.tn) + std::fabs(s.hh)) * signum(s.hh);
    const double tp = s.tn - s.hused - tfuzz;
    if ((t - tp) * s.hh < 0.0) return IDAENS_BAD_T;
    double cjk_1[MXORDP1] = {0.0};
    for (int j = 0; j < MXORDP1; ++j) cjk[j] = 0.0;
    const double delt = t - s.tn;
    double psij_1 = 0.0;
    for (int i = 0; i <= k; ++i) {
        const double scalar_i = (double)i;
        if (i == 0) {
            cjk[i] = 1.0;
        } else {
            cjk[i] = cjk[i - 1] * scalar_i / s.psi[i - 1];
            psij_1 = s.psi[i - 1];
        }
        for (int j = i + 1; j <= s.kused - k + i; ++j) {
            cjk[j] = (scalar_i * cjk_1[j - 1] + cjk[j - 1] * (delt + psij_1)) / s.psi[j - 1];
            psij_1 = s.psi[j - 1];
        }
        for (int j = i + 1; j <= s.kused - k + i; ++j) cjk_1[j] = cjk[j];
    }
    return 0;
}

int idaens_get_dky(idaens* e, double t, int k, double* hDky, int32_t* hStatus) {
    if (!e || !hDky || !hStatus) return -1;
    std::vector<int32_t> idx, k0, k1;
    std::vector<double> coef;
    for (int b = 0; b < e->batch; ++b) {
        double cjk[MXORDP1];
        hStatus[b] = get_dky_coeffs(e->sys[b], t, k, cjk);
        if (hStatus[b] != 0) continue;
        idx.push_back(b);
        k0.push_back(k);
        k1.push_back(e->sys[b].kused);
        coef.insert(coef.end(), cjk, cjk + MXORDP1);
    }
    if (idx.empty()) return 0;
    std::vector<double> out(idx.size() * (size_t)e->n);
    ENS_CALL(e, idahip_get_dky(e->ctx, k0.data(), k1.data(), coef.data(), out.data(), idx.data(), (int)idx.size()));
    for (size_t s = 0; s < idx.size(); ++s)
        std::memcpy(hDky + (size_t)idx[s] * e->n, out.data() + s * e->n, sizeof(double) * e->n);
    return 0;
}

int64_t idaens_total_newton_iters(const idaens* e) {
    int64_t t = 0;
    if (e) {
        t = e->retired_iters;
        for (const Sys& s : e->sys) t += s.niters;
    }
    return t;
}
int64_t idaens_total_rounds(const idaens* e) { return e ? e->total_rounds : 0; }

int idaens_trace_system(idaens* e, int sys) {
    if (!e || sys < -1 || sys >= e->batch) return -1;
    e->trace_sys = sys;
    e->trace.clear();
    return 0;
}
long idaens_trace_len(const idaens* e) { return e ? (long)(e->trace.size() / 3) : 0; }
int idaens_trace_get(const idaens* e, double* out) {
    if (!e || !out) return -1;
    for (size_t i = 0; i < e->trace.size(); ++i) out[i] = e->trace[i];
    return 0;
}

}  // extern "C"
