// Device-resident BDF stepper for small systems (n <= TINY_N): the WHOLE of Ida::solve -- step-size and order controller
// included -- runs on the device, one thread per IVP, for as many step attempts as the caller allows. No host round trip per
// Newton iteration, per step attempt or per output time: one launch per idaens_solve / _schedule / _stream call.
// (before this, a lock-step round of config 2 was ~12 launches and ~5 host synchronisations of pure latency.)
//
// The scalar decisions are the SAME source as the host stepper's (host/ida_controller.hpp: set_coeffs, begin_attempt,
// conv_test, test_error, handle_n_flag, complete_step_scalars, get_solution_coeffs), compiled for the device with
// pow = glibc_pow::pow (glibc_pow.hpp: glibc's __pow_fma restated instruction for instruction), so h, order and every counter
// carry the reference's bits. The vector parts repeat the arithmetic of the batched kernels (vector_kernels.hpp,
// problem_kernels.hpp, solve_kernels.hpp, lu_kernels.hpp's tiny_getrf) element for element; with one thread per system every
// sum is sequential by construction. Control flow follows host/ensemble_ida.cpp (solve_core, attempt_round,
// newton_solve_batched) for one system:
//   Ida::solve            /root/reference/src/impl_solve.rs:69-376     (first-call block, loop-top checks, stop tests)
//   Ida::step             /root/reference/src/lib.rs:613-711
//   nonlinear_solve       /root/reference/src/lib.rs:787-890
//   Newton::solve         /root/reference/crates/nonlinear/src/newton.rs:51-167 (Q3: break on ConvergenceRecover with a current J)
//   IdaNLProblem          /root/reference/src/ida_nls.rs:118-266
//   complete_step         /root/reference/src/impl_complete_step.rs:22-177
// Not handled here (the host stepper keeps those cases): root finding, IDA_ONE_STEP, host-callback problems, per-step traces.
#pragma once
#include "glibc_pow.hpp"
#include "../host/ida_controller.hpp"
#include "../../include/ida_ensemble.h"
#include "lu_kernels.hpp"
#include "problem_kernels.hpp"
#include "solve_kernels.hpp"
#include "vector_kernels.hpp"

namespace idahip {

struct TinyIdaArgs {
    idactl::SysCore* sys;  // [batch] the controller state, uploaded before and downloaded after the launch
    VecState v;            // phi, yy, yp, yypredict, yppredict, ewt, ee, delta
    double* savres;
    double* lu;            // [batch][n*n] Jacobian, factored in place
    long long* piv;        // [batch][n]
    const double* params;
    int nparam;
    const double *ic_y, *ic_yp;  // initial conditions (idaens_stream's restarts)
    const double* touts;         // [ntout] (device)
    int ntout;
    int recycle;                 // idaens_stream: a system that finished its schedule starts over at once
    int resume;                  // continuing a round-limited schedule call: idle systems have finished
    long max_rounds;             // step attempts per system in this launch (0: until done)
    long mxstep;
    int maxord;
    long maxnef, maxncf;
    double epcon, hmax_inv, t0;
    const long long* start_round;  // [batch] or null: idaens_stream's staggered start
    long long round_base;          // rounds executed before this launch
    double *yout, *ypout;          // [ntout][batch][n] or null: y, y' at every tout reached
    long long* rounds_done;        // [batch] rounds this system took part in during this launch
    unsigned long long* acc;       // [2] retired Newton iterations, completed passes (idaens_stream)
    int batch;
};

template <int KIND>
struct TinyIda {
    const TinyIdaArgs& a;
    idactl::SysCore& s;
    const int b, n;
    const long vb;

    __device__ double& phi(int j, int i) const { return a.v.phi[j * a.v.phistride + vb + i]; }

    // initial_setup's ewt_set(phi[0]) and the two norms of the first call (vector_kernels.hpp: init_first_kernel)
    __device__ void init_first(double* ypnorm, double* p0nrm) const {
        double s1 = 0.0, s0 = 0.0;
        for (int i = 0; i < n; ++i) {
            const double y = phi(0, i);
            const double e = ewt_of(a.v, y, i);
            a.v.ewt[vb + i] = e;
            const double p = phi(1, i) * e;
            s1 = s1 + p * p;
            const double q = y * e;
            s0 = s0 + q * q;
        }
        *ypnorm = sqrt(s1 / (double)n);
        *p0nrm = sqrt(s0 / (double)n);
    }
    __device__ void scale_phi1(double f) const {
        for (int i = 0; i < n; ++i) phi(1, i) *= f;
    }
    // predict_kernel
    __device__ void predict() const {
        for (int i = 0; i < n; ++i) {
            double yyp = 0.0, ypp = 0.0;
            for (int j = 0; j <= s.kk; ++j) {
                double p = phi(j, i);
                if (j >= s.ns) {
                    p *= s.beta[j];
                    phi(j, i) = p;
                }
                yyp = yyp + p;
                if (j >= 1) ypp = ypp + s.gamma[j] * p;
            }
            a.v.yypredict[vb + i] = yyp;
            a.v.yppredict[vb + i] = ypp;
        }
    }
    // idaNlsResidual (tiny_sys_kernel)
    __device__ void nls_sys(bool reset_ee) const {
        double yy[TINY_N], yp[TINY_N], r[TINY_N];
        for (int i = 0; i < n; ++i) {
            double yc = a.v.ee[vb + i];
            if (reset_ee) {
                yc = 0.0;
                a.v.ee[vb + i] = 0.0;
            }
            yy[i] = a.v.yypredict[vb + i] + yc;
            yp[i] = a.v.yppredict[vb + i] + s.cj * yc;
            a.v.yy[vb + i] = yy[i];
            a.v.yp[vb + i] = yp[i];
        }
        if (KIND == IDAHIP_ROBERTS) roberts_res(yy, yp, r);
        else lorenz_res(a.params + (long)b * a.nparam, yy, yp, r);
        for (int i = 0; i < n; ++i) {
            a.v.delta[vb + i] = r[i];
            a.savres[vb + i] = r[i];
        }
        s.nre += 1;
    }
    // idaNlsLSetup: jac at the current yy (tiny_jac_kernel) + dense_get_rf in place (tiny_getrf); returns info
    __device__ int lsetup() const {
        double y[TINY_N], J[TINY_N * TINY_N];
        for (int i = 0; i < n; ++i) y[i] = a.v.yy[vb + i];
        if (KIND == IDAHIP_ROBERTS) roberts_jac(s.cj, y, J);
        else lorenz_jac(a.params + (long)b * a.nparam, s.cj, y, J);
        double* M = a.lu + (long)b * n * n;
        for (int e = 0; e < n * n; ++e) M[e] = J[e];
        int lperm[TINY_N];
        return tiny_getrf(M, n, a.piv + vb, lperm);
    }
    // one Newton iteration body (tiny_newton_iter_kernel); returns delnrm
    __device__ double newton_iter() const {
        double vv[TINY_N];
        for (int i = 0; i < n; ++i) vv[i] = -a.v.delta[vb + i];
        tiny_getrs(a.lu + (long)b * n * n, n, a.piv + vb, vv);
        const double sc = s.cjratio != 1.0 ? 2.0 / (1.0 + s.cjratio) : 1.0;  // ida_ls.rs:406-410
        double acc = 0.0;
        for (int i = 0; i < n; ++i) {
            const double d = vv[i] * sc;
            a.v.delta[vb + i] = d;
            a.v.ee[vb + i] = a.v.ee[vb + i] + d;
            const double p = d * a.v.ewt[vb + i];
            acc = acc + p * p;
        }
        return sqrt(acc / (double)n);
    }
    // Newton::solve for this attempt (newton.rs:51-167 as ensemble_ida.cpp's newton_solve_batched runs it for one system)
    __device__ void newton_solve() const {
        bool restart = true;
        for (;;) {
            if (restart) {
                nls_sys(true);  // sys(y0), y <- y0 = 0 (newton.rs:73)
                if (s.call_lsetup) {
                    idactl::after_lsetup(s, lsetup());
                    if (s.nls_ret == idactl::NLS_LSETUP_RECVR) {
                        s.nconvfails += 1;  // jcur is true: no retry (newton.rs:146-153 with Q3)
                        return;
                    }
                }
                s.curiter = 0;
                restart = false;
            }
            const double delnrm = newton_iter();
            s.niters += 1;
            bool converged = false;
            int ret = idactl::conv_test(s, delnrm, &converged);
            if (ret == idactl::NLS_SUCCESS && converged) {
                s.jcur = false;
                s.nls_ret = idactl::NLS_SUCCESS;
                return;
            }
            if (ret == idactl::NLS_SUCCESS) {
                s.curiter += 1;
                if (s.curiter >= idactl::MAXNLSIT) ret = idactl::NLS_CONV_RECVR;
            }
            if (ret == idactl::NLS_SUCCESS) {
                nls_sys(false);  // sys(y), then iterate again
                continue;
            }
            s.nconvfails += 1;  // ConvergenceRecover
            if (!s.jcur) {
                s.call_lsetup = true;
                restart = true;
                continue;
            }
            s.nls_ret = idactl::NLS_CONV_RECVR;
            return;
        }
    }
    // final yy/yp and the four error-test norms (post_newton_kernel)
    __device__ void post_newton(double* norms) const {
        const int kk = s.kk;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        for (int i = 0; i < n; ++i) {
            const double e = a.v.ee[vb + i];
            const double w = a.v.ewt[vb + i];
            a.v.yy[vb + i] = a.v.yypredict[vb + i] + e;
            a.v.yp[vb + i] = a.v.yppredict[vb + i] + s.cj * e;
            double p = e * w;
            s0 = s0 + p * p;
            double d = 0.0;
            if (kk > 1) {
                d = phi(kk, i) + e;
                p = d * w;
                s1 = s1 + p * p;
            }
            if (kk > 2) {
                d = d + phi(kk - 1, i);
                p = d * w;
                s2 = s2 + p * p;
            }
            if (kk + 1 < MXORDP1) {
                const double tmp = e - phi(kk + 1, i);
                p = tmp * w;
                s3 = s3 + p * p;
            }
        }
        norms[0] = sqrt(s0 / (double)n);
        norms[1] = sqrt(s1 / (double)n);
        norms[2] = sqrt(s2 / (double)n);
        norms[3] = sqrt(s3 / (double)n);
    }
    // restore_kernel with the kk / ns of the failed attempt (cvals from restore_scalars)
    __device__ void restore_vec(int kk_att, int ns_att) const {
        if (ns_att > kk_att) return;
        for (int i = 0; i < n; ++i)
            for (int j = ns_att; j <= kk_att; ++j) phi(j, i) *= s.cvals[j - ns_att];
    }
    // complete_step_kernel: the phi recurrence, ee *= ck, the new ewt and ||phi[0]||
    __device__ void complete_step_vec(int kused, double ck) const {
        bool bad = false;
        double acc = 0.0;
        for (int i = 0; i < n; ++i) {
            const double e = a.v.ee[vb + i];
            if (kused < a.maxord) phi(kused + 1, i) = e;
            double tmp = e;
            for (int j = kused; j >= 0; --j) {
                tmp = tmp + phi(j, i);
                phi(j, i) = tmp;
            }
            a.v.ee[vb + i] = e * ck;
            const double w = ewt_of(a.v, tmp, i);
            a.v.ewt[vb + i] = w;
            if (!(w > 0.0)) bad = true;
            const double p = tmp * w;
            acc = acc + p * p;
        }
        s.phi0nrm = sqrt(acc / (double)n);
        s.ewt_bad = bad;
    }
    // get_solution(t) into yy/yp (get_solution_kernel); returns 0 or IDAENS_BAD_T
    __device__ int get_solution(double t) const {
        int kord = 1;
        const int rc = idactl::get_solution_coeffs(s, t, &kord);
        if (rc) return rc;
        for (int i = 0; i < n; ++i) {
            double y = 0.0, yp = 0.0;
            for (int j = 0; j <= kord; ++j) {
                const double p = phi(j, i);
                y = y + s.cvals[j] * p;
                if (j >= 1) yp = yp + s.dvals[j - 1] * p;
            }
            a.v.yy[vb + i] = y;
            a.v.yp[vb + i] = yp;
        }
        return 0;
    }
    __device__ void emit_output() const {  // the output of the tout just reached (idaens_solve_schedule's hYout / hYPout)
        if (a.yout)
            for (int i = 0; i < n; ++i) a.yout[((long)s.sched_i * a.batch + b) * n + i] = a.v.yy[vb + i];
        if (a.ypout)
            for (int i = 0; i < n; ++i) a.ypout[((long)s.sched_i * a.batch + b) * n + i] = a.v.yp[vb + i];
    }
    // stop_test1 / stop_test2 in IDA_NORMAL mode without tstop (impl_stop_test.rs:36-211)
    __device__ int stop_test1(double tout) const {
        if (tout == s.tretlast) {
            s.tretlast = tout;
            s.tret = tout;
            return IDAENS_SUCCESS;
        }
        if ((s.tn - tout) * s.hh >= 0.0) {
            const int ier = get_solution(tout);
            if (ier) return ier;
            s.tretlast = tout;
            s.tret = tout;
            return IDAENS_SUCCESS;
        }
        return IDAENS_UNFINISHED;
    }
    __device__ int stop_test2(double tout) const {
        if ((s.tn - tout) * s.hh >= 0.0) {
            s.tret = tout;
            s.tretlast = tout;
            (void)get_solution(tout);
            return IDAENS_SUCCESS;
        }
        return IDAENS_UNFINISHED;
    }
    // entry of one Ida::solve(s.tout_cur) call (impl_solve.rs:179-241, no roots)
    __device__ int enter_call() const {
        s.nstloc = 0;
        s.toutc = s.tout_cur;
        s.taskc = IDAENS_NORMAL;
        if (s.nst > 0) {
            const int istate = stop_test1(s.tout_cur);
            if (istate != IDAENS_UNFINISHED) {
                if (istate < 0) s.dead = true;
                return istate;
            }
        }
        return IDAENS_UNFINISHED;
    }
    // the call has returned (s.status set, phase idle): with IDAENS_SUCCESS and touts left it enters the next call at once;
    // true = stepping again
    __device__ bool continue_schedule() const {
        for (;;) {
            if (s.status == IDAENS_SUCCESS) emit_output();
            if (s.status != IDAENS_SUCCESS || s.sched_i + 1 >= a.ntout) return false;
            s.sched_i += 1;
            s.tout_cur = a.touts[s.sched_i];
            const int ist = enter_call();
            if (ist == IDAENS_UNFINISHED) {
                s.ph = idactl::PH_LOOP_TOP;
                return true;
            }
            s.status = ist;
        }
    }
    // (re)enter the schedule: the first-call block for a system that has not started (impl_solve.rs:84-173), then the entry
    // of its first Ida::solve call; true = the system steps
    __device__ bool start_system() const {
        const double eps = 2.220446049250313e-16;
        const double tout = a.touts[0];
        if (s.ph == idactl::PH_IDLE && s.nst == 0 && !s.setup_done && !s.dead) {
            double ypnorm, p0nrm;
            init_first(&ypnorm, &p0nrm);
            const double tdist = fabs(tout - s.tn);
            const double troundoff = 2.0 * eps * (fabs(s.tn) + fabs(tout));
            if (tdist == 0.0 || tdist < troundoff) {
                s.status = IDAENS_ILL_INPUT;  // "tout too close to t0 to start integration"
                s.tret = s.tn;
            } else {
                s.setup_done = true;
                s.hh = s.hin;
                if (s.hh == 0.0) {
                    s.hh = 0.001 * tdist;
                    if (ypnorm > 2.0 / s.hh) s.hh = 0.5 / ypnorm;  // Q7 kept (impl_solve.rs:127)
                    if (tout < s.tn) s.hh = -s.hh;
                }
                const double rh = fabs(s.hh) * a.hmax_inv;
                if (rh > 1.0) s.hh /= rh;
                s.h0u = s.hh;
                s.kk = 0;
                s.kused = 0;
                s.eps_newt = a.epcon;
                s.toldel = 0.0001 * s.eps_newt;
                s.phi0nrm = p0nrm;
                scale_phi1(s.hh);  // phi[1] = hh * y'
            }
        }
        if (s.dead || !s.setup_done) return false;  // earlier fatal error / ILL_INPUT at the first call: status is sticky
        s.sched_i = 0;
        s.tout_cur = tout;
        const int ist = enter_call();
        if (ist == IDAENS_UNFINISHED) {
            s.ph = idactl::PH_LOOP_TOP;
            return true;
        }
        s.status = ist;
        return continue_schedule();
    }
    // loop-top checks of a new step (impl_solve.rs:246-297); false = the call returns
    __device__ bool loop_top() const {
        const double eps = 2.220446049250313e-16;
        if (a.mxstep > 0 && s.nstloc >= a.mxstep) {
            s.tret = s.tn;
            s.tretlast = s.tn;
            s.status = IDAENS_TOO_MUCH_WORK;  // recoverable for the caller: the next solve call continues
            s.ph = idactl::PH_IDLE;
            return false;
        }
        if (s.nst > 0 && s.ewt_bad) {
            (void)get_solution(s.tn);
            s.tret = s.tn;
            s.tretlast = s.tn;
            s.status = IDAENS_ILL_INPUT;
            s.dead = true;
            s.ph = idactl::PH_IDLE;
            return false;
        }
        s.tolsf = eps * s.phi0nrm;
        if (s.tolsf > 1.0) {
            s.tolsf *= 10.0;
            s.tret = s.tn;
            s.tretlast = s.tn;
            if (s.nst > 0) (void)get_solution(s.tn);
            s.status = IDAENS_TOO_MUCH_ACC;
            s.dead = true;
            s.ph = idactl::PH_IDLE;
            return false;
        }
        return true;
    }
    // one step attempt (ensemble_ida.cpp's attempt_round for one system); true = the system steps on
    __device__ bool attempt() const {
        idactl::begin_attempt(s);
        predict();
        newton_solve();
        double norms[4];
        post_newton(norms);
        int nflag = idactl::NFLAG_NONE;
        double err_k = 0.0, err_km1 = 0.0;
        if (s.nls_ret == idactl::NLS_SUCCESS) {
            if (!idactl::test_error(s, s.ck, norms, &err_k, &err_km1)) nflag = idactl::NFLAG_TEST_FAIL;
        } else if (s.nls_ret == idactl::NLS_CONV_RECVR) {
            nflag = idactl::NFLAG_CONV_RECVR;
        } else {
            nflag = idactl::NFLAG_LSETUP_RECVR;
        }
        if (nflag != idactl::NFLAG_NONE) {
            const int kk_att = s.kk, ns_att = s.ns;
            idactl::restore_scalars(s);
            restore_vec(kk_att, ns_att);
            const int kflag = idactl::handle_n_flag(s, nflag, err_k, err_km1, a.maxnef, a.maxncf);
            if (kflag != 0) {  // step failed for good: Ida::solve's failed-step path (impl_solve.rs:300-313)
                if (get_solution(s.tn) == 0) {
                    s.tret = s.tn;
                    s.tretlast = s.tn;
                }
                s.status = kflag;
                s.dead = true;
                s.ph = idactl::PH_IDLE;
                return false;
            }
            if (s.nst == 0) {  // reset(): psi[0] = hh; phi[1] *= rr  (Q5)
                s.psi[0] = s.hh;
                scale_phi1(s.rr);
            }
            return true;  // predict again
        }
        idactl::complete_step_scalars(s, err_k, err_km1, norms[3], a.maxord, a.hmax_inv);
        complete_step_vec(s.kused, s.ck);
        s.nstloc += 1;
        s.ph = idactl::PH_LOOP_TOP;
        const int istate = stop_test2(s.tout_cur);
        if (istate != IDAENS_UNFINISHED) {
            s.status = istate;
            s.ph = idactl::PH_IDLE;
            return continue_schedule();
        }
        return true;
    }
    // Ida::new again (idaens_stream): restore_initial_kernel + a fresh controller state
    __device__ void recycle() const {
        for (int i = 0; i < n; ++i) {
            const double y = a.ic_y[vb + i], yp = a.ic_yp[vb + i];
            phi(0, i) = y;
            phi(1, i) = yp;
            a.v.yy[vb + i] = y;
            a.v.yp[vb + i] = yp;
        }
    }
};

template <int KIND>
__global__ __launch_bounds__(64) void tiny_ida_kernel(TinyIdaArgs a) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.batch) return;
    idactl::SysCore s = a.sys[b];
    const TinyIda<KIND> T{a, s, b, a.v.n, (long)b * a.v.n};
    long long ground = a.round_base;  // global round counter (idaens_stream: every system takes part in every round)
    long long done = 0;
    bool stepping;
    if (s.ph != idactl::PH_IDLE) stepping = true;  // left mid-flight by a round limit: resume
    else if (a.recycle && a.start_round && a.start_round[b] > ground) stepping = false;  // staggered start: not yet
    else if (!a.resume) stepping = T.start_system();
    else stepping = false;
    for (;;) {
        if (a.max_rounds > 0 && done >= a.max_rounds) break;
        if (!stepping && !a.recycle) break;
        if (stepping) {
            if (s.ph == idactl::PH_LOOP_TOP && !T.loop_top()) {
                stepping = false;
                if (!a.recycle) break;  // (the call returned at the top of a round the system takes no part in)
            } else {
                stepping = T.attempt();
            }
        }
        done += 1;
        ground += 1;
        if (a.recycle) {
            if (!stepping && s.ph == idactl::PH_IDLE && !s.dead && s.setup_done && s.status == IDAENS_SUCCESS && s.sched_i == a.ntout - 1 &&
                s.nst > 0) {
                atomicAdd(&a.acc[0], (unsigned long long)s.niters);
                atomicAdd(&a.acc[1], 1ull);
                s = idactl::SysCore();
                s.tn = a.t0;
                T.recycle();
                stepping = T.start_system();
            } else if (!stepping && a.start_round && a.start_round[b] == ground && s.ph == idactl::PH_IDLE && s.nst == 0 && !s.setup_done) {
                stepping = T.start_system();  // staggered start: this system's turn
            } else if (!stepping && s.status < 0) {
                break;  // failed while streaming: the host reports it
            }
        }
    }
    a.sys[b] = s;
    a.rounds_done[b] = done;
}

}  // namespace idahip
