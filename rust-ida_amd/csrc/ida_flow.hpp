// Ida::solve for ONE system as device code: the control flow of host/ensemble_ida.cpp (solve_core, attempt_round) on the shared
// scalar controller (host/ida_controller.hpp), with the vector work behind a backend V:
//   * TinyVec (tiny_ida.hpp):  one thread owns the system, vectors are loops of that thread (n <= 8);
//   * WgVec   (round_ida.hpp): one workgroup owns the system, every thread runs the scalar logic on its own copy of the state
//                              (uniform control flow) and the vector primitives are cooperative, their sums sequential.
// Mirrors   Ida::solve  /root/reference/src/impl_solve.rs:69-376 (first-call block, loop-top checks, stop tests),
//           Ida::step   /root/reference/src/lib.rs:613-711 (attempt loop), complete_step /root/reference/src/impl_complete_step.rs:22-177,
//           stop tests  /root/reference/src/impl_stop_test.rs:36-211 (IDA_NORMAL, no tstop), get_solution /root/reference/src/lib.rs:1274-1343.
// The Newton solve between attempt_begin() and attempt_end() is the caller's (in-thread for TinyVec, batched kernels for WgVec).
#pragma once
#include "glibc_pow.hpp"
#include "../host/ida_controller.hpp"
#include "../../include/ida_ensemble.h"
#include "common.hpp"

namespace idahip {

struct FlowArgs {
    const double* touts;  // [ntout] (device)
    int ntout;
    int recycle;          // idaens_stream: a system that finished its schedule starts over at once
    int resume;           // continuing a round-limited schedule call: idle systems have finished
    long mxstep;
    int maxord;
    long maxnef, maxncf;
    double epcon, hmax_inv, t0;
    const long long* start_round;  // [batch] or null: idaens_stream's staggered start (absolute round numbers)
    unsigned long long* acc;       // [2] retired Newton iterations, completed passes (idaens_stream)
    int batch;
};

// V provides: init_first(&ypnorm, &p0nrm), scale_phi1(f), predict(s), post_newton(s, norms[4]), restore_vec(s, kk_att, ns_att),
// complete_step_vec(s, kused, ck, maxord), get_solution_vec(s, kord), emit_output(slot), restore_initial()
template <class V>
struct IdaFlow {
    const FlowArgs& a;
    idactl::SysCore& s;
    V& v;

    // get_solution(t) into yy/yp; returns 0 or IDAENS_BAD_T
    __device__ int get_solution(double t) const {
        int kord = 1;
        const int rc = idactl::get_solution_coeffs(s, t, &kord);
        if (rc) return rc;
        v.get_solution_vec(s, kord);
        return 0;
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
            if (s.status == IDAENS_SUCCESS) v.emit_output(s.sched_i);
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
            v.init_first(&ypnorm, &p0nrm);
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
                v.scale_phi1(s.hh);  // phi[1] = hh * y'
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
    // a step attempt up to the Newton solve: step() prologue, set_coeffs, tn += hh, lsetup decision, prediction
    __device__ void attempt_begin() const {
        idactl::begin_attempt(s);
        v.predict(s);
    }
    // the rest of the attempt once the Newton solve has set s.nls_ret (ensemble_ida.cpp's attempt_round after
    // newton_solve_batched); true = the system steps on
    __device__ bool attempt_end() const {
        double norms[4];
        v.post_newton(s, norms);
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
            v.restore_vec(s, kk_att, ns_att);
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
                v.scale_phi1(s.rr);
            }
            return true;  // predict again
        }
        idactl::complete_step_scalars(s, err_k, err_km1, norms[3], a.maxord, a.hmax_inv);
        v.complete_step_vec(s, s.kused, s.ck, a.maxord);
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
    // what follows a round in idaens_stream (ensemble_ida.cpp: the `recycle` block of solve_core): a system that finished its
    // schedule is created anew (Ida::new) and starts over; a staggered system starts when its round has come. `ground` = the
    // number of rounds completed. `count`: exactly one thread per system adds to the totals. Returns the new stepping state.
    __device__ bool after_round_stream(bool stepping, long long ground, int b, bool count) const {
        if (!stepping && s.ph == idactl::PH_IDLE && !s.dead && s.setup_done && s.status == IDAENS_SUCCESS && s.sched_i == a.ntout - 1 && s.nst > 0) {
            if (count) {
                atomicAdd(&a.acc[0], (unsigned long long)s.niters);
                atomicAdd(&a.acc[1], 1ull);
            }
            s = idactl::SysCore();
            s.tn = a.t0;
            v.restore_initial();
            return start_system();
        }
        if (!stepping && a.start_round && a.start_round[b] == ground && s.ph == idactl::PH_IDLE && s.nst == 0 && !s.setup_done)
            return start_system();  // staggered start: this system's turn
        return stepping;
    }
    // how a system enters a call (ensemble_ida.cpp: the first block of solve_core)
    __device__ bool enter(long long ground, int b) const {
        if (s.ph != idactl::PH_IDLE) return true;  // left mid-flight by a round limit: resume
        if (a.recycle && a.start_round && a.start_round[b] > ground) return false;  // staggered start: not yet
        if (!a.resume) return start_system();
        return false;
    }
};

}  // namespace idahip
