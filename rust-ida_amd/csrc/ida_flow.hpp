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
    // root functions g_i = y[rt_comp[i]] - rt_thr[i] (nrt == 0: no root finding); their per-system state travels with the caller
    int nrt = 0;
    int rt_comp[IDAHIP_MAX_ROOTS] = {0};
    double rt_thr[IDAHIP_MAX_ROOTS] = {0};
};

// V provides: init_first(&ypnorm, &p0nrm), scale_phi1(f), predict(s), post_newton(s, norms[4]), restore_vec(s, kk_att, ns_att),
// complete_step_vec(s, kused, ck, maxord), get_solution_vec(s, kord), emit_output(slot), restore_initial(), and for the root
// functions: sync() (the vector primitives' results are visible to every thread of the system), yy_at(i), phi_at(j, i),
// yy_from_phi01(f) (yy = phi[0] + f * phi[1]), yy_add_phi1(f) (yy += f * phi[1])
// ROOTS = false compiles the root finding out (the steppers' kernels are instantiated both ways: the bracketing code costs
// the no-roots kernels registers and scratch otherwise -- config 2: 57 -> 46 M iters/s with it compiled in).
template <class V, bool ROOTS = false>
struct IdaFlow {
    const FlowArgs& a;
    idactl::SysCore& s;
    V& v;
    idahip_root_state* rt = nullptr;  // this system's root state (a.nrt > 0)

    // ------------------------------------------------------------ root finding (src/impl_r_check.rs:32-576), as
    // host/ensemble_ida.cpp runs it per system on the host -- here the bracketing is device code too: the interpolation of
    // y(t) is the backend's get_solution_vec, the function family is evaluated in place, nothing crosses PCIe per evaluation.
    __device__ void root_fn(double* g) const {  // g_i = y[comp_i] - thr_i at the current yy (examples/roberts.rs:53-56)
        v.sync();
        for (int i = 0; i < a.nrt; ++i) g[i] = v.yy_at(a.rt_comp[i]) - a.rt_thr[i];
    }
    // impl_r_check.rs:32-115 -- at the first call, before phi[1] is scaled by hh
    __device__ void r_check1() const {
        const double eps = 2.220446049250313e-16;
        for (int i = 0; i < a.nrt; ++i) rt->iroots[i] = 0.0;
        s.tlo = s.tn;
        s.ttol = (fabs(s.tn) + fabs(s.hh)) * eps * 100.0;
        for (int i = 0; i < a.nrt; ++i) rt->glo[i] = v.phi_at(0, a.rt_comp[i]) - a.rt_thr[i];  // g(tlo, phi[0], phi[1])
        s.nge = 1;
        bool zroot = false;
        for (int i = 0; i < a.nrt; ++i)
            if (fabs(rt->glo[i]) == 0.0) {
                rt->gactive[i] = 0;
                zroot = true;
            }
        if (zroot) {
            const double hratio = fmax(s.ttol / fabs(s.hh), 0.1);
            const double smallh = hratio * s.hh;
            v.yy_from_phi01(smallh);  // yy = phi[0] + smallh * phi[1]
            root_fn(rt->ghi);
            s.nge += 1;
            for (int i = 0; i < a.nrt; ++i)
                if (!rt->gactive[i] && fabs(rt->ghi[i]) != 0.0) {
                    rt->gactive[i] = 1;
                    rt->glo[i] = rt->ghi[i];
                }
        }
    }
    // impl_r_check.rs:117-219 -- on re-entry after a root return. IDAENS_UNFINISHED (continue), ROOT_RETURN or < 0.
    __device__ int r_check2() const {
        const double eps = 2.220446049250313e-16;
        if (!s.irfnd) return IDAENS_UNFINISHED;
        int rc = get_solution(s.tlo);
        if (rc) return rc;
        root_fn(rt->glo);
        s.nge += 1;
        for (int i = 0; i < a.nrt; ++i) rt->iroots[i] = 0.0;
        bool zroot = false;
        for (int i = 0; i < a.nrt; ++i)
            if (rt->gactive[i] && fabs(rt->glo[i]) == 0.0) {
                zroot = true;
                rt->iroots[i] = 1.0;
            }
        if (zroot) {
            s.ttol = (fabs(s.tn) + fabs(s.hh)) * eps * 100.0;
            const double smallh = s.ttol * idactl::signum(s.hh);
            const double tplus = s.tlo + smallh;
            if ((tplus - s.tn) * s.hh >= 0.0) {
                const double hratio = smallh / s.hh;
                v.sync();
                v.yy_add_phi1(hratio);  // yy += hratio * phi[1]
            } else {
                rc = get_solution(tplus);
                if (rc) return rc;
            }
            root_fn(rt->ghi);
            s.nge += 1;
            bool zroot2 = false;
            for (int i = 0; i < a.nrt; ++i) {
                if (!rt->gactive[i]) continue;
                if (fabs(rt->ghi[i]) == 0.0) {
                    if (rt->iroots[i] > 0.0) return IDAENS_CLOSE_ROOTS;
                    zroot2 = true;
                    rt->iroots[i] = 1.0;
                } else if (rt->iroots[i] > 0.0) {
                    rt->glo[i] = rt->ghi[i];
                }
            }
            if (zroot2) return IDAENS_ROOT_RETURN;
        }
        return IDAENS_UNFINISHED;
    }
    __device__ void scan_roots(const double* gval, bool first, bool* zroot, bool* sgnchg, int* imax) const {
        double maxfrac = 0.0;
        *zroot = false;
        *sgnchg = false;
        for (int i = 0; i < a.nrt; ++i) {
            if (!rt->gactive[i]) continue;
            const bool rootdir_glo_neg = 0.0 * rt->glo[i] <= 0.0;  // rootdir is 0 (no setter in the reference, lib.rs:372)
            if (first) {  // impl_r_check.rs:361-383
                if (fabs(gval[i]) == 0.0) {
                    if (rootdir_glo_neg) *zroot = true;
                    continue;
                }
            } else if (fabs(gval[i]) == 0.0 && rootdir_glo_neg) {  // impl_r_check.rs:486-504
                *zroot = true;
                continue;
            }
            if (rt->glo[i] * gval[i] < 0.0 && rootdir_glo_neg) {
                const double gfrac = fabs(gval[i] / (gval[i] - rt->glo[i]));
                if (gfrac > maxfrac) {
                    *sgnchg = true;
                    maxfrac = gfrac;
                    *imax = i;
                }
            }
        }
    }
    // impl_r_check.rs:343-576 (modified secant / Illinois). IDAENS_UNFINISHED (no root), ROOT_RETURN or < 0.
    __device__ int root_find() const {
        const int nr = a.nrt;
        int imax = 0;
        bool zroot, sgnchg;
        scan_roots(rt->ghi, true, &zroot, &sgnchg, &imax);
        if (!sgnchg) {
            s.trout = s.thi;
            for (int i = 0; i < nr; ++i) rt->grout[i] = rt->ghi[i];
            if (!zroot) return IDAENS_UNFINISHED;
            for (int i = 0; i < nr; ++i) {
                rt->iroots[i] = 0.0;
                if (rt->gactive[i] && fabs(rt->ghi[i]) == 0.0 && 0.0 * rt->glo[i] <= 0.0) rt->iroots[i] = idactl::signum(rt->glo[i]);
            }
            return IDAENS_ROOT_RETURN;
        }
        double alph = 1.0;
        int side = 0, sideprev = -1;
        for (;;) {
            if (fabs(s.thi - s.tlo) <= s.ttol) break;
            if (sideprev == side) alph = (side == 2) ? alph * 2.0 : alph * 0.5;
            else alph = 1.0;
            double tmid = s.thi - (s.thi - s.tlo) * rt->ghi[imax] / (rt->ghi[imax] - alph * rt->glo[imax]);
            if (fabs(tmid - s.tlo) < 0.5 * s.ttol) {
                const double fracint = fabs(s.thi - s.tlo) / s.ttol;
                const double fracsub = (fracint > 5.0) ? 0.1 : 0.5 / fracint;
                tmid = s.tlo + fracsub * (s.thi - s.tlo);
            }
            if (fabs(s.thi - tmid) < 0.5 * s.ttol) {
                const double fracint = fabs(s.thi - s.tlo) / s.ttol;
                const double fracsub = (fracint > 5.0) ? 0.1 : 0.5 / fracint;
                tmid = s.thi - fracsub * (s.thi - s.tlo);
            }
            const int rc = get_solution(tmid);
            if (rc) return rc;
            root_fn(rt->grout);
            s.nge += 1;
            sideprev = side;
            scan_roots(rt->grout, false, &zroot, &sgnchg, &imax);
            if (sgnchg) {
                s.thi = tmid;
                for (int i = 0; i < nr; ++i) rt->ghi[i] = rt->grout[i];
                side = 1;
                if (fabs(s.thi - s.tlo) <= s.ttol) break;
                continue;
            }
            if (zroot) {
                s.thi = tmid;
                for (int i = 0; i < nr; ++i) rt->ghi[i] = rt->grout[i];
                break;
            }
            s.tlo = tmid;
            for (int i = 0; i < nr; ++i) rt->glo[i] = rt->grout[i];
            side = 2;
            if (fabs(s.thi - s.tlo) <= s.ttol) break;
        }
        s.trout = s.thi;
        for (int i = 0; i < nr; ++i) rt->grout[i] = rt->ghi[i];
        for (int i = 0; i < nr; ++i) {
            rt->iroots[i] = 0.0;
            if (rt->gactive[i] && 0.0 * rt->glo[i] <= 0.0 && (fabs(rt->ghi[i]) == 0.0 || rt->glo[i] * rt->ghi[i] < 0.0))
                rt->iroots[i] = idactl::signum(rt->glo[i]);
        }
        return IDAENS_ROOT_RETURN;
    }
    // impl_r_check.rs:221-280 -- after a successful step. IDAENS_UNFINISHED (no root), ROOT_RETURN or < 0.
    __device__ int r_check3() const {
        const double eps = 2.220446049250313e-16;
        if (s.taskc == IDAENS_ONE_STEP) s.thi = s.tn;
        else s.thi = ((s.toutc - s.tn) * s.hh >= 0.0) ? s.tn : s.toutc;
        int rc = get_solution(s.thi);
        if (rc) return rc;
        root_fn(rt->ghi);
        s.nge += 1;
        s.ttol = (fabs(s.tn) + fabs(s.hh)) * eps * 100.0;
        const int ier = root_find();
        if (ier < 0) return ier;
        for (int i = 0; i < a.nrt; ++i)
            if (!rt->gactive[i] && rt->grout[i] != 0.0) rt->gactive[i] = 1;
        s.tlo = s.trout;
        for (int i = 0; i < a.nrt; ++i) rt->glo[i] = rt->grout[i];
        if (ier == IDAENS_ROOT_RETURN) {
            rc = get_solution(s.trout);
            if (rc) return rc;
        }
        return ier;
    }

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
    // entry of one Ida::solve(s.tout_cur) call (impl_solve.rs:179-241)
    __device__ int enter_call() const {
        s.nstloc = 0;
        s.toutc = s.tout_cur;
        s.taskc = IDAENS_NORMAL;
        if (ROOTS && s.nst > 0 && a.nrt > 0) {  // impl_solve.rs:187-229
            const double eps = 2.220446049250313e-16;
            int ier = r_check2();
            if (ier < 0) {
                s.dead = true;
                return ier;
            }
            if (ier == IDAENS_ROOT_RETURN) {
                s.tretlast = s.tlo;
                s.tret = s.tlo;
                return IDAENS_ROOT_RETURN;
            }
            const double troundoff = (fabs(s.tn) + fabs(s.hh)) * eps * 100.0;
            if (fabs(s.tn - s.tretlast) > troundoff) {
                ier = r_check3();
                if (ier < 0) {
                    s.dead = true;
                    return ier;
                }
                if (ier == IDAENS_UNFINISHED) {
                    s.irfnd = false;
                } else {  // root found
                    s.irfnd = true;
                    s.tretlast = s.tlo;
                    s.tret = s.tlo;
                    return IDAENS_ROOT_RETURN;
                }
            }
        }
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
                if (ROOTS && a.nrt > 0) {  // impl_solve.rs:157-159
                    r_check1();
                    v.sync();
                }
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
        if (ROOTS && a.nrt > 0) {  // impl_solve.rs:343-356
            const int ier = r_check3();
            if (ier < 0) {
                s.status = ier;
                s.dead = true;
                s.ph = idactl::PH_IDLE;
                return false;
            }
            if (ier == IDAENS_ROOT_RETURN) {
                s.irfnd = true;
                s.tretlast = s.tlo;
                s.tret = s.tlo;
                s.status = IDAENS_ROOT_RETURN;
                s.ph = idactl::PH_IDLE;
                return false;
            }
        }
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
