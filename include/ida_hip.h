/* ida_hip.h -- C ABI of libidahip.so: the MI355X (gfx950) implementation of rust-ida's BDF/Newton hot path for an
 * ensemble (batch) of independent IVPs.
 *
 * Every entry point is the batched generalisation of one trait method / function of the reference
 * (jondo2010/rust-ida, paths relative to the reference root); with batch = 1 and nsys = 1 a call reproduces the
 * reference call exactly. A Rust `impl LSolver / NLSolver / NLProblem / IdaProblem` forwards to these symbols
 * (binding sketch: INTEGRATION.md).
 *
 * Conventions
 *   - all floating point is fp64; no FMA contraction anywhere (the reference is plain Rust arithmetic);
 *   - matrices are column-major per system, A(i,j) = a[j*n + i] (nalgebra layout of crates/linear/src/dense.rs);
 *   - "h" pointers are host memory, borrowed for the duration of the call; "d" pointers are device memory;
 *   - hIdx[0..nsys) lists the systems a call acts on (the reference acts on one `Ida` at a time); per-call scalar
 *     arguments (hTn, hCj, ...) are indexed by list position, not by system id;
 *   - return value: 0 ok; > 0 recoverable (some listed system flagged, see the per-system output); < 0 fatal
 *     (bad argument, HIP error) -- the taxonomy documented at crates/nonlinear/src/traits.rs:17-22;
 *   - nothing panics, throws or aborts across this boundary; idahip_last_error() describes the last failure;
 *   - a ctx is not thread-safe (all reference methods take &mut self); different ctxs may be driven concurrently.
 */
#ifndef IDA_HIP_H
#define IDA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct idahip_ctx idahip_ctx; /* opaque; owns all device state of one ensemble on one device */

/* User problems shipped as device code (src/traits.rs:12-70 `Residual`/`Jacobian`; H5 in SURVEY.md: a generic Rust
 * closure cannot cross an FFI into device code, so the set is closed and selected by enum). */
typedef enum {
    IDAHIP_ROBERTS = 0,      /* src/sample_problems/roberts.rs:47-91                                  (n = 3) */
    IDAHIP_LORENZ63 = 1,     /* tests/lorenz63.rs:17-25,47-53; params [p, r, b] per system            (n = 3) */
    IDAHIP_LINEAR_DENSE = 2, /* F = A y' + B y - c, A/B dense column-major per system */
    IDAHIP_HEAT1D = 3,       /* 1-D heat, method of lines; params [kappa/dx^2] per system (config 4)           */
    IDAHIP_HOST_CALLBACK = 4 /* any IdaProblem: res / jac are host functions (idahip_set_host_problem); slow path   */
} idahip_problem;

/* The user problem of kind IDAHIP_HOST_CALLBACK: Residual::res and Jacobian::jac of src/traits.rs:12-70 as C callbacks, called on
 * the host for one listed system at a time (`sys` = its index in the batch, `user` = the pointer given at registration).
 *   res: resval[0..n) = F(tt, yy, yp)                                                       (traits.rs:28-37)
 *   jac: J = dF/dy + cj dF/dy', written column-major (J[j * n + i] = row i, column j) into a matrix the library has zeroed,
 *        as idaLsSetup does (src/ida_ls.rs:255); resvec is the residual at (yy, yp)        (traits.rs:58-69)
 * A nonzero return aborts the call that triggered it (it returns -7). Everything else -- batched LU, triangular solves,
 * norms, the stepper's vectors -- runs on the device as for the built-in problems; y, y' and the residual cross PCIe once
 * per evaluation, the Jacobian once per linear setup. */
typedef int (*idahip_res_fn)(int sys, double tt, const double* yy, const double* yp, double* resval, void* user);
typedef int (*idahip_jac_fn)(int sys, double tt, double cj, const double* yy, const double* yp, const double* resvec, double* J,
                             void* user);

/* ctx-resident vectors of the reference's IdaNLProblem / Ida structs (src/ida_nls.rs:27-59, src/lib.rs:104-126) */
typedef enum {
    IDAHIP_F_YY = 0,
    IDAHIP_F_YP = 1,
    IDAHIP_F_YYPREDICT = 2,
    IDAHIP_F_YPPREDICT = 3,
    IDAHIP_F_EWT = 4,
    IDAHIP_F_EE = 5,
    IDAHIP_F_DELTA = 6, /* Newton::delta (crates/nonlinear/src/newton.rs:21): residual in, update out */
    IDAHIP_F_SAVRES = 7,
    IDAHIP_F_PHI0 = 8, /* phi[j] = IDAHIP_F_PHI0 + j, j = 0..5 (MXORDP1 = 6, src/constants.rs:6) */
    IDAHIP_F_PHI5 = 13
} idahip_field;

/* ---- lifetime: LS::new(n) + NLS::new(n, maxiters) + IdaNLProblem::new (src/lib.rs:399-400, src/ida_ls.rs:192) ---- */
int idahip_create(idahip_ctx** ctx, int device, int n, int batch, idahip_problem kind, void* hip_stream /* or NULL */);
int idahip_destroy(idahip_ctx* ctx);
/* `count` HIP streams on `device` for contexts that are to work SIDE BY SIDE (idaens_stream_group, ida_ensemble.h). The HIP
 * runtime maps its streams onto a few hardware queues as it sees fit, and two streams on one queue take turns; this call
 * creates streams and keeps those a probe kernel shows to run concurrently with every stream kept before. streams_out[count]
 * receives hipStream_t handles for idahip_create's hip_stream argument (the caller keeps ownership: idahip_release_streams
 * after the contexts are destroyed); *nconcurrent (optional) = how many of them, from the front, are mutually concurrent --
 * less than count when the runtime offers fewer hardware queues (GPU_MAX_HW_QUEUES, 4 by default). */
int idahip_concurrent_streams(int device, int count, void** streams_out, int* nconcurrent);
int idahip_release_streams(int device, int count, void** streams);
/* Diagnostic: how evenly the device shares itself between two streams whose kernels each want the whole chip -- *frac = the part
 * of two probe grids' joint span in which both had workgroups running (about 1: the dispatcher interleaves them; about 0.5: the
 * second grid's workgroups only start when the first's are all out). */
int idahip_stream_pair_share(int device, void* streamA, void* streamB, double* frac);
const char* idahip_last_error(const idahip_ctx* ctx);
int idahip_sync(idahip_ctx* ctx);
int idahip_n(const idahip_ctx* ctx);
int idahip_batch(const idahip_ctx* ctx);
int idahip_kind(const idahip_ctx* ctx); /* the idahip_problem the ctx was created with */

/* TolControlSS (natol == 1) / TolControlSV (natol == n), src/tol_control.rs:6-82; shared by the ensemble */
int idahip_set_tolerances(idahip_ctx* ctx, double rtol, const double* hAtol, int natol);
/* per-system problem parameters, systems [first, first+count): LORENZ63 [count][3], HEAT1D [count][1] */
int idahip_set_problem_params(idahip_ctx* ctx, int first, int count, const double* hParams, int nparam);
/* LINEAR_DENSE data, systems [first, first+count): hA, hB [count][n*n] column-major, hC [count][n] */
int idahip_set_linear_dense(idahip_ctx* ctx, int first, int count, const double* hA, const double* hB, const double* hC);
/* IDAHIP_HOST_CALLBACK: the problem's callbacks (both required) and the pointer handed back to them */
int idahip_set_host_problem(idahip_ctx* ctx, idahip_res_fn res, idahip_jac_fn jac, void* user);
/* state movement: field of systems [first, first+count) <-> host [count][n] */
int idahip_upload(idahip_ctx* ctx, idahip_field f, int first, int count, const double* h);
int idahip_download(idahip_ctx* ctx, idahip_field f, int first, int count, double* h);
/* factored Jacobian of one system (column-major n*n), its pivots and the zero-pivot flag -- for tests/inspection */
int idahip_download_lu(idahip_ctx* ctx, int sys, double* hLU, int64_t* hPiv);

/* raw device memory for the stand-alone solver calls below (so callers need no HIP runtime of their own) */
void* idahip_dev_alloc(idahip_ctx* ctx, size_t bytes);
int idahip_dev_free(idahip_ctx* ctx, void* d);
int idahip_memcpy_h2d(idahip_ctx* ctx, void* d, const void* h, size_t bytes);
int idahip_memcpy_d2h(idahip_ctx* ctx, void* h, const void* d, size_t bytes);

/* ---- LSolver trait (crates/linear/src/traits.rs:27-91) on caller-owned device buffers [batch][n*n], [batch][n] ----
 * setup  = Dense::setup  -> dense_get_rf (crates/linear/src/dense.rs:38-44, 86-158): in-place PA = LU, pivots out;
 *          hInfo[s] = 0 | 1-based zero-pivot column (linear::Error::LUFactFail{col}, crates/linear/src/lib.rs:11-12).
 * solve  = Dense::solve  -> x <- b; dense_get_rs (dense.rs:46-63, 165-206). `tol` is ignored by a direct solver.   */
int idahip_ls_setup(idahip_ctx* ctx, double* dA, int64_t* dPiv, int32_t* hInfo, const int32_t* hIdx, int nsys);
int idahip_ls_solve(idahip_ctx* ctx, const double* dLU, const int64_t* dPiv, double* dX, const double* dB, double tol,
                    const int32_t* hIdx, int nsys);
/* LSolver::get_type (crates/linear/src/traits.rs:36-38, LSolverType: crates/linear/src/lib.rs:15-20): what idaLsSolve branches on
 * (src/ida_ls.rs:316-329, 387-410: the tolerance handed to the solver, which vector is copied to b, whether the 2/(1+cjratio)
 * scaling applies, the nli / ncfl counters). This library's solver is the dense direct one: IDAHIP_LS_DIRECT, no iterations,
 * no residual norm (Dense::num_iters / res_norm, traits.rs:82-90); the host stepper's bookkeeping takes the type from here. */
typedef enum { IDAHIP_LS_DIRECT = 0, IDAHIP_LS_ITERATIVE = 1, IDAHIP_LS_MATRIX_ITERATIVE = 2 } idahip_ls_kind;
int idahip_ls_type(const idahip_ctx* ctx);
int idahip_ls_num_iters(const idahip_ctx* ctx);
double idahip_ls_res_norm(const idahip_ctx* ctx);
/* NormRms::norm_wrms (src/norm_rms.rs:31-38): hOut[s] = sqrt(sum_i (x_i w_i)^2 / n), summed left to right */
int idahip_wrms(idahip_ctx* ctx, const double* dX, const double* dW, double* hOut, const int32_t* hIdx, int nsys);

/* IdaNLProblem::sys immediately followed by IdaNLProblem::setup for the same systems -- the order Newton::solve runs them
 * in when call_lsetup is set (crates/nonlinear/src/newton.rs:73-96; src/ida_nls.rs:118-188). Same results as
 * idahip_nls_sys + idahip_nls_lsetup; for the linear dense problem the residual pass also forms J = B + cj*A, so A and B are
 * read once instead of twice. hInfo / return value as idahip_nls_lsetup. */
int idahip_nls_sys_setup(idahip_ctx* ctx, const double* hTn, const double* hCj, int reset_ee, int32_t* hInfo,
                         const int32_t* hIdx, int nsys);

/* ---- NLProblem trait as implemented by IdaNLProblem (src/ida_nls.rs:118-266), on the ctx-resident state ----
 * sys    = idaNlsResidual (:118-153): yy = yypredict + ycor; yp = yppredict + cj*ycor; delta = savres = F(tn,yy,yp).
 *          ycor is the accumulated correction `ee` (Newton's y); reset_ee != 0 first sets ee = 0 (Newton's y <- y0 = 0,
 *          crates/nonlinear/src/newton.rs:75,93).
 * lsetup = idaNlsLSetup (:156-187) + idaLsSetup (src/ida_ls.rs:232-290): J <- 0; jac(tn, cj, yy, yp, res); LU(J).
 *          hInfo[s] = 0 | 1-based zero-pivot column (recoverable). The caller resets cjold/cjratio/ss as :177-179.
 * newton_iter = one pass of the Newton loop body (newton.rs:98-110): delta = -delta; idaNlsLSolve (src/ida_nls.rs:190,
 *          src/ida_ls.rs:298-455: getrs, then delta *= hScale[s] with hScale = 2/(1+cjratio), or 1.0 when cjratio == 1);
 *          ee += delta; hDelnrm[s] = ||delta||_wrms(ewt) for idaNlsConvTest (:218-266), which the host evaluates
 *          (its `powf` must be the platform libm's).                                                   */
int idahip_nls_sys(idahip_ctx* ctx, const double* hTn, const double* hCj, int reset_ee, const int32_t* hIdx, int nsys);
int idahip_nls_lsetup(idahip_ctx* ctx, const double* hTn, const double* hCj, int32_t* hInfo, const int32_t* hIdx, int nsys);
int idahip_newton_iter(idahip_ctx* ctx, const double* hScale, double* hDelnrm, const int32_t* hIdx, int nsys);
/* The first two iterations of Newton::solve in one call, without the host in between (first slice):
 *   newton body (m = 0) -> idaNlsConvTest -> NLProblem::sys -> newton body (m = 1) -> idaNlsConvTest
 * for the listed systems, each of which must be at curiter = 0 with its residual in `delta` (i.e. right after idahip_nls_sys
 * or idahip_nls_sys_setup). The two tests need no powf (m = 0: two comparisons; m = 1: rate = delnrm / oldnrm exactly,
 * src/ida_nls.rs:243-262), so they are decided on the device with the caller's per-system hToldel, hSs, hEpsNewt; a system
 * that has ended is skipped by the kernels that follow. Out: hDelnrm[s][0..2) = the norms of the two corrections (the second
 * is 0 if not run), hConv[s] = 0 go on with m = 2 (NLProblem::sys first) | 1 converged at m = 0 | 2 converged at m = 1 |
 * 3 ConvergenceRecover at m = 1. The caller updates oldnrm = hDelnrm[s][0] and, if the second iteration ran and did not end
 * in 3, ss = rate / (1 - rate) with rate = hDelnrm[s][1] / hDelnrm[s][0]. Not for IDAHIP_HOST_CALLBACK. */
int idahip_newton_iter2(idahip_ctx* ctx, const double* hScale, const double* hTn, const double* hCj, const double* hToldel,
                        const double* hSs, const double* hEpsNewt, double* hDelnrm, int32_t* hConv, const int32_t* hIdx, int nsys);

/* ---- vector parts of the stepper that touch the same device-resident state ----
 * init_first : initial_setup + first-call block of Ida::solve (src/lib.rs:537-545, src/impl_solve.rs:120-126):
 *              ewt = ewt_set(phi[0]); hYpnorm[s] = ||phi[1]||_wrms(ewt); hPhi0Nrm[s] = ||phi[0]||_wrms(ewt) (the first
 *              tolsf test, impl_solve.rs:289-295).
 * scale_phi1 : phi[1] *= hFac[s]   (phi[1] = hh*y', impl_solve.rs:167-168; reset() after a failed first step)
 * predict    : set_coeffs' phi-star scaling phi[j] *= beta[j], j = ns..kk (lib.rs:768-779) followed by IDAPredict
 *              (lib.rs:894-959). hKkNs [nsys][2], hBeta/hGamma [nsys][6].
 * post_newton: yy = yypredict + ee; yp = yppredict + cj*ee (lib.rs:845-849) and the four norms the error test /
 *              order selection may need (lib.rs:983-1004, impl_complete_step.rs:74-77):
 *              hNorms[s] = { ||ee||, ||ee+phi[kk]||, ||ee+phi[kk]+phi[kk-1]||, ||ee-phi[kk+1]|| } (0 where undefined).
 * restore    : IDARestore's phi part, phi[j] *= cvals[j-ns], j = ns..kk (lib.rs:1057-1082). hCvals [nsys][6].
 * complete_step: phi[kused+1] = ee (if kused < maxord), the phi recurrence (impl_complete_step.rs:152-176), ee *= ck
 *              (lib.rs:708), then the next step's ewt_set(phi[0]) and tolsf norm (impl_solve.rs:266-295):
 *              hPhi0Nrm[s] = ||phi[0]||_wrms(ewt), hEwtBad[s] != 0 if some ewt component <= 0.
 * get_solution: IDAGetSolution's linear combinations (lib.rs:1319-1340). hCvals [nsys][6], hDvals [nsys][5].       */
int idahip_init_first(idahip_ctx* ctx, double* hYpnorm, double* hPhi0Nrm, const int32_t* hIdx, int nsys);
int idahip_scale_phi1(idahip_ctx* ctx, const double* hFac, const int32_t* hIdx, int nsys);
int idahip_predict(idahip_ctx* ctx, const int32_t* hKkNs, const double* hBeta, const double* hGamma, const int32_t* hIdx, int nsys);
int idahip_post_newton(idahip_ctx* ctx, const double* hCj, const int32_t* hKk, double* hNorms, const int32_t* hIdx, int nsys);
int idahip_restore(idahip_ctx* ctx, const int32_t* hKkNs, const double* hCvals, const int32_t* hIdx, int nsys);
int idahip_complete_step(idahip_ctx* ctx, const int32_t* hKused, const double* hCk, int maxord, double* hPhi0Nrm,
                         int32_t* hEwtBad, const int32_t* hIdx, int nsys);
int idahip_get_solution(idahip_ctx* ctx, const int32_t* hKord, const double* hCvals, const double* hDvals,
                        const int32_t* hIdx, int nsys);
/* IDAGetDky (src/lib.rs:424-529), the vector part: hOut[s][0..n) = sum_{j = hKfirst[s] .. hKlast[s]} hCjk[s][j] * phi[j] of
 * listed system s, accumulated from zero in ascending j (lib.rs:517-526). The coefficients c_j^(k)(t) (lib.rs:464-508) are
 * the host's (libidaens: idaens_get_dky); 0 <= kfirst <= klast <= 5. hOut is a host array [nsys][n]. */
int idahip_get_dky(idahip_ctx* ctx, const int32_t* hKfirst, const int32_t* hKlast, const double* hCjk /*[nsys][6]*/, double* hOut,
                   const int32_t* hIdx, int nsys);

/* Ida::new again for some systems of a running ensemble (src/lib.rs:278-405: phi[0] = yy = y0, phi[1] = yp = y0'):
 * idahip_snapshot_initial keeps a device copy of the current phi[0], phi[1] of every system (call it right after the
 * initial conditions were uploaded); idahip_restore_initial puts the listed systems back to it. */
int idahip_snapshot_initial(idahip_ctx* ctx);
int idahip_restore_initial(idahip_ctx* ctx, const int32_t* hIdx, int nsys);

/* ---- device-resident stepper for small systems (n <= 8, IDAHIP_ROBERTS / IDAHIP_LORENZ63) ----
 * The whole of Ida::solve (src/impl_solve.rs:69-376: first-call block, loop-top checks, Ida::step with its attempt loop,
 * Newton::solve, error test, complete_step, stop tests and the interpolation to tout), for every system of the ctx, in ONE
 * launch: one thread per IVP runs its own time loop, the step-size and order controller included. The
 * controller is the same source as libidaens' host stepper (rust-ida_amd/host/ida_controller.hpp) compiled for the device
 * with a pow that reproduces glibc's bits (rust-ida_amd/csrc/glibc_pow.hpp).
 *   hSys       [batch] controller states (idactl::SysCore of ida_controller.hpp, sys_bytes = sizeof of it), in and out
 *   call       the schedule and the limits of this call (what idaens_solve / _solve_schedule / _stream take)
 *   hRoundsDone[batch] out: step-attempt rounds each system took part in
 *   hAcc       [2] out: Newton iterations of the integrations retired by `recycle`, number of those integrations
 *   hYout/hYPout optional raw dumps [ntout][batch][n] of the device-side output slots (a slot is written when its tout is
 *              reached; the caller knows from the states which slots are new)
 * idahip_pow_batch: the controller's pow for n argument pairs, computed on the device (test hook for glibc_pow.hpp). */
/* Root finding on the device (src/impl_r_check.rs:32-576) for the function family of the reference's own example,
 * g_i(t, y, y') = y[comp[i]] - threshold[i] (examples/roberts.rs:53-56): per-system state of `Ida`'s root fields
 * (ida_glo, ida_ghi, ida_grout, ida_iroots, ida_gactive: src/lib.rs:225-244), moved with the controller record. */
#define IDAHIP_MAX_ROOTS 4
typedef struct idahip_root_state {
    double glo[IDAHIP_MAX_ROOTS], ghi[IDAHIP_MAX_ROOTS], grout[IDAHIP_MAX_ROOTS], iroots[IDAHIP_MAX_ROOTS];
    int32_t gactive[IDAHIP_MAX_ROOTS];
} idahip_root_state;
typedef struct idahip_tiny_call {
    const double* touts; /* [ntout] host */
    int ntout;
    int recycle;         /* idaens_stream: a system that finished its schedule is created anew and starts over */
    int resume;          /* continuing a round-limited schedule call: idle systems have finished */
    long max_rounds;     /* step attempts per system in this launch; 0 = until every system has returned */
    long mxstep;
    int maxord;
    long maxnef, maxncf;
    double epcon, hmax_inv, t0;
    const int64_t* start_round; /* [batch] host or NULL: idaens_stream's staggered start (absolute round numbers) */
    int64_t round_base;         /* rounds executed before this call */
    int nroots;                 /* 0, or the number of root functions (<= IDAHIP_MAX_ROOTS) */
    const int32_t* root_comps;  /* [nroots] host */
    const double* root_thresholds; /* [nroots] host */
    idahip_root_state* root_states; /* [batch] host, in and out (NULL iff nroots == 0) */
} idahip_tiny_call;
int idahip_tiny_solve(idahip_ctx* ctx, void* hSys, size_t sys_bytes, const idahip_tiny_call* call, int64_t* hRoundsDone, uint64_t* hAcc,
                      double* hYout, double* hYPout);
int idahip_pow_batch(idahip_ctx* ctx, const double* hX, const double* hY, double* hOut, size_t count);
/* The same call for larger systems (8 < n <= 4096, IDAHIP_LINEAR_DENSE or IDAHIP_HEAT1D, LU variant 4) as LOCK-STEP ROUNDS driven from the
 * device side: a round = one step attempt of every stepping system (Ida::step's attempt loop body, src/lib.rs:613-711, with
 * Newton::solve, crates/nonlinear/src/newton.rs:51-167, and the batched kernels of this library inside), the controller
 * state of every system and the index lists live on the device, the host only enqueues each round's fixed launch sequence:
 * one synchronisation per round (to learn whether any system still steps), none at all when `recycle` is set (throughput
 * mode runs exactly max_rounds rounds). Arguments as idahip_tiny_solve; rounds_run: rounds executed by this call. */
int idahip_round_solve(idahip_ctx* ctx, void* hSys, size_t sys_bytes, const idahip_tiny_call* call, int64_t* hRoundsDone, uint64_t* hAcc,
                       double* hYout, double* hYPout, int64_t* rounds_run);

/* LU implementation choice. Both variants are bit-identical to dense_get_rf; they factor in 64-column
 * super-panels with a rank-64 trailing update in wave-private 16-row strips and differ in how a super-panel is factored:
 *   4 = default: one wavefront per matrix factors the whole super-panel (<= 512 live rows);
 *   3 = two 32-column panels with two rows per lane and a narrow update between them (the cross-check in the tests, and what
 *       the leading super-panels of matrices with more than 512 rows use; more than 1024 rows: 8-column panels).
 * Any other value is refused. (Round 2's variant 5, the same kernels with FMA-contracted updates, is gone: DESIGN.md.) */
int idahip_set_lu_variant(idahip_ctx* ctx, int variant);
int idahip_lu_variant(const idahip_ctx* ctx); /* the variant in force */
/* Matrices with more than 1024 rows: how a 64-column super-panel is factored. 1 = in one launch of the left-looking
 * workgroup-per-matrix kernel (lu_superpanel_kernel): the faster pipeline for BANDED matrices in dense storage (config 4's heat
 * Jacobians: 238 against 322 us per 4096 x 4096 matrix) and the default for IDAHIP_HEAT1D; 0 = eight 8-column panel launches with a
 * narrow update after each, faster on DENSE matrices (they spread the update over several workgroups per matrix) and the default for
 * every other problem kind. A hint about structure, never about results: the factors are bit-identical either way. */
int idahip_set_lu_superpanel(idahip_ctx* ctx, int on);
int idahip_lu_superpanel(const idahip_ctx* ctx); /* the setting in force */
/* Device lock-step stepper (idahip_round_solve): linear setups -- Jacobian + dense_get_rf, ida_ls.rs:232-290 -- batched over rounds.
 * With rounds = k > 1 a lock-step round sets nothing up unless at least (k - 1) / k of the systems that are stepping ask for a setup
 * (lib.rs:806-817) or k - 1 rounds in a row have waited already; a system that asks waits, its attempt begun, while the others go
 * on stepping, and a round without setups launches no factorisation at all. 1 = every round serves its setups (the default); at
 * most 64. A scheduling choice: every system performs the same attempts with the same arithmetic, so its results and counters do not
 * change (tests/test_gpu_device_controller.py). It pays where a batched factorisation costs about the same for 50 matrices as for
 * 250 -- matrices of thousands of rows, a workgroup per matrix and a long chain of launches (config 4) -- and costs throughput where
 * the factorisation's time is proportional to the batch (config 3). */
int idahip_set_lu_period(idahip_ctx* ctx, int rounds);
int idahip_lu_period(const idahip_ctx* ctx); /* the setting in force */
/* 0 for the product library. 1 for a TIMING BUILD (-DIDAHIP_TIMING_BUILD, rust-ida_amd/csrc/exp_switches.hpp): a library in
 * which parts of kernels were removed or replaced to measure what they cost -- its results are garbage by design; a caller
 * that cares (tests, bench.py) refuses to run on one. No ctx, no device needed. */
int idahip_timing_build(void);

/* ---- measurement hooks (bench.py / profiles): device time of the launches of the last call, by HIP events on the
 * ctx stream, and launch counters per kernel class ---- */
typedef enum {
    IDAHIP_K_NEWTON_ITER = 0, IDAHIP_K_SYS = 1, IDAHIP_K_JAC = 2, IDAHIP_K_LU = 3, IDAHIP_K_VECTOR = 4, IDAHIP_K_SOLVE = 5,
    IDAHIP_K_SYS_JAC = 6, /* fused residual + Jacobian pass of idahip_nls_sys_setup */
    /* single kernels of the batched LU (timing level 2 only): every launch of the kernel is bracketed by its own pair of
     * events; `systems` counts matrices x launches */
    IDAHIP_K_LU_PANEL = 7,    /* lu_wavepanel_kernel (variant 3, or more than 512 live rows: lu_panel2_kernel / lu_panelr_kernel + narrow lu_trail_kernel) */
    IDAHIP_K_LU_TRAIL = 8,    /* lu_trail64w_kernel */
    IDAHIP_K_LU_FINALIZE = 9, /* lu_finalize_kernel */
    IDAHIP_K_COUNT = 10
} idahip_kclass;
/* on = 0: no events, no synchronisation (launch counters still run); 1: device time per kernel class (one event pair and
 * one hipEventSynchronize per class call); 2: additionally per kernel of the LU (which perturbs the class figure of the
 * LU: use separate passes). */
int idahip_timing_enable(idahip_ctx* ctx, int on);
/* accumulated device milliseconds and launch count of a kernel class since the last reset */
int idahip_timing_get(idahip_ctx* ctx, idahip_kclass k, double* ms, int64_t* launches, int64_t* systems);
int idahip_timing_reset(idahip_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* IDA_HIP_H */
