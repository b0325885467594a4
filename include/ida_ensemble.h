/* ida_ensemble.h -- C entry points of libidaens.so: the host-side BDF stepper for an ensemble of independent IVPs.
 *
 * This is the caller side of the drop-in boundary: it plays the role of the reference's `Ida` object
 * (src/lib.rs:89-244) -- `Ida::new`, `Ida::solve`, the getters of src/ida_io.rs -- for `batch` systems at once, and
 * reaches the device only through include/ida_hip.h. In the reference this layer is Rust and stays Rust; no Rust
 * toolchain exists in the build image, so the stand-in is C++ with the reference's structure (IdaNLProblem /
 * IdaLProblem / Newton state per system, same names, same error behaviour). All scalar control logic (set_coeffs,
 * lsetup decision, idaNlsConvTest incl. powf, test_error decisions, handle_n_flag, complete_step order/step selection,
 * stop tests, get_solution coefficients) is one source (rust-ida_amd/host/ida_controller.hpp) that runs on the host with the
 * platform libm, per system, exactly as src/lib.rs / src/impl_*.rs do -- or, for small systems, on the device with a pow
 * that reproduces that libm's bits (idaens_set_device_controller); vectors never leave the device.
 */
#ifndef IDA_ENSEMBLE_H
#define IDA_ENSEMBLE_H

#include <stdint.h>

#include "ida_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct idaens idaens;

/* IdaTask (src/lib.rs:50-54) */
enum { IDAENS_NORMAL = 0, IDAENS_ONE_STEP = 1 };
/* per-system return of solve: IdaSolveStatus (src/lib.rs:56-62) >= 0, IdaError (src/error.rs) < 0 */
enum {
    IDAENS_SUCCESS = 0,
    IDAENS_TSTOP_RETURN = 1,
    IDAENS_ROOT_RETURN = 2,
    IDAENS_UNFINISHED = 99, /* round limit hit before the system reached tout (bench mode only) */
    IDAENS_TOO_MUCH_WORK = -1,
    IDAENS_TOO_MUCH_ACC = -2,
    IDAENS_ERR_FAIL = -3,
    IDAENS_CONV_FAIL = -4,
    IDAENS_LSETUP_FAIL = -6,
    IDAENS_CLOSE_ROOTS = -10, /* IdaError::CloseRoots (impl_r_check.rs:199) */
    IDAENS_RTFUNC_FAIL = -12, /* a host root function returned non-zero (C IDA's IDA_RTFUNC_FAIL; the reference has the check
                                 commented out, impl_r_check.rs:83) */
    IDAENS_ILL_INPUT = -22,
    IDAENS_BAD_K = -25,
    IDAENS_BAD_T = -26
};

/* Ida::new(problem, yy0, yp0, tol_control) for every system of ctx (src/lib.rs:278-405); ctx must already hold the
 * tolerances and problem data. hYY0, hYP0: [batch][n]. */
int idaens_create(idaens** e, idahip_ctx* ctx, const double* hYY0, const double* hYP0);
int idaens_destroy(idaens* e);
const char* idaens_last_error(const idaens* e);

/* optional inputs (the reference has defaults only, src/lib.rs:309-321; setters follow C IDA's names) */
int idaens_set_max_num_steps(idaens* e, long mxstep); /* 0 = unlimited; default 500 (MXSTEP_DEFAULT) */
int idaens_set_max_ord(idaens* e, int maxord);          /* 1..5, default 5 */
/* on (default; always off for IDAHIP_HOST_CALLBACK problems): a Newton solve runs its first two iterations and their
 * convergence tests in one device call (idahip_newton_iter2) instead of one host round trip per iteration. Results are
 * identical either way; the switch exists for measurements. */
int idaens_set_fused_newton(idaens* e, int on);
/* on (default): the step-size and order controller runs on the device where a device stepper exists
 * (IDA_NORMAL, no root functions, no trace):
 *   - small systems (n <= 8: Roberts, Lorenz63): a solve / solve_schedule / stream call is ONE launch in which every system
 *     runs its own time loop (idahip_tiny_solve);
 *   - linear dense and heat problems with 8 < n <= 4096: lock-step rounds as below, but enqueued without a host round trip
 *     inside a round -- one synchronisation per round, none in idaens_stream (idahip_round_solve); for n > 1024 one more,
 *     hidden behind the residual kernels: the length of the round's LU list comes back to size the factorisation's launches.
 *     A Newton solve that has to start over with a fresh Jacobian does so in the next round, so a system may need one round
 *     more than with the host stepper; its steps, orders, counters and results are the same.
 * off: the lock-step host stepper for every problem. Same results either way (one controller source, pow with glibc's
 * bits); the switch is the A/B. idaens_create compares the device pow with this host's std::pow on the controller's argument
 * ranges once per process: if they differ (another libm than the one glibc_pow.hpp restates) the device steppers stay off,
 * idaens_last_error says so and this call returns 1 for on != 0. */
int idaens_set_device_controller(idaens* e, int on);
/* Which stepper an idaens_solve / _solve_schedule call (IDAENS_NORMAL) of this ensemble would run on as things stand:
 * 0 = the host stepper, 1 = the device stepper with one thread per system (n <= 8), 2 = the device lock-step rounds. A test
 * or a benchmark that means to measure a device stepper asserts this instead of trusting the setter. */
int idaens_device_controller_active(const idaens* e);
/* Root finding (the Root trait, src/traits.rs:72-94; src/impl_r_check.rs): nroots functions g_i(t, y, y') = y[comps[i]] -
 * thresholds[i] for every system -- the form of the reference's Roberts example (g0 = y0 - 1e-4, g1 = y2 - 0.01). Call
 * before the first solve; idaens_solve then reports IDAENS_ROOT_RETURN with tret = the root, yy/yp = the solution there,
 * and idaens_get_roots gives rootsfound (ida_iroots: -1/0/+1 per function). Root closures cannot cross the C ABI, so the
 * functions are this parametrised family. */
int idaens_set_roots(idaens* e, int nroots, const int32_t* comps, const double* thresholds);
int idaens_get_roots(const idaens* e, int32_t* out /* [batch][nroots] */);
/* Root::root (src/traits.rs:72-90) for any user function: gout[0..nroots) = g(t, yy, yp) of system `sys`, evaluated on the
 * host with y(t), y'(t) interpolated on the device and copied back (yy, yp: n doubles, valid during the call). Return 0, or
 * non-zero to fail that system with IDAENS_RTFUNC_FAIL. Replaces idaens_set_roots' family; same calling rules. The bracketing
 * (impl_r_check.rs) is per-system scalar work on the host either way: roots are rare events of single systems. */
typedef int (*idaens_root_fn)(void* user, int32_t sys, double t, const double* yy, const double* yp, int32_t nroots, double* gout);
int idaens_set_root_fn(idaens* e, int nroots, idaens_root_fn fn, void* user);

/* Ida::solve(tout, &mut tret, itask) for every system (src/impl_solve.rs:69-376). hTret/hStatus: [batch].
 * max_rounds > 0 bounds the number of lock-step attempt rounds (systems still stepping report IDAENS_UNFINISHED and
 * resume on the next call with the same tout). Returns 0, or < 0 on a device/ABI failure.
 * A negative per-system status (an IdaError) is sticky: later solve calls leave that system alone and report the same status
 * again (the reference's Ida::solve can be called again after an error and would try to continue). */
int idaens_solve(idaens* e, double tout, int itask, double* hTret, int32_t* hStatus, long max_rounds);

/* The same for a whole output schedule: for every system Ida::solve(touts[0]), Ida::solve(touts[1]), ... in IDA_NORMAL
 * mode -- each return processed exactly as the reference does (stop tests, interpolation to tout) -- but a system that
 * returns from one call enters the next at once instead of waiting for the slowest system of the batch, so the lock-step
 * rounds stay full. A system's schedule ends early with its first status other than IDAENS_SUCCESS (root return, error).
 * hReached[batch] (optional): number of touts returned with IDAENS_SUCCESS; hYout / hYPout (optional, [ntout][batch][n]):
 * y, y' at every tout reached. With max_rounds > 0 the call may stop early (IDAENS_UNFINISHED); calling again with the same
 * schedule continues it. */
int idaens_solve_schedule(idaens* e, const double* touts, int ntout, double* hTret, int32_t* hStatus, int32_t* hReached,
                          double* hYout, double* hYPout, long max_rounds);

/* Throughput mode: the schedule as above, and a system that has returned from its last tout is created anew from its
 * initial conditions (Ida::new) and starts the schedule again at once -- the batch never drains, every lock-step round
 * works on every system, each somewhere else in its integration. Runs max_rounds rounds; call again to continue.
 * stagger_rounds > 0 (first call only): system b enters at round b * stagger_rounds / batch, which spreads the systems
 * evenly over the phases of an integration from the start. passes_done (optional): integrations completed since the
 * ensemble was created. Per-system counters restart with the system; idaens_total_newton_iters keeps the total. */
int idaens_stream(idaens* e, const double* touts, int ntout, long max_rounds, long stagger_rounds, int64_t* passes_done);

/* Several ensembles side by side on ONE device. The reference integrates one IVP per `Ida` object and
 * nothing couples two objects (src/lib.rs:89-244); how many of them a host puts into one lock-step batch is a scheduling
 * choice. A lock-step round of one large batch is a serial chain of launches, and in some of them most of the chip idles
 * (the panel kernels' pivot chains, the later Newton passes that serve a few hundred systems, round begin / end). Split
 * into `ngroups` ensembles -- each created on its OWN idahip_ctx, hence its own HIP stream -- and driven by one host
 * thread each, one group's idle stretches are filled by the other groups' launches by the device's own scheduler: same
 * systems, same per-system results (every system is integrated exactly as alone), more of them per second.
 * idaens_stream_group = idaens_stream for every ens[g], concurrently; group g starts g * offset_us microseconds after
 * group 0 (0 = together). passes_done: [ngroups] or null. idaens_solve_schedule_group = idaens_solve_schedule for every
 * ens[g], concurrently (arrays of per-group result pointers: hTret[g], hStatus[g] of length batch(ens[g]); hReached may be
 * null). Returns 0, or the first failing group's (negative) code -- that group's idaens_last_error has the text.
 * Host threads: the calling thread drives group 0, ngroups - 1 std::threads the others; a ctx is used by one thread only.
 * Create the contexts on streams from idahip_concurrent_streams (ida_hip.h): the HIP runtime is free to put two ordinary
 * streams on one hardware queue, and those two groups then take turns instead of running side by side. */
int idaens_stream_group(idaens* const* ens, int ngroups, const double* touts, int ntout, long max_rounds, long stagger_rounds, long offset_us,
                        int64_t* passes_done);
int idaens_solve_schedule_group(idaens* const* ens, int ngroups, const double* touts, int ntout, double* const* hTret, int32_t* const* hStatus,
                                int32_t* const* hReached, long max_rounds);

/* getters (src/ida_io.rs:11-117), arrays of length batch */
enum {
    IDAENS_C_NST = 0, IDAENS_C_NRE = 1, IDAENS_C_NJE = 2, IDAENS_C_NSETUPS = 3, IDAENS_C_NNI = 4, IDAENS_C_NETF = 5,
    IDAENS_C_NCFN = 6, IDAENS_C_NATTEMPTS = 7, IDAENS_C_NLS_NCONVFAILS = 8, IDAENS_C_KUSED = 9, IDAENS_C_KK = 10,
    IDAENS_C_NGE = 11 /* root-function evaluations (ida_nge) */,
    /* times the system took a path on which this library follows C IDA and not the reference's text:      */
    IDAENS_C_NLUFAIL = 12 /* Q2: zero pivot reported by the factorisation, treated as recoverable                          */,
    IDAENS_C_NCONV_JCUR = 13 /* Q3/Q4: Newton's ConvergenceRecover with a current Jacobian, treated as recoverable          */,
    IDAENS_C_NFAIL_FIRST = 14 /* Q5: failed attempts before the first step (reset() rescales phi[1] only)                   */,
    IDAENS_C_NLI = 15 /* idaLsSolve: linear iterations (0 with a direct LSolver, src/ida_ls.rs:389-400) */,
    IDAENS_C_NCFL = 16 /* idaLsSolve: linear convergence failures (src/ida_ls.rs:413-415) */
};
int idaens_get_counter(const idaens* e, int which, int64_t* out);
enum { IDAENS_R_TN = 0, IDAENS_R_HUSED = 1, IDAENS_R_HH = 2, IDAENS_R_H0U = 3, IDAENS_R_TOLSF = 4 };
int idaens_get_real(const idaens* e, int which, double* out);
/* get_yy / get_yp: [batch][n] */
int idaens_get_yy(idaens* e, double* hYY);
int idaens_get_yp(idaens* e, double* hYP);
/* Ida::get_dky(t, k, dky) for every system (src/lib.rs:424-529, IDAGetDky): the k-th derivative of the interpolating
 * polynomial of the last step at t. hDky: [batch][n]; hStatus[batch]: IDAENS_SUCCESS, IDAENS_BAD_K (k > kused of that system)
 * or IDAENS_BAD_T (t outside the last step), rows of hDky with a bad status are left untouched. The coefficient recurrence
 * runs here (C IDA's inner-loop bound kused - k + i; the reference's kused - k + 1 agrees for k <= 1, drops terms for
 * k >= 2 and indexes out of bounds for k = 0, kused = 5 -- SURVEY.md quirk Q9), the sums on the device (idahip_get_dky). */
int idaens_get_dky(idaens* e, double t, int k, double* hDky, int32_t* hStatus);
/* totals over the ensemble since creation */
int64_t idaens_total_newton_iters(const idaens* e);
int64_t idaens_total_rounds(const idaens* e);
/* per-accepted-step trace of one system (tn, hused, kused), for parity tests; enable before solving */
int idaens_trace_system(idaens* e, int sys);
long idaens_trace_len(const idaens* e);
int idaens_trace_get(const idaens* e, double* out /* [len][3] */);

#ifdef __cplusplus
}
#endif
#endif
