// ORACLE (test infrastructure, never shipped / never on the product path).
//
// CPU restatement of the reference's Newton nonlinear solver.
//   NLProblem / NLSolver traits <- /root/reference/crates/nonlinear/src/traits.rs:5-209
//   Newton::solve               <- /root/reference/crates/nonlinear/src/newton.rs:51-167
//   error taxonomy              <- /root/reference/crates/nonlinear/src/lib.rs:10-35
//
// Deviation: in the reference, ConvergenceRecover with jcur == true falls out of
// the match arm without `break`, so the 'outer loop re-runs forever (newton.rs:146-153). C SUNDIALS breaks
// out and returns the error; so does this restatement. Not exercised by any reference golden.
//
// Pinned by the reference's own known-answer test newton.rs:306-343 (tests/golden/newton_golden.json).
#pragma once
#include <vector>

namespace oracle {

// Return-code convention documented at traits.rs:17-22: 0 ok, >0 recoverable, <0 unrecoverable.
enum NlsCode {
    NLS_SUCCESS = 0,
    NLS_CONV_RECVR = 1,     // nonlinear::Error::ConvergenceRecover
    NLS_LSETUP_RECVR = 2,   // linear::Error::LUFactFail surfaced as a recoverable lsetup failure (Q2)
    NLS_ILL_INPUT = -1,
};

struct Newton;

struct NLProblem {
    virtual ~NLProblem() {}
    /// f = F(ycor)                                   (traits.rs `sys`)
    virtual int sys(const double* ycor, double* f) = 0;
    /// set up the linear solver; *jcur = true if the Jacobian was refreshed (traits.rs `setup`)
    virtual int setup(const double* ycor, const double* f, bool jbad, bool* jcur) = 0;
    /// solve J x = b in place                        (traits.rs `solve`)
    virtual int solve(const double* ycor, double* b) = 0;
    /// convergence test: *converged set; returns NLS_SUCCESS or an error (traits.rs `ctest`)
    virtual int ctest(const Newton& solver, const double* y, const double* del, double tol, const double* ewt,
                      bool* converged) = 0;
};

struct Newton {
    int n;
    std::vector<double> delta;  // Newton update vector (newton.rs:21)
    bool jcur = false;
    int curiter = 0;
    int maxiters;
    long niters = 0;
    long nconvfails = 0;

    Newton(int n_, int maxiters_) : n(n_), delta(n_, 0.0), maxiters(maxiters_) {}

    int get_cur_iter() const { return curiter; }

    // newton.rs:51-167
    int solve(NLProblem& problem, const double* y0, double* y, const double* w, double tol, bool call_lsetup) {
        bool jbad = false;
        int retval;
        for (;;) {  // 'outer
            retval = problem.sys(y0, delta.data());
            if (retval == NLS_SUCCESS && call_lsetup) {
                bool jc = jcur;
                retval = problem.setup(y0, delta.data(), jbad, &jc);
                jcur = jc;  // C IDA marks the Jacobian current even when lsetup reports a recoverable failure
            }
            if (retval == NLS_SUCCESS) {
                curiter = 0;
                for (int i = 0; i < n; ++i) y[i] = y0[i];
                for (;;) {  // 'inner
                    niters += 1;
                    for (int i = 0; i < n; ++i) delta[i] = -delta[i];
                    retval = problem.solve(y, delta.data());
                    if (retval != NLS_SUCCESS) break;
                    for (int i = 0; i < n; ++i) y[i] += delta[i];
                    bool converged = false;
                    retval = problem.ctest(*this, y, delta.data(), tol, w, &converged);
                    if (retval != NLS_SUCCESS) break;
                    if (converged) {
                        jcur = false;
                        return NLS_SUCCESS;
                    }
                    curiter += 1;
                    if (curiter >= maxiters) {
                        retval = NLS_CONV_RECVR;
                        break;
                    }
                    retval = problem.sys(y, delta.data());
                    if (retval != NLS_SUCCESS) break;
                }
            }
            // recoverable failure with stale Jacobian data: retry with a fresh lsetup (newton.rs:146-152)
            if (retval == NLS_CONV_RECVR && !jcur) {
                nconvfails += 1;
                call_lsetup = true;
                jbad = true;
                continue;
            }
            break;  // Q3: C semantics
        }
        nconvfails += 1;
        return retval;
    }
};

}  // namespace oracle
