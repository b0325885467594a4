// ORACLE (test infrastructure, never shipped / never on the product path).
//
// Flat C entry points over the CPU restatement, for ctypes-driven tests, golden replay and the
// `cpu_baseline` leg of bench.py. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may
// load this library.
#include <chrono>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "dense.hpp"
#include "ida.hpp"
#include "newton.hpp"
#include "problems.hpp"

using namespace oracle;

namespace {

// src/tests/mod.rs:18-63 -- `Dummy` problem with empty res/jac used by the state-injection tests
struct Dummy : Problem {
    int model_size() const override { return 3; }
    void res(double, const double*, const double*, double*) const override {}
    void jac(double, double, const double*, const double*, const double*, double*) const override {}
};

// crates/nonlinear/src/newton.rs:182-304 -- the 3-equation test system of the Newton known-answer test
struct NewtonTestProblem : NLProblem {
    double a[9];  // column-major
    double x[3];
    Dense lsolver{3};
    int sys(const double* y, double* f) override {
        const double X = y[0], Y = y[1], Z = y[2];
        f[0] = X * X + Y * Y + Z * Z - 1.0;
        f[1] = 2.0 * (X * X) + Y * Y - 4.0 * Z;
        f[2] = 3.0 * (X * X) - 4.0 * Y + Z * Z;
        return NLS_SUCCESS;
    }
    int setup(const double* y, const double*, bool, bool* jcur) override {
        const double X = y[0], Y = y[1], Z = y[2];
        a[0] = 2.0 * X; a[1] = 4.0 * X; a[2] = 6.0 * X;   // column 0
        a[3] = 2.0 * Y; a[4] = 2.0 * Y; a[5] = -4.0;      // column 1
        a[6] = 2.0 * Z; a[7] = -4.0;    a[8] = 2.0 * Z;   // column 2
        if (lsolver.setup(a) != 0) return NLS_LSETUP_RECVR;
        *jcur = true;
        return NLS_SUCCESS;
    }
    int solve(const double*, double* b) override {
        lsolver.solve(a, x, b);
        for (int i = 0; i < 3; ++i) b[i] = x[i];
        return NLS_SUCCESS;
    }
    int ctest(const Newton&, const double*, const double* del, double tol, const double* ewt, bool* converged) override {
        *converged = norm_wrms(del, ewt, 3) <= tol;
        return NLS_SUCCESS;
    }
};

struct Handle {
    std::unique_ptr<Problem> problem;
    std::unique_ptr<Ida> ida;
};

std::unique_ptr<Problem> make_problem(int kind, int n, const double* params, const double* A, const double* B, const double* c) {
    switch (kind) {
        case 0: return std::unique_ptr<Problem>(new Roberts());
        case 1: {
            auto* p = new Lorenz63();
            if (params) { p->p = params[0]; p->r = params[1]; p->b = params[2]; }
            return std::unique_ptr<Problem>(p);
        }
        case 2: {
            auto* p = new LinearDense();
            p->n = n; p->A = A; p->B = B; p->c = c;
            return std::unique_ptr<Problem>(p);
        }
        case 3: {
            auto* p = new Heat1D();
            p->n = n; p->coef = params[0];
            return std::unique_ptr<Problem>(p);
        }
        default: return std::unique_ptr<Problem>(new Dummy());
    }
}

TolControl make_tol(double rtol, const double* atol, int natol) {
    TolControl tc;
    tc.rtol = rtol;
    if (natol <= 1) tc.atol_s = atol[0];
    else tc.atol_v.assign(atol, atol + natol);
    return tc;
}

}  // namespace

extern "C" {

int oracle_dense_getrf(double* a, int m, int n, int64_t* pivot) { return dense_get_rf(a, m, n, pivot); }
void oracle_dense_getrs(const double* a, int n, const int64_t* pivot, double* b) { dense_get_rs(a, n, pivot, b); }
double oracle_norm_wrms(const double* x, const double* w, int n) { return norm_wrms(x, w, n); }
double oracle_norm_wrms_masked(const double* x, const double* w, const uint8_t* id, int n) { return norm_wrms_masked(x, w, id, n); }

// LSolver trait use as in dense.rs:313-328 (`test_dense1`): setup then solve through the solver object.
int oracle_dense_lsolver(double* a, int n, const double* b, double* x, int64_t* pivots_out) {
    Dense d(n);
    const int info = d.setup(a);
    if (info) return info;
    d.solve(a, x, b);
    for (int i = 0; i < n; ++i) pivots_out[i] = d.pivots[i];
    return 0;
}

// newton.rs:306-343
int oracle_newton_test(const double* y0, const double* w, double tol, int maxiters, double* y, long* niters, long* nconvfails) {
    NewtonTestProblem p;
    Newton newton(3, maxiters);
    const int r = newton.solve(p, y0, y, w, tol, true);
    *niters = newton.niters;
    *nconvfails = newton.nconvfails;
    return r;
}

// -------------------------------------------------------------------------------------------- Ida object
void* oracle_ida_create(int kind, int n, const double* params, const double* A, const double* B, const double* c,
                        const double* yy0, const double* yp0, double rtol, const double* atol, int natol) {
    Handle* h = new Handle();
    h->problem = make_problem(kind, n, params, A, B, c);
    h->ida.reset(new Ida(h->problem.get(), yy0, yp0, make_tol(rtol, atol, natol)));
    return h;
}
void oracle_ida_destroy(void* vh) { delete (Handle*)vh; }

int oracle_ida_solve(void* vh, double tout, double* tret, int itask) {
    return ((Handle*)vh)->ida->solve(tout, tret, (IdaTask)itask);
}

// Scalar field access by name (ints are passed as doubles). Returns 0 if the name is known.
#define SCALAR_FIELDS(X)                                                                                         \
    X("kk", ida.ida_kk) X("kused", ida.ida_kused) X("knew", ida.ida_knew) X("phase", ida.ida_phase) X("ns", ida.ida_ns)   \
    X("hh", ida.ida_hh) X("hused", ida.ida_hused) X("rr", ida.ida_rr) X("h0u", ida.ida_h0u) X("hin", ida.ida_hin)         \
    X("cj", ida.nlp.lp.ida_cj) X("cjold", ida.nlp.lp.ida_cjold) X("cjratio", ida.nlp.lp.ida_cjratio)                       \
    X("cjlast", ida.ida_cjlast) X("ss", ida.nlp.ida_ss) X("oldnrm", ida.nlp.ida_oldnrm) X("toldel", ida.nlp.ida_toldel)   \
    X("eps_newt", ida.ida_eps_newt) X("tn", ida.nlp.ida_tn) X("tretlast", ida.ida_tretlast) X("tolsf", ida.ida_tolsf)     \
    X("nst", ida.ida_nst) X("ncfn", ida.ida_ncfn) X("netf", ida.ida_netf) X("nre", ida.nlp.ida_nre)                       \
    X("nsetups", ida.nlp.ida_nsetups) X("nje", ida.nlp.lp.nje) X("nni", ida.nls.niters)                                   \
    X("nls_nconvfails", ida.nls.nconvfails) X("nge", ida.ida_nge) X("maxord", ida.ida_maxord)                             \
    X("hmax_inv", ida.ida_hmax_inv) X("mxstep", ida.ida_mxstep) X("n_attempts", ida.n_attempts)                           \
    X("suppressalg", ida.ida_suppressalg) X("tlo", ida.ida_tlo) X("trout", ida.ida_trout) X("jcur", ida.nls.jcur)

int oracle_ida_get_scalar(void* vh, const char* name, double* out) {
    Ida& ida = *((Handle*)vh)->ida;
#define X(nm, fld) if (!strcmp(name, nm)) { *out = (double)(fld); return 0; }
    SCALAR_FIELDS(X)
#undef X
    return -1;
}
int oracle_ida_set_scalar(void* vh, const char* name, double v) {
    Ida& ida = *((Handle*)vh)->ida;
#define X(nm, fld) if (!strcmp(name, nm)) { fld = (decltype(fld))v; return 0; }
    SCALAR_FIELDS(X)
#undef X
    return -1;
}

static double* vec_field(Ida& ida, const char* name, int* len) {
    const int n = ida.n;
#define V(nm, ptr, l) if (!strcmp(name, nm)) { *len = (l); return (ptr); }
    V("phi", ida.ida_phi.data(), MXORDP1 * n)
    V("psi", ida.ida_psi, MXORDP1) V("alpha", ida.ida_alpha, MXORDP1) V("beta", ida.ida_beta, MXORDP1)
    V("sigma", ida.ida_sigma, MXORDP1) V("gamma", ida.ida_gamma, MXORDP1) V("cvals", ida.ida_cvals, MXORDP1)
    V("dvals", ida.ida_dvals, MAXORD_DEFAULT)
    V("ee", ida.ida_ee.data(), n) V("delta", ida.ida_delta.data(), n) V("ewt", ida.nlp.ida_ewt.data(), n)
    V("yy", ida.nlp.ida_yy.data(), n) V("yp", ida.nlp.ida_yp.data(), n)
    V("yypredict", ida.nlp.ida_yypredict.data(), n) V("yppredict", ida.nlp.ida_yppredict.data(), n)
    V("savres", ida.nlp.ida_savres.data(), n) V("mat_j", ida.nlp.lp.mat_j.data(), n * n)
    V("iroots", ida.ida_iroots.data(), ida.ida_nrtfn)
#undef V
    return nullptr;
}
int oracle_ida_get_vec(void* vh, const char* name, double* out, int cap) {
    int len = 0;
    double* p = vec_field(*((Handle*)vh)->ida, name, &len);
    if (!p || cap < len) return -1;
    memcpy(out, p, sizeof(double) * len);
    return len;
}
int oracle_ida_set_vec(void* vh, const char* name, const double* in, int count) {
    int len = 0;
    double* p = vec_field(*((Handle*)vh)->ida, name, &len);
    if (!p || count > len) return -1;
    memcpy(p, in, sizeof(double) * count);
    return 0;
}

// Seams of the reference's state-injection tests (src/tests/*.rs): call one private method.
double oracle_ida_set_coeffs(void* vh) { return ((Handle*)vh)->ida->set_coeffs(); }
void oracle_ida_predict(void* vh) { ((Handle*)vh)->ida->predict(); }
void oracle_ida_restore(void* vh, double saved_t) { ((Handle*)vh)->ida->restore(saved_t); }
int oracle_ida_test_error(void* vh, double ck, double* err_k, double* err_km1) {
    return ((Handle*)vh)->ida->test_error(ck, err_k, err_km1) ? 1 : 0;
}
void oracle_ida_complete_step(void* vh, double err_k, double err_km1) { ((Handle*)vh)->ida->complete_step(err_k, err_km1); }
int oracle_ida_get_solution(void* vh, double t) { return ((Handle*)vh)->ida->get_solution(t); }
// IDAGetDky (lib.rs:424-529); literal_q9 != 0: the reference's own inner-loop bound, for the cross-check
int oracle_ida_get_dky(void* vh, double t, int k, double* dky, int literal_q9) {
    return ((Handle*)vh)->ida->get_dky(t, k, dky, literal_q9 != 0);
}
int oracle_ida_nonlinear_solve(void* vh) { return ((Handle*)vh)->ida->nonlinear_solve(); }
// lsetup seam: evaluate J at the object's current yy/yp/cj and factor it (ida_nls.rs:156-187)
int oracle_ida_lsetup(void* vh) {
    Ida& ida = *((Handle*)vh)->ida;
    bool jc = false;
    std::vector<double> res(ida.n, 0.0);
    return ida.nlp.setup(nullptr, res.data(), false, &jc);
}

void oracle_ida_record_steps(void* vh, int on) { ((Handle*)vh)->ida->record_steps = on != 0; }
long oracle_ida_num_recorded(void* vh) { return (long)((Handle*)vh)->ida->steps.size(); }
// out: [nrec][5] = tn, hused, kused, nni, nsetups
void oracle_ida_get_recorded(void* vh, double* out) {
    const auto& s = ((Handle*)vh)->ida->steps;
    for (size_t i = 0; i < s.size(); ++i) {
        out[5 * i + 0] = s[i].tn; out[5 * i + 1] = s[i].hused; out[5 * i + 2] = (double)s[i].kused;
        out[5 * i + 3] = (double)s[i].nni; out[5 * i + 4] = (double)s[i].nsetups;
    }
}

// -------------------------------------------------------------------------------------------- ensemble runner
// Integrates `nsys` independent IVPs (the reference's one-Ida-per-IVP model) to each of touts[0..ntout), using
// `nthreads` std::threads (systems dealt round-robin). Inputs are per-system contiguous blocks.
//   params: [nsys][nparam]; A,B: [nsys][n*n] col-major; c, yy0, yp0: [nsys][n]
// Outputs: yy_out, yp_out [ntout][nsys][n]; counters [nsys][8] = nst, nre, nje, nsetups, nni, netf, ncfn, n_attempts;
//          status [nsys] = last solve() return; last_k [nsys], last_h [nsys].
// Returns elapsed wall seconds of the integration loop (problem construction excluded).
double oracle_run_ensemble(int kind, int n, int nsys, int nparam, const double* params, const double* A, const double* B,
                           const double* c, const double* yy0, const double* yp0, double rtol, const double* atol,
                           int natol, const double* touts, int ntout, int nthreads, double* yy_out, double* yp_out,
                           double* counters, int* status, double* last_k_h) {
    std::vector<std::unique_ptr<Problem>> problems(nsys);
    std::vector<std::unique_ptr<Ida>> idas(nsys);
    const size_t nn = (size_t)n * n;
    for (int s = 0; s < nsys; ++s) {
        problems[s] = make_problem(kind, n, params ? params + (size_t)s * nparam : nullptr, A ? A + s * nn : nullptr,
                                   B ? B + s * nn : nullptr, c ? c + (size_t)s * n : nullptr);
        idas[s].reset(new Ida(problems[s].get(), yy0 + (size_t)s * n, yp0 + (size_t)s * n, make_tol(rtol, atol, natol)));
    }
    if (nthreads < 1) nthreads = 1;
    auto worker = [&](int tid) {
        for (int s = tid; s < nsys; s += nthreads) {
            Ida& ida = *idas[s];
            int st = 0;
            for (int k = 0; k < ntout; ++k) {
                double tret = 0.0;
                for (;;) {
                    st = ida.solve(touts[k], &tret, IDA_NORMAL);
                    if (st != IDA_ROOT_RETURN) break;  // roots are reported, integration continues
                }
                if (st < 0) break;
                memcpy(yy_out + ((size_t)k * nsys + s) * n, ida.nlp.ida_yy.data(), sizeof(double) * n);
                memcpy(yp_out + ((size_t)k * nsys + s) * n, ida.nlp.ida_yp.data(), sizeof(double) * n);
            }
            status[s] = st;
            double* cn = counters + (size_t)s * 8;
            cn[0] = (double)ida.ida_nst; cn[1] = (double)ida.nlp.ida_nre; cn[2] = (double)ida.nlp.lp.nje;
            cn[3] = (double)ida.nlp.ida_nsetups; cn[4] = (double)ida.nls.niters; cn[5] = (double)ida.ida_netf;
            cn[6] = (double)ida.ida_ncfn; cn[7] = (double)ida.n_attempts;
            last_k_h[2 * s + 0] = (double)ida.ida_kused;
            last_k_h[2 * s + 1] = ida.ida_hused;
        }
    };
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int t = 1; t < nthreads; ++t) th.emplace_back(worker, t);
    worker(0);
    for (auto& t : th) t.join();
    const auto t1 = std::chrono::steady_clock::now();
    return std::chrono::duration<double>(t1 - t0).count();
}

// Timed kernels for the cpu_baseline leg: LU + solve on `nsys` matrices [nsys][n*n] (in place), rhs [nsys][n].
double oracle_time_lu_solve(double* a, double* b, int n, int nsys, int nthreads, int* info) {
    const size_t nn = (size_t)n * n;
    auto worker = [&](int tid) {
        std::vector<int64_t> piv(n);
        for (int s = tid; s < nsys; s += nthreads) {
            info[s] = dense_get_rf(a + s * nn, n, n, piv.data());
            if (!info[s]) dense_get_rs(a + s * nn, n, piv.data(), b + (size_t)s * n);
        }
    };
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int t = 1; t < nthreads; ++t) th.emplace_back(worker, t);
    worker(0);
    for (auto& t : th) t.join();
    const auto t1 = std::chrono::steady_clock::now();
    return std::chrono::duration<double>(t1 - t0).count();
}

int oracle_hardware_concurrency() { return (int)std::thread::hardware_concurrency(); }

}  // extern "C"
