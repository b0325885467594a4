// ORACLE (test infrastructure, never shipped / never on the product path).
//
// User problems F(t, y, y') = 0 with analytic Jacobian J = dF/dy + cj * dF/dy'.
//   trait surface  <- /root/reference/src/traits.rs:12-94 (ModelSpec/Residual/Jacobian/Root)
//   Roberts        <- /root/reference/src/sample_problems/roberts.rs:47-113 (op order kept verbatim)
//   Lorenz63       <- /root/reference/tests/lorenz63.rs:17-25,47-53 (parameters + commented RHS only;
//                     the reference has no residual/Jacobian code, ICs or expected values for it:
//                     PARITY UNPINNED for this problem -- the expression order below is this build's)
//   LinearDense    <- not in the reference: F = A y' + B y - c. PARITY UNPINNED.
//   Heat1D         <- not in the reference. PARITY UNPINNED.
//
// Jacobians are written column-major (J(i,j) = jac[j*n + i]), the layout of the gen-B dense solver
// (crates/linear/src/dense.rs:108). The gen-A Roberts code indexes `jac[[row, col]]` (roberts.rs:80-90).
#pragma once
#include <cstddef>
#include <vector>

namespace oracle {

struct Problem {
    virtual ~Problem() {}
    virtual int model_size() const = 0;
    virtual void res(double tt, const double* yy, const double* yp, double* rr) const = 0;
    virtual void jac(double tt, double cj, const double* yy, const double* yp, const double* rr,
                     double* jac /* col-major n*n, pre-zeroed by the caller (ida_ls.rs:252-255) */) const = 0;
    virtual int num_roots() const { return 0; }
    virtual void root(double, const double*, const double*, double*) const {}
};

// roberts.rs:47-113
struct Roberts : Problem {
    int model_size() const override { return 3; }
    void res(double, const double* yy, const double* yp, double* r) const override {
        r[0] = -0.04 * yy[0] + 1.0e4 * yy[1] * yy[2];
        r[1] = -r[0] - 3.0e7 * yy[1] * yy[1] - yp[1];
        r[0] -= yp[0];
        r[2] = yy[0] + yy[1] + yy[2] - 1.0;
    }
    void jac(double, double cj, const double* yy, const double*, const double*, double* J) const override {
        const int n = 3;
        J[0 * n + 0] = -0.04 - cj;
        J[1 * n + 0] = 1.0e4 * yy[2];
        J[2 * n + 0] = 1.0e4 * yy[1];
        J[0 * n + 1] = 0.04;
        J[1 * n + 1] = -1.0e4 * yy[2] - 6.0e7 * yy[1] - cj;
        J[2 * n + 1] = -1.0e4 * yy[1];
        J[0 * n + 2] = 1.0;
        J[1 * n + 2] = 1.0;
        J[2 * n + 2] = 1.0;
    }
    int num_roots() const override { return 2; }
    void root(double, const double* y, const double*, double* g) const override {
        g[0] = y[0] - 0.0001;
        g[1] = y[2] - 0.01;
    }
};

// tests/lorenz63.rs:17-25 (defaults 10, 28, 8/3), RHS from the comments at :47-53.
struct Lorenz63 : Problem {
    double p = 10.0, r = 28.0, b = 8.0 / 3.0;
    int model_size() const override { return 3; }
    void res(double, const double* y, const double* yp, double* f) const override {
        f[0] = yp[0] - p * (y[1] - y[0]);
        f[1] = yp[1] - (y[0] * (r - y[2]) - y[1]);
        f[2] = yp[2] - (y[0] * y[1] - b * y[2]);
    }
    void jac(double, double cj, const double* y, const double*, const double*, double* J) const override {
        const int n = 3;
        J[0 * n + 0] = p + cj;
        J[1 * n + 0] = -p;
        J[2 * n + 0] = 0.0;
        J[0 * n + 1] = -(r - y[2]);
        J[1 * n + 1] = 1.0 + cj;
        J[2 * n + 1] = y[0];
        J[0 * n + 2] = -y[1];
        J[1 * n + 2] = -y[0];
        J[2 * n + 2] = b + cj;
    }
};

// F = A y' + B y - c, A/B column-major n x n. Defined summation order (this build's choice, mirrored
// bit-for-bit by the HIP kernel): two independent left-to-right chains over ascending column j,
//   ra_i = (((0 + A_i0*yp_0) + A_i1*yp_1) + ...),  rb_i likewise with B, y;  F_i = (ra_i + rb_i) - c_i.
struct LinearDense : Problem {
    int n = 0;
    const double* A = nullptr;  // not owned
    const double* B = nullptr;
    const double* c = nullptr;
    int model_size() const override { return n; }
    void res(double, const double* yy, const double* yp, double* r) const override {
        std::vector<double> ra(n, 0.0), rb(n, 0.0);
        for (int j = 0; j < n; ++j) {
            const double* Aj = A + (size_t)j * n;
            const double* Bj = B + (size_t)j * n;
            const double ypj = yp[j], yyj = yy[j];
            for (int i = 0; i < n; ++i) {
                ra[i] = ra[i] + Aj[i] * ypj;
                rb[i] = rb[i] + Bj[i] * yyj;
            }
        }
        for (int i = 0; i < n; ++i) r[i] = (ra[i] + rb[i]) - c[i];
    }
    void jac(double, double cj, const double*, const double*, const double*, double* J) const override {
        const size_t nn = (size_t)n * n;
        for (size_t e = 0; e < nn; ++e) J[e] = B[e] + cj * A[e];
    }
};

// 1-D heat equation, method of lines, Dirichlet ends as algebraic equations.
//   F_0 = y_0, F_{n-1} = y_{n-1}, F_i = y'_i - coef*((y_{i-1} - 2 y_i) + y_{i+1}),  coef = kappa/dx^2.
// coef is an input (computed once by the caller) so that oracle and device use the same bits.
struct Heat1D : Problem {
    int n = 0;
    double coef = 0.0;
    int model_size() const override { return n; }
    void res(double, const double* y, const double* yp, double* f) const override {
        f[0] = y[0];
        for (int i = 1; i + 1 < n; ++i) f[i] = yp[i] - coef * ((y[i - 1] - 2.0 * y[i]) + y[i + 1]);
        f[n - 1] = y[n - 1];
    }
    void jac(double, double cj, const double*, const double*, const double*, double* J) const override {
        J[0] = 1.0;
        for (int i = 1; i + 1 < n; ++i) {
            J[(size_t)(i - 1) * n + i] = -coef;
            J[(size_t)i * n + i] = cj + 2.0 * coef;
            J[(size_t)(i + 1) * n + i] = -coef;
        }
        J[(size_t)(n - 1) * n + (n - 1)] = 1.0;
    }
};

}  // namespace oracle
