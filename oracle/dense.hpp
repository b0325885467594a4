// ORACLE (test infrastructure, never shipped / never on the product path).
//
// CPU restatement of the reference's dense direct linear solver and WRMS norm.
//   dense_get_rf  <- /root/reference/crates/linear/src/dense.rs:86-158
//   dense_get_rs  <- /root/reference/crates/linear/src/dense.rs:165-206
//   norm_wrms     <- /root/reference/src/norm_rms.rs:31-38  (gen-A, ndarray)
//                    /root/reference/crates/nonlinear/src/norm_wrms.rs:19-34 (gen-B, nalgebra)
//
// Arithmetic conventions: fp64, no FMA contraction (build with
// -ffp-contract=off), reciprocal-then-multiply for the LU multipliers, true division in the
// back substitution, left-to-right sequential sums starting from 0.0.
//
// Parity pinned by the reference's own unit goldens (tests/golden/dense_goldens.json,
// extracted from dense.rs:208-329 and norm_rms.rs:64-70) -- see tests/test_oracle_dense.py.
#pragma once
#include <cmath>
#include <cstdint>
#include <utility>

namespace oracle {

// Matrices are column-major (nalgebra convention, dense.rs:108 `mat_a.column(k)`):
// element (i, j) of an m x n matrix is a[j * m + i].

/// LU factorisation with partial (row) pivoting, in place.
/// Returns 0 on success, or k+1 (1-based column) when a zero pivot is met (dense.rs:120-122).
inline int dense_get_rf(double* a, int m, int n, int64_t* pivot) {
    for (int k = 0; k < n; ++k) {
        double* col_k = a + (size_t)k * m;

        // find l = pivot row number; strict '>' keeps the lowest row on ties (dense.rs:111-117)
        int l = k;
        for (int i = k + 1; i < m; ++i) {
            if (std::fabs(col_k[i]) > std::fabs(col_k[l])) l = i;
        }
        pivot[k] = l;

        // check for zero pivot element (dense.rs:120-122)
        if (col_k[l] == 0.0) return k + 1;

        // swap a(k,1:n) and a(l,1:n) if necessary -- full rows, all n columns (dense.rs:126-130)
        if (l != k) {
            for (int i = 0; i < n; ++i) std::swap(a[(size_t)i * m + k], a[(size_t)i * m + l]);
        }

        // multipliers: a(i,k) *= 1/a(k,k)  (reciprocal first, dense.rs:134-137)
        const double mult = 1.0 / a[(size_t)k * m + k];
        for (int i = k + 1; i < m; ++i) a[(size_t)k * m + i] *= mult;

        // column-oriented trailing update, skipped when a(k,j) == 0 (dense.rs:142-154)
        for (int j = k + 1; j < n; ++j) {
            const double a_kj = a[(size_t)j * m + k];
            if (a_kj != 0.0) {
                double* col_j = a + (size_t)j * m;
                for (int i = k + 1; i < m; ++i) {
                    const double a_ik = col_k[i];
                    col_j[i] -= a_kj * a_ik;  // unfused: mul then sub
                }
            }
        }
    }
    return 0;
}

/// Solve A x = b given the LU factors and pivots; solution overwrites b (dense.rs:165-206).
inline void dense_get_rs(const double* a, int n, const int64_t* pivot, double* b) {
    // Permute b, sequential swaps k = 0..n-1 (dense.rs:181-185)
    for (int k = 0; k < n; ++k) {
        const int64_t pk = pivot[k];
        if (pk != k) std::swap(b[k], b[pk]);
    }
    // Solve Ly = b, unit diagonal, column oriented (dense.rs:188-194)
    for (int k = 0; k + 1 < n; ++k) {
        const double* col_k = a + (size_t)k * n;
        const double bk = b[k];
        for (int i = k + 1; i < n; ++i) b[i] -= col_k[i] * bk;
    }
    // Solve Ux = y, column oriented, true division (dense.rs:197-205)
    for (int k = n - 1; k >= 1; --k) {
        const double* col_k = a + (size_t)k * n;
        b[k] /= col_k[k];
        const double bk = b[k];
        for (int i = 0; i < k; ++i) b[i] -= col_k[i] * bk;
    }
    b[0] /= a[0];
}

/// Weighted root-mean-square norm: sqrt( sum_i (x_i*w_i)^2 / N ), sequential sum from 0.0,
/// divide then sqrt (norm_rms.rs:31-38).
inline double norm_wrms(const double* x, const double* w, int n) {
    double acc = 0.0;
    for (int i = 0; i < n; ++i) {
        const double p = x[i] * w[i];
        acc = acc + p * p;  // powi(2) == p*p
    }
    return std::sqrt(acc / (double)n);
}

/// Masked variant (norm_rms.rs:49-57): mask multiplies the product (1.0 / 0.0).
inline double norm_wrms_masked(const double* x, const double* w, const uint8_t* id, int n) {
    double acc = 0.0;
    for (int i = 0; i < n; ++i) {
        const double p = (x[i] * w[i]) * (id[i] ? 1.0 : 0.0);
        acc = acc + p * p;
    }
    return std::sqrt(acc / (double)n);
}

}  // namespace oracle
