/* A C (not C++, not Python) consumer of include/ida_hip.h: the first compile-and-link proof of the header outside C++.
 * Built by tests/test_c_consumer.py with `gcc -std=c99 -Wall -Wextra -pedantic -Werror` against libidahip.so.
 *
 * It plays the reference's LSolver / NormRms call sites for batch = 2 systems of n = 3:
 *   LSolver::setup  (crates/linear/src/traits.rs:52-57, dense.rs:38-44)   -> idahip_ls_setup
 *   LSolver::solve  (crates/linear/src/traits.rs:59-80, dense.rs:46-63)   -> idahip_ls_solve
 *   NormRms::norm_wrms (src/norm_rms.rs:31-38)                            -> idahip_wrms
 * Input: one line per array on stdin (the golden inputs of the reference, written by the test from the JSON fixtures);
 * output: the results as C99 hex floats, one array per line, compared by the test with the reference's expected values. */
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>

#include "ida_hip.h"

#define N 3
#define B 2

static int read_doubles(double* v, int count) {
    int i;
    for (i = 0; i < count; ++i)
        if (scanf("%la", &v[i]) != 1) return -1;
    return 0;
}
static void print_doubles(const char* tag, const double* v, int count) {
    int i;
    printf("%s", tag);
    for (i = 0; i < count; ++i) printf(" %a", v[i]);
    printf("\n");
}
static int check(idahip_ctx* ctx, int rc, const char* what) {
    if (rc < 0) {
        fprintf(stderr, "%s failed (%d): %s\n", what, rc, idahip_last_error(ctx));
        exit(2);
    }
    return rc;
}

int main(void) {
    double a[B * N * N], lu_in[B * N * N], b[B * N], x[B * N], w[B * N], lu_out[B * N * N], sol[B * N], nrm[B];
    int64_t piv_in[B * N], piv_out[B * N];
    int32_t info[B], idx[B] = {0, 1};
    idahip_ctx* ctx = NULL;
    double *dA, *dLU, *dB, *dX, *dW;
    int64_t *dPiv, *dPiv2;
    int i;

    /* column-major matrices to factor, then factored matrices + pivots + right-hand sides to solve, then x, w for the norm */
    if (read_doubles(a, B * N * N) || read_doubles(lu_in, B * N * N)) return 3;
    for (i = 0; i < B * N; ++i) {
        double p;
        if (scanf("%la", &p) != 1) return 3;
        piv_in[i] = (int64_t)p;
    }
    if (read_doubles(b, B * N) || read_doubles(x, B * N) || read_doubles(w, B * N)) return 3;

    if (idahip_create(&ctx, 0, N, B, IDAHIP_LORENZ63, NULL) != 0 || !ctx) {
        fprintf(stderr, "idahip_create failed\n");
        return 2;
    }
    dA = (double*)idahip_dev_alloc(ctx, sizeof a);
    dLU = (double*)idahip_dev_alloc(ctx, sizeof lu_in);
    dB = (double*)idahip_dev_alloc(ctx, sizeof b);
    dX = (double*)idahip_dev_alloc(ctx, sizeof x);
    dW = (double*)idahip_dev_alloc(ctx, sizeof w);
    dPiv = (int64_t*)idahip_dev_alloc(ctx, sizeof piv_out);
    dPiv2 = (int64_t*)idahip_dev_alloc(ctx, sizeof piv_in);
    if (!dA || !dLU || !dB || !dX || !dW || !dPiv || !dPiv2) return 2;

    /* LSolver::setup */
    check(ctx, idahip_memcpy_h2d(ctx, dA, a, sizeof a), "h2d");
    check(ctx, idahip_ls_setup(ctx, dA, dPiv, info, idx, B), "idahip_ls_setup");
    check(ctx, idahip_memcpy_d2h(ctx, lu_out, dA, sizeof lu_out), "d2h");
    check(ctx, idahip_memcpy_d2h(ctx, piv_out, dPiv, sizeof piv_out), "d2h");
    print_doubles("lu", lu_out, B * N * N);
    printf("piv");
    for (i = 0; i < B * N; ++i) printf(" %" PRId64, piv_out[i]);
    printf("\ninfo %d %d\n", (int)info[0], (int)info[1]);

    /* LSolver::solve */
    check(ctx, idahip_memcpy_h2d(ctx, dLU, lu_in, sizeof lu_in), "h2d");
    check(ctx, idahip_memcpy_h2d(ctx, dPiv2, piv_in, sizeof piv_in), "h2d");
    check(ctx, idahip_memcpy_h2d(ctx, dB, b, sizeof b), "h2d");
    check(ctx, idahip_ls_solve(ctx, dLU, dPiv2, dX, dB, 0.0, idx, B), "idahip_ls_solve");
    check(ctx, idahip_memcpy_d2h(ctx, sol, dX, sizeof sol), "d2h");
    print_doubles("x", sol, B * N);

    /* NormRms::norm_wrms */
    check(ctx, idahip_memcpy_h2d(ctx, dX, x, sizeof x), "h2d");
    check(ctx, idahip_memcpy_h2d(ctx, dW, w, sizeof w), "h2d");
    check(ctx, idahip_wrms(ctx, dX, dW, nrm, idx, B), "idahip_wrms");
    print_doubles("wrms", nrm, B);

    idahip_dev_free(ctx, dA); idahip_dev_free(ctx, dLU); idahip_dev_free(ctx, dB); idahip_dev_free(ctx, dX);
    idahip_dev_free(ctx, dW); idahip_dev_free(ctx, dPiv); idahip_dev_free(ctx, dPiv2);
    idahip_destroy(ctx);
    return 0;
}
