// Test driver (CPU): glibc_pow::pow (rust-ida_amd/csrc/glibc_pow.hpp, compiled for the host) against this machine's libm pow,
// bit for bit, on random arguments of the step-size controller's domain, on general arguments, on the special cases of C99
// and on results near overflow / underflow. usage: pow_check <cases>; exit status 0 = no mismatch.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include "glibc_pow.hpp"  // -I rust-ida_amd/csrc
int main(int argc, char** argv) {
    long n = argc > 1 ? atol(argv[1]) : 10000000;
    std::mt19937_64 g(12345);
    std::uniform_real_distribution<double> U(0, 1);
    long bad = 0;
    // controller domain: base in (1e-6, 1e6) log-uniform, exponents +-1/m, m = 1..6; plus general exponents
    for (long i = 0; i < n; ++i) {
        double x = std::exp((U(g) * 2 - 1) * 14.0);
        int m = 1 + (int)(U(g) * 6);
        double y = (i & 1) ? 1.0 / m : -(1.0 / m);
        if ((i & 7) == 7) y = (U(g) * 2 - 1) * 50.0;
        if ((i & 15) == 15) x = std::exp((U(g) * 2 - 1) * 700.0);
        volatile double xv = x, yv = y;
        double a = std::pow(xv, yv), b = glibc_pow::pow(x, y);
        if (memcmp(&a, &b, 8) != 0 && !(a != a && b != b)) {
            if (bad < 10) printf("MISMATCH x=%a y=%a glibc=%a mine=%a\n", x, y, a, b);
            ++bad;
        }
    }
    // specials
    double sp[] = {0.0, -0.0, 1.0, -1.0, 2.0, -2.0, 0.5, -0.5, INFINITY, -INFINITY, NAN, 5e-324, 1e-310, -1e-310, 3.0, -3.0, 1e300, 1e-300, 0x1p-66, 0x1p63, 1.0000000000000002, 0.9999999999999999, 709.0, -745.0, 1075.5, 7.0};
    int ns = sizeof sp / sizeof sp[0];
    for (int i = 0; i < ns; ++i) for (int j = 0; j < ns; ++j) {
        volatile double xv = sp[i], yv = sp[j];
        double a = std::pow(xv, yv), b = glibc_pow::pow(sp[i], sp[j]);
        if (memcmp(&a, &b, 8) != 0 && !(a != a && b != b)) { printf("SPECIAL x=%a y=%a glibc=%a mine=%a\n", sp[i], sp[j], a, b); ++bad; }
    }
    // results near over/underflow
    for (long i = 0; i < n / 10; ++i) {
        double x = std::exp((U(g) * 2 - 1) * 5.0), y = (U(g) * 2 - 1) * 800.0 / std::fabs(std::log(x) + 1e-3);
        volatile double xv = x, yv = y;
        double a = std::pow(xv, yv), b = glibc_pow::pow(x, y);
        if (memcmp(&a, &b, 8) != 0 && !(a != a && b != b)) { if (bad < 20) printf("EDGE x=%a y=%a glibc=%a mine=%a\n", x, y, a, b); ++bad; }
    }
    printf("%ld cases, %ld mismatches\n", n, bad);
    return bad != 0;
}
