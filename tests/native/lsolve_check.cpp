// Host-side unit check of idactl::lsolve_tol / after_lsolve (rust-ida_amd/host/ida_controller.hpp) for the three LSolverType
// values: prints one line per case, tests/test_lsolver_type.py compares them with what the text of
// /root/reference/src/ida_ls.rs:316-418 prescribes (tol :323-329, nli :389-400, 2/(1+cjratio) :405-410, ncfl :413-415).
#include <cmath>
#include <cstdio>
#include <initializer_list>

#include "../../include/ida_hip.h"
#include "ida_controller.hpp"

int main() {
    using namespace idactl;
    const int types[3] = {IDAHIP_LS_DIRECT, IDAHIP_LS_ITERATIVE, IDAHIP_LS_MATRIX_ITERATIVE};
    const double sqrt_n = std::sqrt(512.0);
    for (int t : types) {
        for (double cjratio : {1.0, 0.7}) {
            for (int failed = 0; failed < 2; ++failed) {
                SysCore s;
                s.cjratio = cjratio;
                s.nli = 10;
                s.ncfl = 3;
                const double tol = lsolve_tol(t, sqrt_n, EPLIFAC);
                const bool scale = after_lsolve(s, t, 7, failed != 0);
                std::printf("type %d cjratio %.17g failed %d tol %.17g nli %ld ncfl %ld scale %d\n", t, cjratio, failed, tol, s.nli, s.ncfl,
                            scale ? 1 : 0);
            }
        }
    }
    return 0;
}
