// Device residual / Jacobian kernels: the `Residual::res` / `Jacobian::jac` user callbacks of
// /root/reference/src/traits.rs:12-70, fused with idaNlsResidual (/root/reference/src/ida_nls.rs:118-153):
//     yy = yypredict + ycor;  yp = yppredict + cj*ycor (mul, then add);  delta = savres = F(tn, yy, yp).
// ycor is the accumulated Newton correction `ee` (zeroed first when reset_ee is set).
//
//   Roberts      <- /root/reference/src/sample_problems/roberts.rs:47-91 (operation order kept verbatim)
//   Lorenz63     <- /root/reference/tests/lorenz63.rs:17-25,47-53 (parameters + commented RHS; expression order ours)
//   LinearDense  <- SURVEY.md 8(d) config 3: F = A y' + B y - c with the summation order fixed in
//                   oracle/problems.hpp (two left-to-right chains over ascending column j) -- which is exactly
//                   what a thread-per-row, column-sweeping GPU kernel computes, with coalesced 16-byte loads.
//   Heat1D       <- SURVEY.md 8(d) config 4.
#pragma once
#include "common.hpp"

namespace idahip {

struct SysArgs {
    const double* yypredict;
    const double* yppredict;
    double* yy;
    double* yp;
    double* ee;
    double* delta;
    double* savres;
    const int* idx;
    const double* tn;  // [nsys]
    const double* cj;  // [nsys]
    int n;
    int reset_ee;
    const int* skip = nullptr;  // optional, per list position: nonzero = leave this system alone (idahip_newton_iter2)
};

// ------------------------------------------------------------------------------------------------ tiny problems
__device__ __forceinline__ void roberts_res(const double* yy, const double* yp, double* r) {
    r[0] = -0.04 * yy[0] + 1.0e4 * yy[1] * yy[2];
    r[1] = -r[0] - 3.0e7 * yy[1] * yy[1] - yp[1];
    r[0] -= yp[0];
    r[2] = yy[0] + yy[1] + yy[2] - 1.0;
}
__device__ __forceinline__ void roberts_jac(double cj, const double* yy, double* J) {
    J[0] = -0.04 - cj;               J[3] = 1.0e4 * yy[2];                          J[6] = 1.0e4 * yy[1];
    J[1] = 0.04;                     J[4] = -1.0e4 * yy[2] - 6.0e7 * yy[1] - cj;    J[7] = -1.0e4 * yy[1];
    J[2] = 1.0;                      J[5] = 1.0;                                    J[8] = 1.0;
}
__device__ __forceinline__ void lorenz_res(const double* prm, const double* y, const double* yp, double* f) {
    const double p = prm[0], r = prm[1], b = prm[2];
    f[0] = yp[0] - p * (y[1] - y[0]);
    f[1] = yp[1] - (y[0] * (r - y[2]) - y[1]);
    f[2] = yp[2] - (y[0] * y[1] - b * y[2]);
}
__device__ __forceinline__ void lorenz_jac(const double* prm, double cj, const double* y, double* J) {
    const double p = prm[0], r = prm[1], b = prm[2];
    J[0] = p + cj;        J[3] = -p;         J[6] = 0.0;
    J[1] = -(r - y[2]);   J[4] = 1.0 + cj;   J[7] = y[0];
    J[2] = -y[1];         J[5] = -y[0];      J[8] = b + cj;
}

template <int KIND>
__global__ void tiny_sys_kernel(SysArgs a, const double* __restrict__ params, int nparam, int nsys) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsys) return;
    if (a.skip && a.skip[s] != 0) return;
    const int b = a.idx[s];
    const long vb = (long)b * 3;
    const double cj = a.cj[s];
    double yy[3], yp[3], r[3];
    for (int i = 0; i < 3; ++i) {
        double yc = a.ee[vb + i];
        if (a.reset_ee) {
            yc = 0.0;
            a.ee[vb + i] = 0.0;
        }
        yy[i] = a.yypredict[vb + i] + yc;
        yp[i] = a.yppredict[vb + i] + cj * yc;
        a.yy[vb + i] = yy[i];
        a.yp[vb + i] = yp[i];
    }
    if (KIND == IDAHIP_ROBERTS) roberts_res(yy, yp, r);
    else lorenz_res(params + (long)b * nparam, yy, yp, r);
    for (int i = 0; i < 3; ++i) {
        a.delta[vb + i] = r[i];
        a.savres[vb + i] = r[i];
    }
}

// J <- 0; jac(tn, cj, yy, yp, res) -- into mats[b] (column-major)
template <int KIND>
__global__ void tiny_jac_kernel(double* mats, const double* __restrict__ yy, const double* __restrict__ params, int nparam,
                                const int* idx, const double* cjs, int nsys) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsys) return;
    const int b = idx[s];
    double y[3], J[9];
    for (int i = 0; i < 3; ++i) y[i] = yy[(long)b * 3 + i];
    if (KIND == IDAHIP_ROBERTS) roberts_jac(cjs[s], y, J);
    else lorenz_jac(params + (long)b * nparam, cjs[s], y, J);
    for (int e = 0; e < 9; ++e) mats[(long)b * 9 + e] = J[e];
}

// ------------------------------------------------------------------------------------------------ linear dense
// One workgroup (256 threads) per system; thread t owns rows {VEC*t + v + VEC*256*pass}; sweeps columns j ascending.
// WITH_JAC: the sweep also writes the Newton matrix J = B + cj*A (mul, then add -- as linear_jac_kernel) to Jout, column-
// major: when the reference's Newton::solve calls setup right after sys (call_lsetup), A and B are read once for both.
template <int VEC, bool WITH_JAC>
__global__ __launch_bounds__(256) void linear_sys_kernel(SysArgs a, const double* __restrict__ A, const double* __restrict__ Bm,
                                                         const double* __restrict__ C, double* __restrict__ Jout) {
    extern __shared__ __align__(16) double sm[];
    // columns in flight per thread: 16 x 2 matrices x 16 B = 512 B per thread, 128 KB per workgroup. (8: a lone workgroup -- the
    // later Newton iterations of a round serve 30-150 systems -- streamed at 35 GB/s and lived 119 us; 16: 72 GB/s, 58 us, and the
    // full launches gain 5 % at two workgroups per CU instead of four. 12, 20, 24 measured slower.)
    constexpr int UNR = tb::SYS_UNR;
    const int n = a.n;
    double* syy = sm;
    double* syp = sm + n;
    if (a.skip && a.skip[blockIdx.x] != 0) return;
    const int b = a.idx[blockIdx.x];
    const long vb = (long)b * n;
    const double cj = a.cj[blockIdx.x];
    const int t = threadIdx.x;
    for (int i = t; i < n; i += 256) {
        double yc = a.ee[vb + i];
        if (a.reset_ee) {
            yc = 0.0;
            a.ee[vb + i] = 0.0;
        }
        const double y = a.yypredict[vb + i] + yc;
        const double yp = a.yppredict[vb + i] + cj * yc;
        a.yy[vb + i] = y;
        a.yp[vb + i] = yp;
        syy[i] = y;
        syp[i] = yp;
    }
    __syncthreads();
    const double* __restrict__ Ab = A + (long)b * n * n;
    const double* __restrict__ Bb = Bm + (long)b * n * n;
    double* __restrict__ Jb = WITH_JAC ? Jout + (long)b * n * n : nullptr;
    for (int i = VEC * t; i < n; i += VEC * 256) {
        double ra[VEC], rb[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) ra[v] = rb[v] = 0.0;
        int j = 0;
        for (; j + UNR <= n; j += UNR) {
            double av[UNR][VEC], bv[UNR][VEC];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const double* pa = Ab + (long)(j + u) * n + i;
                const double* pb = Bb + (long)(j + u) * n + i;
                if constexpr (VEC == 2) {
                    const double2 qa = *reinterpret_cast<const double2*>(pa);
                    const double2 qb = *reinterpret_cast<const double2*>(pb);
                    av[u][0] = qa.x; av[u][1] = qa.y;
                    bv[u][0] = qb.x; bv[u][1] = qb.y;
                } else {
                    av[u][0] = *pa;
                    bv[u][0] = *pb;
                }
            }
            if constexpr (WITH_JAC) {
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    double* pj = Jb + (long)(j + u) * n + i;
                    if constexpr (VEC == 2) {
                        double2 o;
                        o.x = bv[u][0] + cj * av[u][0];
                        o.y = bv[u][1] + cj * av[u][1];
                        *reinterpret_cast<double2*>(pj) = o;
                    } else {
                        *pj = bv[u][0] + cj * av[u][0];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const double ypj = syp[j + u], yyj = syy[j + u];
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    ra[v] = ra[v] + av[u][v] * ypj;
                    rb[v] = rb[v] + bv[u][v] * yyj;
                }
            }
        }
        for (; j < n; ++j) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                if (i + v < n) {
                    const double ae = Ab[(long)j * n + i + v], be = Bb[(long)j * n + i + v];
                    if constexpr (WITH_JAC) Jb[(long)j * n + i + v] = be + cj * ae;
                    ra[v] = ra[v] + ae * syp[j];
                    rb[v] = rb[v] + be * syy[j];
                }
            }
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            if (i + v < n) {
                const double r = (ra[v] + rb[v]) - C[vb + i + v];
                a.delta[vb + i + v] = r;
                a.savres[vb + i + v] = r;
            }
        }
    }
}

// J = B + cj*A (mul, then add), elementwise over the listed systems
__global__ __launch_bounds__(256) void linear_jac_kernel(double* __restrict__ mats, const double* __restrict__ A, const double* __restrict__ Bm,
                                                         long nn, const int* __restrict__ idx, const double* __restrict__ cjs, int chunks) {
    const int b = idx[blockIdx.x];
    const double cj = cjs[blockIdx.x];
    const long per = (nn + chunks - 1) / chunks;
    const long beg = (long)blockIdx.y * per;
    const long end = (beg + per < nn) ? beg + per : nn;
    const double* __restrict__ Ab = A + (long)b * nn;
    const double* __restrict__ Bb = Bm + (long)b * nn;
    double* __restrict__ J = mats + (long)b * nn;
    if ((nn & 1) == 0 && (beg & 1) == 0) {
        for (long e = beg + 2 * threadIdx.x; e < end; e += 512) {
            if (e + 1 < end) {
                const double2 qa = *reinterpret_cast<const double2*>(Ab + e);
                const double2 qb = *reinterpret_cast<const double2*>(Bb + e);
                double2 o;
                o.x = qb.x + cj * qa.x;
                o.y = qb.y + cj * qa.y;
                *reinterpret_cast<double2*>(J + e) = o;
            } else {
                J[e] = Bb[e] + cj * Ab[e];
            }
        }
    } else {
        for (long e = beg + threadIdx.x; e < end; e += 256) J[e] = Bb[e] + cj * Ab[e];
    }
}

// ------------------------------------------------------------------------------------------------ heat 1-D
__global__ __launch_bounds__(256) void heat_sys_kernel(SysArgs a, const double* __restrict__ params) {
    extern __shared__ __align__(16) double sm[];
    const int n = a.n;
    double* syy = sm;
    if (a.skip && a.skip[blockIdx.x] != 0) return;
    const int b = a.idx[blockIdx.x];
    const long vb = (long)b * n;
    const double cj = a.cj[blockIdx.x];
    const double coef = params[b];
    for (int i = threadIdx.x; i < n; i += 256) {
        double yc = a.ee[vb + i];
        if (a.reset_ee) {
            yc = 0.0;
            a.ee[vb + i] = 0.0;
        }
        const double y = a.yypredict[vb + i] + yc;
        const double yp = a.yppredict[vb + i] + cj * yc;
        a.yy[vb + i] = y;
        a.yp[vb + i] = yp;
        syy[i] = y;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256) {
        double r;
        if (i == 0 || i == n - 1) r = syy[i];
        else r = a.yp[vb + i] - coef * ((syy[i - 1] - 2.0 * syy[i]) + syy[i + 1]);
        a.delta[vb + i] = r;
        a.savres[vb + i] = r;
    }
}

__global__ __launch_bounds__(256) void heat_jac_kernel(double* __restrict__ mats, int n, const double* __restrict__ params,
                                                       const int* __restrict__ idx, const double* __restrict__ cjs, int chunks,
                                                       const int* __restrict__ skip, const int* __restrict__ zeroed) {
    if (skip && skip[blockIdx.x] != 0) return;  // (the device lock-step stepper's per-system flags)
    const int b = idx[blockIdx.x];
    const double cj = cjs[blockIdx.x];
    const double coef = params[b];
    double* __restrict__ J = mats + (long)b * n * n;
    const int per = (n + chunks - 1) / chunks;  // columns per block
    const int jbeg = blockIdx.y * per;
    const int jend = (jbeg + per < n) ? jbeg + per : n;
    if (zeroed && zeroed[b] != 0) {
        // the last factorisation has left this system's matrix all +0.0 (LuWs::jwzero): J <- 0 (ida_ls.rs:254) is in place already, the
        // band is all there is to write -- 3 entries of a column instead of n
        for (int j = jbeg + threadIdx.x; j < jend; j += 256) {
            for (int i = (j > 0 ? j - 1 : 0); i <= j + 1 && i < n; ++i) {
                double v = 0.0;
                if (i == 0 || i == n - 1) {
                    v = (j == i) ? 1.0 : 0.0;
                } else {
                    if (j == i) v = cj + 2.0 * coef;
                    else v = -coef;
                }
                J[(long)j * n + i] = v;
            }
        }
        return;
    }
    for (int j = jbeg; j < jend; ++j) {
        for (int i = threadIdx.x; i < n; i += 256) {
            double v = 0.0;
            if (i == 0 || i == n - 1) {
                v = (j == i) ? 1.0 : 0.0;
            } else {
                if (j == i) v = cj + 2.0 * coef;
                else if (j == i - 1 || j == i + 1) v = -coef;
            }
            J[(long)j * n + i] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------ host-callback problems
// idaNlsResidual around a user residual that lives on the host (ida_nls.rs:118-153): `pre` forms yy, yp on the device and
// packs them for the listed systems, the host evaluates F, `post` scatters the residuals into delta and savres.
__global__ __launch_bounds__(256) void callback_pre_kernel(SysArgs a, double* __restrict__ stage) {
    const int n = a.n;
    if (a.skip && a.skip[blockIdx.x] != 0) return;
    const int b = a.idx[blockIdx.x];
    const long vb = (long)b * n;
    const double cj = a.cj[blockIdx.x];
    double* __restrict__ sy = stage + (long)blockIdx.x * 3 * n;
    for (int i = threadIdx.x; i < n; i += 256) {
        double yc = a.ee[vb + i];
        if (a.reset_ee) {
            yc = 0.0;
            a.ee[vb + i] = 0.0;
        }
        const double y = a.yypredict[vb + i] + yc;
        const double yp = a.yppredict[vb + i] + cj * yc;
        a.yy[vb + i] = y;
        a.yp[vb + i] = yp;
        sy[i] = y;
        sy[n + i] = yp;
    }
}
__global__ __launch_bounds__(256) void callback_post_kernel(SysArgs a, const double* __restrict__ stage) {
    const int n = a.n;
    if (a.skip && a.skip[blockIdx.x] != 0) return;
    const int b = a.idx[blockIdx.x];
    const long vb = (long)b * n;
    const double* __restrict__ r = stage + (long)blockIdx.x * 3 * n + 2 * n;
    for (int i = threadIdx.x; i < n; i += 256) {
        a.delta[vb + i] = r[i];
        a.savres[vb + i] = r[i];
    }
}
// current yy, yp, savres of the listed systems, packed for the user's jac
__global__ __launch_bounds__(256) void callback_pack_kernel(const double* __restrict__ yy, const double* __restrict__ yp,
                                                            const double* __restrict__ res, const int* __restrict__ idx, int n,
                                                            double* __restrict__ stage) {
    const int b = idx[blockIdx.x];
    const long vb = (long)b * n;
    double* __restrict__ sy = stage + (long)blockIdx.x * 3 * n;
    for (int i = threadIdx.x; i < n; i += 256) {
        sy[i] = yy[vb + i];
        sy[n + i] = yp[vb + i];
        sy[2 * n + i] = res[vb + i];
    }
}

// the user's Jacobians of a chunk of listed systems (staged contiguously, column-major) into their work matrices
__global__ __launch_bounds__(256) void callback_scatter_jac_kernel(const double* __restrict__ stage, double* __restrict__ work,
                                                                   const int* __restrict__ idx, long nn) {
    const double* __restrict__ src = stage + (long)blockIdx.x * nn;
    double* __restrict__ dst = work + (long)idx[blockIdx.x] * nn;
    for (long e = (long)blockIdx.y * 256 + threadIdx.x; e < nn; e += (long)gridDim.y * 256) dst[e] = src[e];
}

}  // namespace idahip
