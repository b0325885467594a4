// Elementwise / norm kernels for the vector parts of the stepper that share the device-resident IDA state
// (SURVEY.md 8(f)-1). One workgroup per listed system; all sums that feed decisions are accumulated left to right
// by one lane so they carry the reference's bits.
//   ewt_set        <- /root/reference/src/tol_control.rs:36-44, 71-82
//   set_coeffs phi <- /root/reference/src/lib.rs:768-779
//   predict        <- /root/reference/src/lib.rs:894-959
//   final yy/yp    <- /root/reference/src/lib.rs:845-849
//   test_error     <- /root/reference/src/lib.rs:983-1004 (vector part)
//   restore        <- /root/reference/src/lib.rs:1057-1082
//   complete_step  <- /root/reference/src/impl_complete_step.rs:74-77, 152-176; /root/reference/src/lib.rs:708
//   get_solution   <- /root/reference/src/lib.rs:1319-1340
#pragma once
#include "common.hpp"
#include "solve_kernels.hpp"

namespace idahip {

struct VecState {
    double* phi;  // [6][batch][n]
    long phistride;  // batch*n
    double *yy, *yp, *yypredict, *yppredict, *ewt, *ee, *delta;
    int n;
    double rtol, atol_s;
    const double* atol_v;
};

__device__ __forceinline__ double ewt_of(const VecState& s, double y, int i) {
    const double at = s.atol_v ? s.atol_v[i] : s.atol_s;
    return 1.0 / (s.rtol * fabs(y) + at);
}

// ewt = ewt_set(phi[0]); out[2s] = sum (phi[1]*ewt)^2; out[2s+1] = sum (phi[0]*ewt)^2
__global__ __launch_bounds__(256) void init_first_kernel(VecState s, const int* __restrict__ idx, double* __restrict__ out) {
    extern __shared__ __align__(16) double sm[];
    const int b = idx[blockIdx.x];
    const long vb = (long)b * s.n;
    for (int i = threadIdx.x; i < s.n; i += 256) {
        const double y = s.phi[vb + i];
        const double e = ewt_of(s, y, i);
        s.ewt[vb + i] = e;
        const double p = s.phi[s.phistride + vb + i] * e;
        sm[i] = p * p;
        const double q = y * e;
        sm[s.n + i] = q * q;
    }
    __syncthreads();
    if (threadIdx.x < 2) out[2 * blockIdx.x + threadIdx.x] = seq_sum_lds(sm + threadIdx.x * s.n, s.n);
}

__global__ __launch_bounds__(256) void scale_phi1_kernel(VecState s, const int* __restrict__ idx, const double* __restrict__ fac) {
    const int b = idx[blockIdx.x];
    const long vb = (long)b * s.n;
    const double f = fac[blockIdx.x];
    for (int i = threadIdx.x; i < s.n; i += 256) s.phi[s.phistride + vb + i] *= f;
}

// phi[j] *= beta[j] (j = ns..kk) when ns <= kk; yypredict = sum_{j<=kk} phi[j]; yppredict = sum_{1<=j<=kk} gamma[j]*phi[j]
__global__ __launch_bounds__(256) void predict_kernel(VecState s, const int* __restrict__ idx, const int* __restrict__ kkns,
                                                      const double* __restrict__ beta, const double* __restrict__ gamma) {
    const int b = idx[blockIdx.x];
    const long vb = (long)b * s.n;
    const int kk = kkns[2 * blockIdx.x], ns = kkns[2 * blockIdx.x + 1];
    double bt[MXORDP1], gm[MXORDP1];
#pragma unroll
    for (int j = 0; j < MXORDP1; ++j) {
        bt[j] = beta[MXORDP1 * blockIdx.x + j];
        gm[j] = gamma[MXORDP1 * blockIdx.x + j];
    }
    for (int i = threadIdx.x; i < s.n; i += 256) {
        double yyp = 0.0, ypp = 0.0;
#pragma unroll
        for (int j = 0; j < MXORDP1; ++j) {
            if (j <= kk) {
                double p = s.phi[j * s.phistride + vb + i];
                if (j >= ns) {
                    p *= bt[j];
                    s.phi[j * s.phistride + vb + i] = p;
                }
                yyp = yyp + p;
                if (j >= 1) ypp = ypp + gm[j] * p;
            }
        }
        s.yypredict[vb + i] = yyp;
        s.yppredict[vb + i] = ypp;
    }
}

// yy = yypredict + ee; yp = yppredict + cj*ee; four sequential squared-norm sums
__global__ __launch_bounds__(256) void post_newton_kernel(VecState s, const int* __restrict__ idx, const double* __restrict__ cjs,
                                                          const int* __restrict__ kks, double* __restrict__ out) {
    extern __shared__ __align__(16) double sm[];
    const int n = s.n;
    const int b = idx[blockIdx.x];
    const long vb = (long)b * n;
    const double cj = cjs[blockIdx.x];
    const int kk = kks[blockIdx.x];
    for (int i = threadIdx.x; i < n; i += 256) {
        const double e = s.ee[vb + i];
        const double w = s.ewt[vb + i];
        s.yy[vb + i] = s.yypredict[vb + i] + e;
        s.yp[vb + i] = s.yppredict[vb + i] + cj * e;
        double p = e * w;
        sm[i] = p * p;
        double d = 0.0;
        if (kk > 1) {
            d = s.phi[kk * s.phistride + vb + i] + e;  // delta = phi[kk] + ee   (lib.rs:992)
            p = d * w;
            sm[n + i] = p * p;
        } else {
            sm[n + i] = 0.0;
        }
        if (kk > 2) {
            d = d + s.phi[(kk - 1) * s.phistride + vb + i];  // delta += phi[kk-1]  (lib.rs:1002)
            p = d * w;
            sm[2 * n + i] = p * p;
        } else {
            sm[2 * n + i] = 0.0;
        }
        if (kk + 1 < MXORDP1) {
            const double tmp = e - s.phi[(kk + 1) * s.phistride + vb + i];  // ee - phi[kk+1]  (impl_complete_step.rs:75)
            p = tmp * w;
            sm[3 * n + i] = p * p;
        } else {
            sm[3 * n + i] = 0.0;
        }
    }
    __syncthreads();
    if (threadIdx.x < 4) out[4 * blockIdx.x + threadIdx.x] = seq_sum_lds(sm + threadIdx.x * n, n);
}

// phi[j] *= cvals[j-ns], j = ns..kk
__global__ __launch_bounds__(256) void restore_kernel(VecState s, const int* __restrict__ idx, const int* __restrict__ kkns,
                                                      const double* __restrict__ cvals) {
    const int b = idx[blockIdx.x];
    const long vb = (long)b * s.n;
    const int kk = kkns[2 * blockIdx.x], ns = kkns[2 * blockIdx.x + 1];
    if (ns > kk) return;
    for (int i = threadIdx.x; i < s.n; i += 256) {
        for (int j = ns; j <= kk; ++j) s.phi[j * s.phistride + vb + i] *= cvals[MXORDP1 * blockIdx.x + (j - ns)];
    }
}

// phi[kused+1] = ee (kused < maxord); tmp = ee; for j = kused..0: tmp += phi[j]; phi[j] = tmp; ee *= ck;
// then ewt = ewt_set(phi[0]) and out = sum (phi[0]*ewt)^2
__global__ __launch_bounds__(256) void complete_step_kernel(VecState s, const int* __restrict__ idx, const int* __restrict__ kused_a,
                                                            const double* __restrict__ cks, int maxord, double* __restrict__ out,
                                                            int* __restrict__ ewtbad) {
    extern __shared__ __align__(16) double sm[];
    __shared__ int s_bad;
    const int n = s.n;
    const int b = idx[blockIdx.x];
    const long vb = (long)b * n;
    const int kused = kused_a[blockIdx.x];
    const double ck = cks[blockIdx.x];
    if (threadIdx.x == 0) s_bad = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256) {
        const double e = s.ee[vb + i];
        if (kused < maxord) s.phi[(kused + 1) * s.phistride + vb + i] = e;
        double tmp = e;
        for (int j = kused; j >= 0; --j) {
            tmp = tmp + s.phi[j * s.phistride + vb + i];
            s.phi[j * s.phistride + vb + i] = tmp;
        }
        s.ee[vb + i] = e * ck;
        const double w = ewt_of(s, tmp, i);  // tmp == new phi[0]
        s.ewt[vb + i] = w;
        if (!(w > 0.0)) s_bad = 1;
        const double p = tmp * w;
        sm[i] = p * p;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[blockIdx.x] = seq_sum_lds(sm, n);
        ewtbad[blockIdx.x] = s_bad;
    }
}

// yy = sum_{j<=kord} cvals[j]*phi[j]; yp = sum_{1<=j<=kord} dvals[j-1]*phi[j]  (from zero, scaled_add)
__global__ __launch_bounds__(256) void get_solution_kernel(VecState s, const int* __restrict__ idx, const int* __restrict__ kords,
                                                           const double* __restrict__ cvals, const double* __restrict__ dvals) {
    const int b = idx[blockIdx.x];
    const long vb = (long)b * s.n;
    const int kord = kords[blockIdx.x];
    for (int i = threadIdx.x; i < s.n; i += 256) {
        double y = 0.0, yp = 0.0;
        for (int j = 0; j <= kord; ++j) {
            const double p = s.phi[j * s.phistride + vb + i];
            y = y + cvals[MXORDP1 * blockIdx.x + j] * p;
            if (j >= 1) yp = yp + dvals[5 * blockIdx.x + (j - 1)] * p;
        }
        s.yy[vb + i] = y;
        s.yp[vb + i] = yp;
    }
}

// IDAGetDky (lib.rs:517-526): dky = sum_{j = k .. kused} cjk[j] * phi[j], from zero in ascending j, product then sum
__global__ __launch_bounds__(256) void get_dky_kernel(VecState s, const int* __restrict__ idx, const int* __restrict__ kfirst,
                                                      const int* __restrict__ klast, const double* __restrict__ cjk,
                                                      double* __restrict__ out) {
    const int b = idx[blockIdx.x];
    const long vb = (long)b * s.n;
    const int k0 = kfirst[blockIdx.x], k1 = klast[blockIdx.x];
    for (int i = threadIdx.x; i < s.n; i += 256) {
        double d = 0.0;
        for (int j = k0; j <= k1; ++j) d = d + s.phi[j * s.phistride + vb + i] * cjk[MXORDP1 * blockIdx.x + j];
        out[(long)blockIdx.x * s.n + i] = d;
    }
}

// Ida::new for the listed systems (lib.rs:291-293, ida_nls.rs:83-84): phi[0] = yy = y0, phi[1] = yp = y0'
__global__ __launch_bounds__(256) void restore_initial_kernel(VecState s, const double* __restrict__ icy, const double* __restrict__ icyp,
                                                              const int* __restrict__ idx) {
    const int b = idx[blockIdx.x];
    const long vb = (long)b * s.n;
    for (int i = threadIdx.x; i < s.n; i += 256) {
        const double y = icy[vb + i], yp = icyp[vb + i];
        s.phi[vb + i] = y;
        s.phi[s.phistride + vb + i] = yp;
        s.yy[vb + i] = y;
        s.yp[vb + i] = yp;
    }
}

}  // namespace idahip
