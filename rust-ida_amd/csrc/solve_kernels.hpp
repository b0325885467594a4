// Batched triangular solves, the fused Newton-iteration body and the WRMS norm for gfx950.
//   dense_get_rs           <- /root/reference/crates/linear/src/dense.rs:165-206
//   Newton loop body       <- /root/reference/crates/nonlinear/src/newton.rs:98-110
//   idaNlsLSolve/idaLsSolve<- /root/reference/src/ida_nls.rs:190-215, /root/reference/src/ida_ls.rs:298-455
//   norm_wrms              <- /root/reference/src/norm_rms.rs:31-38
//
// Bit-exactness: each b_i receives b_i -= a_ik * b_k in ascending (forward) / descending (backward) k exactly as
// the reference's column-oriented loops do; unfused mul/sub; true division by u_kk. The WRMS sum is accumulated
// left to right by one lane (512 dependent adds ~ 2 us, hidden behind the other resident workgroups' streaming);
// the final sqrt(sum/N) is done by the host libm so the norm equals the reference's to the last bit.
//
// Performance model: the solve reads every LU entry once (8 N^2 B) in coalesced column segments, 16 B per lane;
// it is HBM-bound, so the kernel keeps registers low (8 workgroups of 256 threads per CU) and lets workgroups of
// different systems overlap the short serial diagonal phases with each other's streaming.
#pragma once
#include "common.hpp"

namespace idahip {

// Sequential sum of sq[0..n) by the calling lane (LDS source), left to right from 0.0.
__device__ __forceinline__ double seq_sum_lds(const double* sq, int n) {
    double s = 0.0;
    int i = 0;
    for (; i + 8 <= n; i += 8) {
        const double a0 = sq[i], a1 = sq[i + 1], a2 = sq[i + 2], a3 = sq[i + 3], a4 = sq[i + 4], a5 = sq[i + 5], a6 = sq[i + 6],
                     a7 = sq[i + 7];
        s = s + a0; s = s + a1; s = s + a2; s = s + a3; s = s + a4; s = s + a5; s = s + a6; s = s + a7;
    }
    for (; i < n; ++i) s = s + sq[i];
    return s;
}

// Workgroup-cooperative getrs on the right-hand side held in LDS (bs[0..n)). blockDim.x == T.
// VEC = 2 requires n even (16-byte aligned column segments).
// zmap (T > 256 only, may be null): the factorisation's map of 64 x 64 blocks, zmap[K * 64 + I] = 0 when block (rows I, columns
// K) of LU holds only zeros. A sweep then leaves out the rows of such a block -- b_i -= 0 * b_k changes nothing -- unless that
// is not exactly true: a b_k of the column block is not finite (0 * inf is NaN), or the row's b_i is -0.0 (which -0.0 - (-0.0)
// would turn into +0.0). The factors of a banded matrix (the heat equation's Jacobian) are zero almost everywhere, and the
// reference's solve multiplies through all of it; so did this kernel, 134 MB per system and iteration at n = 4096.
// dg (T > 256 only): 64 x 64 doubles of LDS for the diagonal block. With few, large systems per call a workgroup's life is
// its 2 n / 64 diagonal solves, and each of those was eight dependent round trips to memory (eight columns loaded, eight
// steps, ...): 17 us per block at n = 4096. The block a solve needs is instead fetched by all threads while the sweep before
// it streams (the factors are read-only), and the solve reads LDS.
template <int VEC, int T = 256>
__device__ __forceinline__ void wg_getrs(const double* __restrict__ LU, int n, double* bs, double* dg = nullptr,
                                         const unsigned char* __restrict__ zmap = nullptr) {
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    constexpr bool STAGE = T > 256;
    constexpr int UNR = 8;  // columns in flight per thread (16 changes nothing: a workgroup's stream is bound by its CU, ~30 GB/s)
    constexpr int SPT = STAGE ? (64 * 64) / T : 1;  // staged entries per thread
    double stg[SPT];
    auto stage_load = [&](int kb) {  // diagonal block at (kb, kb): entry e = t + j T is row e & 63 of column e >> 6
        const int kw = (n - kb) < 64 ? (n - kb) : 64;
#pragma unroll
        for (int j = 0; j < SPT; ++j) {
            const int e = t + j * T, c = e >> 6, r = e & 63;
            stg[j] = (c < kw && r < kw) ? LU[(long)(kb + c) * n + kb + r] : 0.0;
        }
    };
    auto stage_store = [&]() {
#pragma unroll
        for (int j = 0; j < SPT; ++j) dg[t + j * T] = stg[j];
    };
    if (STAGE) {
        stage_load(0);
        stage_store();
        __syncthreads();
    }
    // Diagonal blocks from global memory (T = 256): buffer loads whose per-lane offset is out of range for the lanes that take no
    // part -- they read +0.0 -- instead of a guarded load per column. Eight guarded loads are eight branches; the compiler's
    // wait-count bookkeeping then gives up and every step of the solve waits for ALL outstanding loads, the next group's
    // prefetch included: one trip to memory per eight steps, 5 us per diagonal block, 32 blocks per solve at n = 512.
    // (32-bit byte counts and scalar offsets: exact up to the largest matrix this library factors, LU_BIG_MAX_N = 4096 -> 128 MiB)
    static_assert((long long)LU_BIG_MAX_N * LU_BIG_MAX_N * 8 < (1ll << 31), "the buffer descriptor's 32-bit sizes and offsets assume n * n * 8 < 2^31");
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(LU), 0, STAGE ? 0 : n * n * 8, 0x00020000);
    constexpr unsigned OOB = 0xfffffff0u;
    __shared__ int s_kfin;  // the entries of b the running column block multiplies with are all finite
    auto neg_zero = [](double x) { return __double_as_longlong(x) == (long long)0x8000000000000000ull; };

    // ---- forward: L y = b, unit diagonal (dense.rs:188-194)
#pragma unroll 1
    for (int kb = 0; kb < n; kb += 64) {
        const int kw = (n - kb) < 64 ? (n - kb) : 64;
        if (!tb::NODIAG && wave == 0) {  // (tb::NODIAG / tb::NOSWEEP: timing builds of exp_switches.hpp, false in the product)
            const int i = kb + lane;
            double bi = (lane < kw) ? bs[i] : 0.0;
            // The columns of the diagonal block come in groups of UNR; the NEXT group is requested before the steps of the current
            // one run (two register sets). Requested after them, every group was a round trip to memory in the middle of a
            // 64-step dependent chain: ~10 us per diagonal block, 16 blocks per solve at n = 512 -- the lifetime of a workgroup, and
            // of every launch that serves few systems (the later Newton iterations of a round).
            auto ldl = [&](const int k0, double (&l)[UNR]) {
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int k = k0 + u;
                    if (STAGE) l[u] = (k + 1 < kw && lane > k && lane < kw) ? dg[k * 64 + lane] : 0.0;
                    else l[u] = buf_load_f64(rsrc, (k + 1 < kw && lane > k && lane < kw) ? (unsigned)i * 8u : OOB, __builtin_amdgcn_readfirstlane((kb + (k < kw ? k : kw - 1)) * n * 8));  // (groups past the block's end: every lane out of range anyway; the scalar offset stays inside the matrix)
                }
            };
            auto fwd = [&](const int k0, const double (&l)[UNR]) {
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int k = k0 + u;
                    if (k + 1 < kw) {
                        const double bk = readlane_f64(bi, k);
                        if (lane > k && lane < kw) bi -= l[u] * bk;
                    }
                }
            };
            double la[UNR], lb[UNR];
            ldl(0, la);
#pragma unroll 1
            for (int k0 = 0; k0 + 1 < kw; k0 += 2 * UNR) {
                ldl(k0 + UNR, lb);  // (past the block's end: every entry is 0.0, no load)
                fwd(k0, la);
                ldl(k0 + 2 * UNR, la);
                fwd(k0 + UNR, lb);
            }
            if (lane < kw) bs[i] = bi;
            if (STAGE) {
                const unsigned long long bad = __ballot(lane < kw && !(fabs(bi) <= 1.7976931348623157e308));
                if (lane == 0) s_kfin = bad == 0ull;
            }
        }
        __syncthreads();
        const unsigned char* zk = (STAGE && zmap) ? zmap + (kb >> 6) * 64 : nullptr;
        const bool kfin = zk != nullptr && s_kfin != 0;
        const int ibeg = kb + 64;
        if (STAGE && ibeg < n) stage_load(ibeg);  // the next diagonal block, in flight behind the sweep
        if (!tb::NOSWEEP && ibeg < n) {  // rows below a full 64-column block
#pragma unroll 1
            for (int i = ibeg + VEC * t; i < n; i += VEC * T) {
                double acc[VEC];
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc[v] = (i + v < n) ? bs[i + v] : 0.0;
                if (kfin && zk[i >> 6] == 0) {  // a zero block of L: nothing to subtract from these rows
                    bool nz = false;
#pragma unroll
                    for (int v = 0; v < VEC; ++v) nz = nz || neg_zero(acc[v]);
                    if (!nz) continue;
                }
#pragma unroll 1
                for (int k = 0; k < 64; k += UNR) {
                    double lv[UNR][VEC];
#pragma unroll
                    for (int u = 0; u < UNR; ++u) {
                        const double* p = LU + (long)(kb + k + u) * n + i;
                        if constexpr (VEC == 2) {
                            const double2 q = *reinterpret_cast<const double2*>(p);
                            lv[u][0] = q.x;
                            lv[u][1] = q.y;
                        } else {
                            lv[u][0] = *p;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < UNR; ++u) {
                        const double bk = bs[kb + k + u];
#pragma unroll
                        for (int v = 0; v < VEC; ++v) acc[v] -= lv[u][v] * bk;
                    }
                }
#pragma unroll
                for (int v = 0; v < VEC; ++v)
                    if (i + v < n) bs[i + v] = acc[v];
            }
        }
        if (STAGE && ibeg < n) stage_store();  // (the solve that read dg ended before the barrier above)
        __syncthreads();
    }

    // ---- backward: U x = y (dense.rs:197-205): b_k /= u_kk, then b_i -= u_ik b_k for i < k, k descending
    const int nblk = (n + 63) / 64;
#pragma unroll 1
    for (int blk = nblk - 1; blk >= 0; --blk) {
        const int kb = blk * 64;
        const int kw = (n - kb) < 64 ? (n - kb) : 64;
        if (!tb::NODIAG && wave == 0) {
            const int i = kb + lane;
            double bi = (lane < kw) ? bs[i] : 0.0;
            auto ldu = [&](const int k0, double (&uu)[UNR]) {  // (the next group in flight behind the current one's steps, as above)
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int k = k0 - u;
                    if (STAGE) uu[u] = (k >= 0 && lane <= k) ? dg[k * 64 + lane] : 1.0;
                    else uu[u] = buf_load_f64(rsrc, (k >= 0 && lane <= k) ? (unsigned)i * 8u : OOB, __builtin_amdgcn_readfirstlane((kb + (k >= 0 ? k : 0)) * n * 8));  // (0.0 where 1.0 stood: never used)
                }
            };
            auto bwd = [&](const int k0, const double (&uu)[UNR]) {
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int k = k0 - u;
                    if (k >= 0) {
                        if (lane == k) bi = bi / uu[u];
                        const double bk = readlane_f64(bi, k);
                        if (lane < k) bi -= uu[u] * bk;
                    }
                }
            };
            double ua[UNR], ub[UNR];
            ldu(kw - 1, ua);
#pragma unroll 1
            for (int k0 = kw - 1; k0 >= 0; k0 -= 2 * UNR) {
                ldu(k0 - UNR, ub);  // (below column 0: every entry is 1.0, no load)
                bwd(k0, ua);
                ldu(k0 - 2 * UNR, ua);
                bwd(k0 - UNR, ub);
            }
            if (lane < kw) bs[i] = bi;
            if (STAGE) {
                const unsigned long long bad = __ballot(lane < kw && !(fabs(bi) <= 1.7976931348623157e308));
                if (lane == 0) s_kfin = bad == 0ull;
            }
        }
        __syncthreads();
        const unsigned char* zk = (STAGE && zmap) ? zmap + (kb >> 6) * 64 : nullptr;
        const bool kfin = zk != nullptr && s_kfin != 0;
        if (STAGE && kb > 0) stage_load(kb - 64);
        if (!tb::NOSWEEP && kb > 0) {
#pragma unroll 1
            for (int i = VEC * t; i < kb; i += VEC * T) {  // kb is a multiple of 64 (hence even): i + v < kb
                double acc[VEC];
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc[v] = bs[i + v];
                if (kfin && zk[i >> 6] == 0) {  // a zero block of U
                    bool nz = false;
#pragma unroll
                    for (int v = 0; v < VEC; ++v) nz = nz || neg_zero(acc[v]);
                    if (!nz) continue;
                }
#pragma unroll 1
                for (int k0 = kw - 1; k0 >= 0; k0 -= UNR) {
                    double uv[UNR][VEC];
#pragma unroll
                    for (int u = 0; u < UNR; ++u) {
                        const int k = (k0 - u) >= 0 ? (k0 - u) : 0;
                        const double* p = LU + (long)(kb + k) * n + i;
                        if constexpr (VEC == 2) {
                            const double2 q = *reinterpret_cast<const double2*>(p);
                            uv[u][0] = q.x;
                            uv[u][1] = q.y;
                        } else {
                            uv[u][0] = *p;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < UNR; ++u) {
                        const int k = k0 - u;
                        if (k >= 0) {
                            const double bk = bs[kb + k];
#pragma unroll
                            for (int v = 0; v < VEC; ++v) acc[v] -= uv[u][v] * bk;
                        }
                    }
                }
#pragma unroll
                for (int v = 0; v < VEC; ++v) bs[i + v] = acc[v];
            }
        }
        if (STAGE && kb > 0) stage_store();
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ stand-alone solve
// x <- b (dense.rs:59), permute by pivots, forward/backward substitution. One workgroup per system.
template <int VEC>
__global__ __launch_bounds__(256, 4) void ls_solve_kernel(const double* __restrict__ LU, long mstride, const long long* __restrict__ piv,
                                                       long pstride, const int* __restrict__ perm /* or null */,
                                                       double* __restrict__ X, const double* __restrict__ Bv, int n,
                                                       const int* __restrict__ idx) {
    extern __shared__ __align__(16) double sm[];
    double* bs = sm;
    int* sperm = reinterpret_cast<int*>(sm + n);
    const int b = idx[blockIdx.x];
    const int t = threadIdx.x;
    if (perm) {
        for (int i = t; i < n; i += 256) bs[i] = Bv[(long)b * n + perm[(long)b * n + i]];
    } else {
        // compose the reference's sequential swaps (dense.rs:181-185) into a permutation, one lane, in LDS
        for (int i = t; i < n; i += 256) sperm[i] = i;
        __syncthreads();
        if (t == 0) {
            const long long* pv = piv + (long)b * pstride;
            for (int k = 0; k < n; ++k) {
                const int pk = (int)pv[k];
                if (pk != k) {
                    const int tmp = sperm[k];
                    sperm[k] = sperm[pk];
                    sperm[pk] = tmp;
                }
            }
        }
        __syncthreads();
        for (int i = t; i < n; i += 256) bs[i] = Bv[(long)b * n + sperm[i]];
    }
    __syncthreads();
    wg_getrs<VEC>(LU + (long)b * mstride, n, bs);
    for (int i = t; i < n; i += 256) X[(long)b * n + i] = bs[i];
}

// ------------------------------------------------------------------------------------------------ WRMS
__global__ __launch_bounds__(256) void wrms_kernel(const double* __restrict__ X, const double* __restrict__ W, double* __restrict__ out,
                                                   int n, const int* __restrict__ idx) {
    extern __shared__ __align__(16) double sm[];
    const int b = idx[blockIdx.x];
    for (int i = threadIdx.x; i < n; i += 256) {
        const double p = X[(long)b * n + i] * W[(long)b * n + i];
        sm[i] = p * p;
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = seq_sum_lds(sm, n);
}

// ------------------------------------------------------------------------------------------------ fused Newton body
// delta = -delta; delta = LU^-1 delta; delta *= scale; ee += delta; out = sum_i (delta_i ewt_i)^2 (sequential).
// T = 1024 for n >= 2048: one workgroup per system streams 8 n^2 bytes, and with few large systems in a call (config 4: ~128
// of 134 MB each) 256 threads per system do not keep enough loads in flight to fill the memory system.
template <int VEC, int T = 256>
__global__ __launch_bounds__(T, T == 256 ? 4 : 1) void newton_iter_kernel(const double* __restrict__ LU, const int* __restrict__ perm, double* __restrict__ delta,
                                                          double* __restrict__ ee, const double* __restrict__ ewt, int n,
                                                          const int* __restrict__ idx, const double* __restrict__ scale,
                                                          double* __restrict__ out, const int* __restrict__ skip,
                                                          const unsigned char* __restrict__ zmap /* or null: lu_kernels.hpp, LuWs::zmap */) {
    extern __shared__ __align__(16) double sm[];
    double* bs = sm;
    double* sq = sm + n;
    if (skip && skip[blockIdx.x] != 0) return;  // (idahip_newton_iter2: this system's Newton solve has ended)
    const int b = idx[blockIdx.x];
    const int t = threadIdx.x;
    const long vb = (long)b * n;
    for (int i = t; i < n; i += T) bs[i] = -delta[vb + perm[vb + i]];  // neg_mut, then the row permutation
    __syncthreads();
    wg_getrs<VEC, T>(LU + (long)b * n * n, n, bs, T > 256 ? sq : nullptr, zmap ? zmap + (long)b * 4096 : nullptr);  // sq (>= 4096 doubles for T > 256) is free until the solve is done
    const double sc = scale[blockIdx.x];
    for (int i = t; i < n; i += T) {
        const double d = bs[i] * sc;  // ida_ls.rs:406-410 (sc == 1.0 exactly when cjratio == 1)
        delta[vb + i] = d;
        ee[vb + i] = ee[vb + i] + d;  // newton.rs:106
        const double p = d * ewt[vb + i];
        sq[i] = p * p;
    }
    __syncthreads();
    if (t == 0) out[blockIdx.x] = seq_sum_lds(sq, n);
}

// ------------------------------------------------------------------------------------------------ tiny (n <= 8)
__device__ inline void tiny_getrs(const double* __restrict__ a, int n, const long long* __restrict__ pivot, double* b) {
    for (int k = 0; k < n; ++k) {
        const int pk = (int)pivot[k];
        if (pk != k) {
            const double tmp = b[k];
            b[k] = b[pk];
            b[pk] = tmp;
        }
    }
    for (int k = 0; k + 1 < n; ++k) {
        const double bk = b[k];
        for (int i = k + 1; i < n; ++i) b[i] -= a[k * n + i] * bk;
    }
    for (int k = n - 1; k >= 1; --k) {
        b[k] /= a[k * n + k];
        const double bk = b[k];
        for (int i = 0; i < k; ++i) b[i] -= a[k * n + i] * bk;
    }
    b[0] /= a[0];
}

__global__ void tiny_solve_kernel(const double* LU, long mstride, const long long* piv, long pstride, double* X, const double* Bv,
                                  int n, const int* idx, int nsys) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsys) return;
    const int b = idx[s];
    double v[TINY_N];
    for (int i = 0; i < n; ++i) v[i] = Bv[(long)b * n + i];
    tiny_getrs(LU + (long)b * mstride, n, piv + (long)b * pstride, v);
    for (int i = 0; i < n; ++i) X[(long)b * n + i] = v[i];
}

__global__ void tiny_wrms_kernel(const double* X, const double* W, double* out, int n, const int* idx, int nsys) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsys) return;
    const int b = idx[s];
    double acc = 0.0;
    for (int i = 0; i < n; ++i) {
        const double p = X[(long)b * n + i] * W[(long)b * n + i];
        acc = acc + p * p;
    }
    out[s] = acc;
}

__global__ void tiny_newton_iter_kernel(const double* LU, const long long* piv, double* delta, double* ee, const double* ewt, int n,
                                        const int* idx, int nsys, const double* scale, double* out, const int* skip) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsys) return;
    if (skip && skip[s] != 0) return;
    const int b = idx[s];
    const long vb = (long)b * n;
    double v[TINY_N];
    for (int i = 0; i < n; ++i) v[i] = -delta[vb + i];
    tiny_getrs(LU + (long)b * n * n, n, piv + vb, v);
    const double sc = scale[s];
    double acc = 0.0;
    for (int i = 0; i < n; ++i) {
        const double d = v[i] * sc;
        delta[vb + i] = d;
        ee[vb + i] = ee[vb + i] + d;
        const double p = d * ewt[vb + i];
        acc = acc + p * p;
    }
    out[s] = acc;
}

// ------------------------------------------------------------------------------------------------ idaNlsConvTest on the device
// The first two convergence tests of a Newton solve (src/ida_nls.rs:243-262) need no pow: m = 0 is two comparisons, m = 1 has
// rate = (delnrm / oldnrm)^(1/1) = delnrm / oldnrm exactly. sum[q] is the kernel's sequential sum of (delta_i ewt_i)^2;
// delnrm = sqrt(sum / n) -- a correctly rounded division and square root, as on the host (norm_rms.rs:36-37).
// conv[q]: 0 = keep iterating, 1 = converged at m = 0, 2 = converged at m = 1, 3 = ConvergenceRecover at m = 1 (rate > 0.9).
__global__ void ctest_kernel(const double* __restrict__ sum, int n, int m, const double* __restrict__ toldel, const double* __restrict__ ss,
                             const double* __restrict__ eps_newt, double* __restrict__ delnrm /*[nsys][2]*/, int* __restrict__ conv,
                             int nsys) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nsys) return;
    if (m == 0) {
        const double d0 = sqrt(sum[q] / (double)n);
        delnrm[2 * q] = d0;
        delnrm[2 * q + 1] = 0.0;
        int c = 0;
        if (d0 <= 0.0001 * toldel[q]) c = 1;       // ida_nls.rs:245
        else if (ss[q] * d0 <= eps_newt[q]) c = 1;  // ida_nls.rs:260 with the ss of the previous solve / setup
        conv[q] = c;
    } else {
        if (conv[q] != 0) return;
        const double d0 = delnrm[2 * q];
        const double d1 = sqrt(sum[q] / (double)n);
        delnrm[2 * q + 1] = d1;
        const double rate = d1 / d0;               // powf(base, 1/m) with m = 1 (ida_nls.rs:249-253)
        int c = 0;
        if (rate > 0.9) {                          // RATEMAX (ida_nls.rs:15,254)
            c = 3;
        } else {
            const double ss1 = rate / (1.0 - rate);
            if (ss1 * d1 <= eps_newt[q]) c = 2;
        }
        conv[q] = c;
    }
}

}  // namespace idahip
