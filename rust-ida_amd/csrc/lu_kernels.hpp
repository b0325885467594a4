// Batched dense LU with partial pivoting for gfx950: replaces dense_get_rf
// (/root/reference/crates/linear/src/dense.rs:86-158) for a list of independent matrices.
//
// Result contract (bit-exact with the reference): same pivots, same LU bits. That holds because
//   * every element a(i,j) receives its updates a -= a_kj * a_ik in ascending k, each an unfused mul then sub
//     (the file is compiled with -ffp-contract=off), skipped when a_kj == 0 (dense.rs:148);
//   * multipliers are a_ik * (1/a_kk) (dense.rs:134-137); the pivot is the first row *in the reference's current
//     row order* attaining max |a_ik| (strict '>', dense.rs:113).
//
// Design (MI355X-first, not a port of the reference's triple loop):
//   * implicit pivoting -- rows never move during the factorisation. Each physical row r carries `pos[r]`, the
//     position it occupies in the reference's (explicitly swapped) matrix; the argmax key is (|a|, pos) so ties
//     resolve exactly as the reference's scan does. Column-major storage makes physical row swaps a strided,
//     uncoalesced disaster on a GPU; here every access is a coalesced column segment.
//   * right-looking blocked algorithm, panel width NB, three kernels per panel step over the whole list of
//     matrices (lock-step batch, hundreds to thousands of workgroups per launch):
//       lu_panel  : one workgroup per matrix, one live row per thread, the row's NB panel entries in registers;
//                   wave-shuffle + LDS arg-max, pivot row broadcast through LDS.
//       lu_trsm   : U12 = L11^-1 A12, one lane per trailing column, L11 through the scalar cache.
//       lu_update : A22 -= L21 U12, one live row per lane, the row's NB multipliers in registers, 16-column register
//                   tiles, U12 streamed through SGPRs (scalar loads) -- no LDS traffic, no barriers.
//   * a final pass scatters rows to their pivoted positions (the reference layout the solve kernels stream).
// Blocking changes neither the per-element operation order nor any operand, only when each update is applied.
#pragma once
#include <cstdlib>

#include "common.hpp"
#include "lu_left.hpp"

namespace idahip {

struct LuWs {
    double* mats;      // work matrices (physical row order), column-major n x n
    long mstride;      // elements between consecutive systems
    const int* idx;    // [nsys] system ids (device)
    int n;
    int npad16;
    int* pos;          // [batch][n] physical row -> current reference position
    int* live;         // [batch][n] sorted physical indices of not-yet-pivoted rows
    int* prow;         // [batch][n] pivot step -> physical row
    long long* piv;    // [batch][n] reference pivots (position chosen at step k)
    long pstride;      // elements between consecutive systems in piv
    int* info;         // [batch]   0 | 1-based zero-pivot column
    double* l11;       // [batch][NB*NB] row k = multipliers of the k-th pivot row
    double* ubuf;      // [batch][NB][npad16] U12 rows, contiguous along columns
    int* uz;           // [batch][npad16/16] 1 if the 16-column chunk of U12 holds an exact zero (or padding)
};

__global__ void lu_init_kernel(LuWs w) {
    const int b = w.idx[blockIdx.x];
    for (int i = threadIdx.x; i < w.n; i += blockDim.x) {
        w.pos[(long)b * w.n + i] = i;
        w.live[(long)b * w.n + i] = i;
    }
    if (threadIdx.x == 0) w.info[b] = 0;
}

// ------------------------------------------------------------------------------------------------ panel
template <int NB, int MAXT>
__global__ __launch_bounds__(MAXT) void lu_panel_kernel(LuWs w, int k0) {
    const int b = w.idx[blockIdx.x];
    if (w.info[b] != 0) return;
    const int n = w.n;
    double* __restrict__ A = w.mats + (long)b * w.mstride;
    int* __restrict__ pos = w.pos + (long)b * n;
    int* __restrict__ live = w.live + (long)b * n;
    int* __restrict__ prow = w.prow + (long)b * n;
    long long* __restrict__ piv = w.piv + (long)b * w.pstride;
    double* __restrict__ l11 = w.l11 + (long)b * NB * NB;

    const int m = n - k0;
    const int wd = m < NB ? m : NB;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, nwaves = blockDim.x >> 6;

    __shared__ double s_v[2][16];
    __shared__ int s_p[2][16];
    __shared__ double s_prow[2][NB + 1];  // [..][NB] = 1/pivot
    __shared__ int s_cnt[16];
    __shared__ unsigned s_zm[2];

    const bool valid = t < m;
    const int r = valid ? live[t] : 0;
    int mypos = valid ? pos[r] : 0x7fffffff;
    double a[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) a[j] = (valid && j < wd) ? A[(long)(k0 + j) * n + r] : 0.0;

    bool alive = valid;
    int ownk = -1;
    bool failed = false;

#pragma unroll
    for (int k = 0; k < NB; ++k) {
        if (k < wd && !failed) {
            const int kc = k0 + k;
            // candidate key: (|a|, position); NaN only wins if it sits at position kc (dense.rs:111-117 scan semantics)
            double v = -1.0;
            int p = 0x7fffffff;
            if (alive) {
                v = fabs(a[k]);
                p = mypos;
                if (v != v) v = (mypos == kc) ? __builtin_huge_val() : -1.0;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const double ov = shfl_xor_f64(v, off);
                const int op = __shfl_xor(p, off);
                if (ov > v || (ov == v && op < p)) {
                    v = ov;
                    p = op;
                }
            }
            if (lane == 0) {
                s_v[k & 1][wave] = v;
                s_p[k & 1][wave] = p;
            }
            __syncthreads();
            double bv = s_v[k & 1][0];
            int bp = s_p[k & 1][0];
            for (int q = 1; q < nwaves; ++q) {
                const double ov = s_v[k & 1][q];
                const int op = s_p[k & 1][q];
                if (ov > bv || (ov == bv && op < bp)) {
                    bv = ov;
                    bp = op;
                }
            }
            const bool owner = alive && (mypos == bp);
            if (owner) {
                unsigned zm = 0u;
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    if (j >= k) {
                        s_prow[k & 1][j] = a[j];
                        if (j > k && a[j] == 0.0) zm |= 1u << j;
                    }
                s_prow[k & 1][NB] = 1.0 / a[k];  // mult = a(k,k).recip()  (dense.rs:134)
                s_zm[k & 1] = zm;                // columns whose update is skipped (a_kj == 0, dense.rs:148)
            }
            __syncthreads();
            const double pk = s_prow[k & 1][k];
            if (pk == 0.0) {  // zero pivot: Err(k+1)  (dense.rs:120-122)
                if (t == 0) w.info[b] = kc + 1;
                failed = true;
            } else {
                if (t == 0) piv[kc] = (long long)bp;
                if (owner) {
                    prow[kc] = r;
                    ownk = k;
                    alive = false;
                    mypos = kc;
                } else if (alive) {
                    if (mypos == kc) mypos = bp;  // the row that sat at position k moves to the pivot's old position
                    const double mult = s_prow[k & 1][NB];
                    a[k] *= mult;
                    const double aik = a[k];
                    const unsigned zm = (unsigned)__builtin_amdgcn_readfirstlane((int)s_zm[k & 1]);
                    if (zm == 0u) {
#pragma unroll
                        for (int j = 0; j < NB; ++j)
                            if (j > k && j < wd) a[j] -= s_prow[k & 1][j] * aik;  // dense.rs:151
                    } else {
#pragma unroll
                        for (int j = 0; j < NB; ++j)
                            if (j > k && j < wd && !((zm >> j) & 1u)) a[j] -= s_prow[k & 1][j] * aik;
                    }
                }
            }
        }
    }
    if (failed) return;

    // write back the panel, positions, L11 and the compacted live list
    if (valid) {
#pragma unroll
        for (int j = 0; j < NB; ++j)
            if (j < wd) A[(long)(k0 + j) * n + r] = a[j];
        pos[r] = mypos;
        if (ownk >= 0) {
#pragma unroll
            for (int j = 0; j < NB; ++j)
                if (j < ownk) l11[ownk * NB + j] = a[j];
        }
    }
    const unsigned long long bal = __ballot(alive);
    if (lane == 0) s_cnt[wave] = __popcll(bal);
    __syncthreads();
    int base = 0;
    for (int q = 0; q < wave; ++q) base += s_cnt[q];
    if (alive) live[base + __popcll(bal & ((1ull << lane) - 1ull))] = r;
}

// ------------------------------------------------------------------------------------------------ trsm (U12)
template <int NB>
__global__ __launch_bounds__(64) void lu_trsm_kernel(LuWs w, int k0) {
    const int b = w.idx[blockIdx.x];
    if (w.info[b] != 0) return;
    const int n = w.n;
    double* __restrict__ A = w.mats + (long)b * w.mstride;
    const int* __restrict__ prow = w.prow + (long)b * n + k0;
    const double* __restrict__ l11 = w.l11 + (long)b * NB * NB;
    double* __restrict__ U = w.ubuf + (long)b * NB * w.npad16;
    int* __restrict__ uz = w.uz + (long)b * (w.npad16 / 16);

    const int lane = threadIdx.x;
    const int c0 = k0 + NB + blockIdx.y * 64;  // only launched for full panels with a trailing matrix
    const int jc = c0 + lane;
    const bool valid = jc < n;
    const bool inpad = jc < w.npad16;

    double u[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) u[k] = valid ? A[(long)jc * n + ldc(prow + k)] : 0.0;

    // Right-looking order over the source row kk: u[kk] is final once rows 0..kk-1 have been applied, and every target
    // u[k], k > kk, still receives its updates in ascending kk. The a_kj == 0 skip (dense.rs:148) is per column
    // (= per lane) here; one ballot per source row selects the unpredicated body when no lane holds a zero.
    bool anyzero = !valid;
#pragma unroll
    for (int kk = 0; kk < NB; ++kk) {
        const double ukk = u[kk];
        const bool z = (ukk == 0.0);
        anyzero = anyzero || z;
        if (valid) A[(long)jc * n + ldc(prow + kk)] = ukk;
        if (inpad) U[(long)kk * w.npad16 + jc] = ukk;
        if (__ballot(z) == 0ull) {
#pragma unroll
            for (int k = 0; k < NB; ++k)
                if (k > kk) u[k] -= ukk * ldc(l11 + k * NB + kk);  // a(i,j) -= a_kj * a_ik
        } else {
#pragma unroll
            for (int k = 0; k < NB; ++k)
                if (k > kk) {
                    const double t = u[k] - ukk * ldc(l11 + k * NB + kk);
                    u[k] = z ? u[k] : t;
                }
        }
    }
    const unsigned long long bal = __ballot(anyzero);
    if ((lane & 15) == 0 && inpad) uz[jc >> 4] = ((bal >> (lane & 48)) & 0xffffull) != 0ull;
}

// ------------------------------------------------------------------------------------------------ trailing update
template <int NB>
__global__ __launch_bounds__(256) void lu_update_kernel(LuWs w, int k0, int csplit) {
    const int b = w.idx[blockIdx.x];
    if (w.info[b] != 0) return;
    const int n = w.n;
    double* __restrict__ A = w.mats + (long)b * w.mstride;
    const int* __restrict__ live = w.live + (long)b * n;
    const double* __restrict__ Ub = w.ubuf + (long)b * NB * w.npad16;
    const int* __restrict__ uz = w.uz + (long)b * (w.npad16 / 16);

    const int mrem = n - k0 - NB;  // live rows after this panel
    const int rt = blockIdx.y * 256 + threadIdx.x;
    const int wave_first = blockIdx.y * 256 + (threadIdx.x & ~63);
    if (wave_first >= mrem) return;  // whole wave idle
    const bool valid = rt < mrem;
    const int r = live[valid ? rt : (mrem - 1)];

    double l[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) l[k] = A[(long)(k0 + k) * n + r];

    const int cbeg = k0 + NB;
    const int nchunks = (n - cbeg + 15) >> 4;
    for (int ch = blockIdx.z; ch < nchunks; ch += csplit) {
        const int c0 = cbeg + (ch << 4);
        double c[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) c[j] = (c0 + j < n) ? A[(long)(c0 + j) * n + r] : 0.0;
        const double* __restrict__ U = Ub + c0;
        if (ldc(uz + (c0 >> 4)) == 0) {
#pragma unroll
            for (int k = 0; k < NB; ++k) {
#pragma unroll
                for (int j = 0; j < 16; ++j) c[j] -= ldc(U + (long)k * w.npad16 + j) * l[k];
            }
        } else {
#pragma unroll
            for (int k = 0; k < NB; ++k) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const double ukj = ldc(U + (long)k * w.npad16 + j);
                    if (ukj != 0.0) c[j] -= ukj * l[k];
                }
            }
        }
        if (valid) {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (c0 + j < n) A[(long)(c0 + j) * n + r] = c[j];
        }
    }
}

// ------------------------------------------------------------------------------------------------ finalize
// out(pos[r], j) = work(r, j); perm[pos[r]] = r. One workgroup per (matrix, column group).
__global__ __launch_bounds__(256) void lu_finalize_kernel(LuWs w, double* __restrict__ out, long ostride, int* __restrict__ perm,
                                                          int cols_per_block) {
    const int b = w.idx[blockIdx.x];
    if (w.info[b] != 0) return;
    const int n = w.n;
    const double* __restrict__ A = w.mats + (long)b * w.mstride;
    double* __restrict__ O = out + (long)b * ostride;
    const int* __restrict__ pos = w.pos + (long)b * n;
    const int jbeg = blockIdx.y * cols_per_block;
    const int jend = (jbeg + cols_per_block < n) ? jbeg + cols_per_block : n;
    for (int r = threadIdx.x; r < n; r += blockDim.x) {
        const int p = pos[r];
        if (blockIdx.y == 0 && perm) perm[(long)b * n + p] = r;
        for (int j = jbeg; j < jend; ++j) O[(long)j * n + p] = A[(long)j * n + r];
    }
}

// ------------------------------------------------------------------------------------------------ tiny (n <= 8)
// One thread per system: the reference algorithm verbatim on the thread's own matrix (dense.rs:86-158).
__device__ inline int tiny_getrf(double* __restrict__ a, int n, long long* __restrict__ pivot, int* __restrict__ perm) {
    for (int i = 0; i < n; ++i) perm[i] = i;
    for (int k = 0; k < n; ++k) {
        double* col_k = a + k * n;
        int l = k;
        for (int i = k + 1; i < n; ++i)
            if (fabs(col_k[i]) > fabs(col_k[l])) l = i;
        pivot[k] = l;
        if (col_k[l] == 0.0) return k + 1;
        if (l != k) {
            for (int i = 0; i < n; ++i) {
                const double tmp = a[i * n + k];
                a[i * n + k] = a[i * n + l];
                a[i * n + l] = tmp;
            }
            const int tp = perm[k];
            perm[k] = perm[l];
            perm[l] = tp;
        }
        const double mult = 1.0 / a[k * n + k];
        for (int i = k + 1; i < n; ++i) a[k * n + i] *= mult;
        for (int j = k + 1; j < n; ++j) {
            const double a_kj = a[j * n + k];
            if (a_kj != 0.0) {
                for (int i = k + 1; i < n; ++i) a[j * n + i] -= a_kj * a[k * n + i];
            }
        }
    }
    return 0;
}

__global__ void tiny_getrf_kernel(double* mats, long mstride, const int* idx, int nsys, int n, long long* piv, long pstride,
                                  int* perm, int* info) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsys) return;
    const int b = idx[s];
    int lperm[TINY_N];
    info[b] = tiny_getrf(mats + (long)b * mstride, n, piv + (long)b * pstride, lperm);
    if (perm)
        for (int i = 0; i < n; ++i) perm[(long)b * n + i] = lperm[i];
}

// ------------------------------------------------------------------------------------------------ host driver
// Factor the matrices `work[b]` (physical order, destroyed) of the listed systems into out[b] (reference layout).
inline int lu_factor_batched(idahip_ctx* c, double* work, long wstride, double* out, long ostride, long long* piv, long pstride,
                             int* perm, const int* d_idx, int nsys) {
    const int n = c->n;
    if (nsys == 0) return 0;
    if (n <= TINY_N) {
        // tiny path factors in place in `work`, then `out` is a plain copy if distinct
        hipLaunchKernelGGL(tiny_getrf_kernel, dim3((nsys + 63) / 64), dim3(64), 0, c->stream, work, wstride, d_idx, nsys, n, piv,
                           pstride, perm, c->lu_info);
        return 0;
    }
    if (n > LU_MAX_N) return fail(c, -3, "blocked LU supports n <= %d in this build (n = %d)", LU_MAX_N, n);
    static const bool use_right = getenv("IDAHIP_LU_RIGHT") != nullptr;  // A/B switch: multi-kernel right-looking pipeline
    if (!use_right) return lu_left_launch(c, work, wstride, out, ostride, piv, pstride, perm, d_idx, nsys);
    constexpr int NB = LU_NB;
    LuWs w;
    w.mats = work; w.mstride = wstride; w.idx = d_idx; w.n = n; w.npad16 = c->npad16;
    w.pos = c->lu_pos; w.live = c->lu_live; w.prow = c->lu_prow; w.piv = piv; w.pstride = pstride; w.info = c->lu_info;
    w.l11 = c->lu_l11; w.ubuf = c->lu_ubuf; w.uz = c->lu_uz;
    hipLaunchKernelGGL(lu_init_kernel, dim3(nsys), dim3(256), 0, c->stream, w);
    for (int k0 = 0; k0 < n; k0 += NB) {
        const int m = n - k0;
        const int threads = ((m + 63) / 64) * 64;
        if (threads <= 512)
            hipLaunchKernelGGL((lu_panel_kernel<NB, 512>), dim3(nsys), dim3(threads), 0, c->stream, w, k0);
        else
            hipLaunchKernelGGL((lu_panel_kernel<NB, 1024>), dim3(nsys), dim3(threads), 0, c->stream, w, k0);
        const int ntrail = n - k0 - NB;
        if (ntrail > 0) {
            hipLaunchKernelGGL(lu_trsm_kernel<NB>, dim3(nsys, (c->npad16 - (k0 + NB) + 63) / 64), dim3(64), 0, c->stream, w, k0);
            const int rowgroups = (ntrail + 255) / 256;
            const int nchunks = (ntrail + 15) / 16;
            int csplit = 1;
            // aim for a few thousand workgroups per launch so 256 CUs stay busy across the tail
            while ((long)nsys * rowgroups * csplit < 4096 && csplit < nchunks) csplit *= 2;
            if (csplit > nchunks) csplit = nchunks;
            hipLaunchKernelGGL(lu_update_kernel<NB>, dim3(nsys, rowgroups, csplit), dim3(256), 0, c->stream, w, k0, csplit);
        }
    }
    const int cpb = 32;
    hipLaunchKernelGGL(lu_finalize_kernel, dim3(nsys, (n + cpb - 1) / cpb), dim3(256), 0, c->stream, w, out, ostride, perm, cpb);
    return 0;
}

}  // namespace idahip
