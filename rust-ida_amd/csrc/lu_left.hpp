// Left-looking, register-resident batched LU for gfx950 -- one workgroup factors one matrix in ONE launch.
// Replaces dense_get_rf (/root/reference/crates/linear/src/dense.rs:86-158); same bit-exactness contract as
// lu_kernels.hpp (ascending-k unfused updates, reciprocal multipliers, a_kj == 0 skip, reference tie-breaking).
//
// Why left-looking on MI355X: the right-looking pipeline (lu_kernels.hpp) re-reads and re-writes the whole trailing
// matrix once per panel (22 MB per 512x512 matrix at NB = 32) and gathers the scattered pivot rows for U12 with
// 8-16x line amplification. Here every thread owns R physical rows for the whole factorisation; a block column
// (NB columns) lives in registers (R x NB doubles per thread) while ALL earlier panels are applied to it:
//     for q < J:   pivot rows of panel q deposit their entries in LDS -> triangular solve in LDS (U12 block)
//                  -> every still-active row subtracts L(row, panel q) * U12, the multipliers streamed from
//                     L2/HBM in coalesced column segments, U12 broadcast from LDS
//     then the block column is factored in place (wave-shuffle + LDS arg-max, implicit pivoting)
// so each matrix entry is read once and written once as data; only the multipliers are re-read (J times for block
// column J), all accesses coalesced, no workspace, no inter-kernel round trips. A last in-kernel pass scatters the
// rows to their pivoted positions (reference layout) and emits the composed permutation for the solve kernels.
#pragma once
#include "common.hpp"

namespace idahip {

struct LuLeftArgs {
    double* mats;      // work matrices (physical row order), column-major n x n; destroyed
    long mstride;
    double* out;       // factors in reference layout (rows at pivoted positions), column-major
    long ostride;
    const int* idx;    // [nsys] system ids (device)
    int n;
    long long* piv;    // [batch][pstride] reference pivots
    long pstride;
    int* perm;         // [batch][n] or null: perm[pos] = physical row
    int* info;         // [batch] 0 | 1-based zero-pivot column
};

template <int NB, int R, int T>
__global__ __launch_bounds__(T) void lu_left_kernel(LuLeftArgs w) {
    constexpr int NW = T / 64;
    constexpr int BIG = 0x7fffffff;
    const int b = ldc(w.idx + blockIdx.x);
    const int n = w.n;
    double* __restrict__ A = w.mats + (long)b * w.mstride;
    long long* __restrict__ piv = w.piv + (long)b * w.pstride;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);

    __shared__ __align__(16) double s_u[NB][NB];      // final U12 block of (J, q): row k contiguous along j
    __shared__ double s_c[NB][NB + 1];                // deposited pivot rows, solved in place ([k][j], padded)
    __shared__ double s_l[NB][NB + 1];                // L11 of panel q, transposed: s_l[kk][k] = l(pivot row k, kk)
    __shared__ unsigned s_zm[2][NB];                  // per U12 row: bit j set when u(k, j) == 0 (update skipped)
    __shared__ double s_v[2][NW];                     // panel arg-max scratch
    __shared__ int s_p[2][NW];
    __shared__ __align__(16) double s_prow[2][NB + 2];  // pivot row broadcast; [NB] = 1/pivot
    __shared__ unsigned s_pz[2];

    // per-row state (R rows per thread: row ri is physical row t + ri*T)
    int row[R], mypos[R], step[R];
#pragma unroll
    for (int ri = 0; ri < R; ++ri) {
        row[ri] = t + ri * T;
        mypos[ri] = row[ri] < n ? row[ri] : BIG;
        step[ri] = BIG;  // pivot step (global column index) once pivoted
    }

    const int nblk = (n + NB - 1) / NB;
    bool failed = false;

#pragma unroll 1
    for (int J = 0; J < nblk && !failed; ++J) {
        const int c0 = J * NB;
        const int wd = (n - c0) < NB ? (n - c0) : NB;
        double a[R][NB];
#pragma unroll
        for (int ri = 0; ri < R; ++ri)
#pragma unroll
            for (int j = 0; j < NB; ++j) a[ri][j] = (row[ri] < n && j < wd) ? A[(long)(c0 + j) * n + row[ri]] : 0.0;

        // ---------------------------------------------------------------- apply panels q < J to the block column
#pragma unroll 1
        for (int q = 0; q < J; ++q) {
            const int k0q = q * NB;
            if (t < NB) s_zm[q & 1][t] = 0u;
            // (a) pivot rows of panel q deposit their block-column entries and their L11 multipliers
#pragma unroll
            for (int ri = 0; ri < R; ++ri) {
                const int kq = step[ri] - k0q;
                if (kq >= 0 && kq < NB) {
#pragma unroll
                    for (int j = 0; j < NB; ++j) s_c[kq][j] = a[ri][j];
#pragma unroll
                    for (int kk = 0; kk < NB; ++kk)
                        if (kk < kq) s_l[kk][kq] = A[(long)(k0q + kk) * n + row[ri]];
                }
            }
            __syncthreads();
            // (b) triangular solve U12 = L11^-1 C, two columns per wave pass: lanes 0-31 / 32-63 <-> pivot index k
            {
                const int half = lane >> 5, k = lane & 31;
                for (int pr = wave; pr < NB / 2; pr += NW) {
                    const int j = 2 * pr + half;
                    double val = (k < NB) ? s_c[k][j] : 0.0;
#pragma unroll
                    for (int kk = 0; kk + 1 < NB; ++kk) {
                        const double u0 = readlane_f64(val, kk);
                        const double u1 = readlane_f64(val, 32 + kk);
                        const double u = half ? u1 : u0;
                        const double tnew = val - u * s_l[kk][k];   // a(i,j) -= a_kj * a_ik, ascending kk
                        val = (k > kk && u != 0.0) ? tnew : val;    // dense.rs:148 skip
                    }
                    s_u[k][j] = val;
                    if (val == 0.0) atomicOr(&s_zm[q & 1][k], 1u << j);
                }
            }
            __syncthreads();
            // (c) pivot rows take their final values; active rows subtract L(row, panel q) * U12
#pragma unroll
            for (int ri = 0; ri < R; ++ri) {
                const int kq = step[ri] - k0q;
                if (kq >= 0 && kq < NB) {
#pragma unroll
                    for (int j = 0; j < NB; ++j) a[ri][j] = s_u[kq][j];
                }
            }
            // rows pivoted after panel q (or not yet) were live when panel q was eliminated: they take all NB updates
            bool act[R];
            bool anyact = false;
#pragma unroll
            for (int ri = 0; ri < R; ++ri) {
                act[ri] = row[ri] < n && step[ri] >= k0q + NB;
                anyact = anyact || act[ri];
            }
            if (__ballot(anyact) != 0ull) {  // wave-uniform: skip waves whose rows were all pivoted before panel q
                constexpr int KC = 8;
                double l[R][KC], ln[R][KC];
#pragma unroll
                for (int ri = 0; ri < R; ++ri)
#pragma unroll
                    for (int u = 0; u < KC; ++u) l[ri][u] = act[ri] ? A[(long)(k0q + u) * n + row[ri]] : 0.0;
#pragma unroll 1
                for (int kb = 0; kb < NB; kb += KC) {
                    if (kb + KC < NB) {  // prefetch the next chunk of multipliers behind this chunk's arithmetic
#pragma unroll
                        for (int ri = 0; ri < R; ++ri)
#pragma unroll
                            for (int u = 0; u < KC; ++u) ln[ri][u] = act[ri] ? A[(long)(k0q + kb + KC + u) * n + row[ri]] : 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < KC; ++u) {
                        const unsigned zm = (unsigned)__builtin_amdgcn_readfirstlane((int)s_zm[q & 1][kb + u]);
                        if (zm == 0u) {
#pragma unroll
                            for (int jc = 0; jc < NB; jc += 8) {
                                double uu[8];
#pragma unroll
                                for (int j = 0; j < 8; j += 2) {
                                    const double2 v2 = *reinterpret_cast<const double2*>(&s_u[kb + u][jc + j]);
                                    uu[j] = v2.x;
                                    uu[j + 1] = v2.y;
                                }
#pragma unroll
                                for (int ri = 0; ri < R; ++ri) {
                                    if (act[ri]) {
#pragma unroll
                                        for (int j = 0; j < 8; ++j) a[ri][jc + j] -= uu[j] * l[ri][u];  // dense.rs:151
                                    }
                                }
                            }
                        } else {
#pragma unroll
                            for (int j = 0; j < NB; ++j) {
                                if (!((zm >> j) & 1u)) {  // dense.rs:148: a_kj == 0 -> no update of column j
                                    const double uj = s_u[kb + u][j];
#pragma unroll
                                    for (int ri = 0; ri < R; ++ri)
                                        if (act[ri]) a[ri][j] -= uj * l[ri][u];
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int ri = 0; ri < R; ++ri)
#pragma unroll
                        for (int u = 0; u < KC; ++u) l[ri][u] = ln[ri][u];
                }
            }
            // no barrier needed here: the next pass writes s_c/s_l/s_zm[(q+1)&1] only, s_u after its own barrier
        }
        __syncthreads();

        // ---------------------------------------------------------------- factor the block column in place
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            if (k < wd && !failed) {
                const int kc = c0 + k;
                double v = -1.0;
                int p = BIG;
#pragma unroll
                for (int ri = 0; ri < R; ++ri) {
                    if (step[ri] == BIG && row[ri] < n) {
                        double vi = fabs(a[ri][k]);
                        if (vi != vi) vi = (mypos[ri] == kc) ? __builtin_huge_val() : -1.0;  // NaN: dense.rs:111-117 scan
                        if (vi > v || (vi == v && mypos[ri] < p)) {
                            v = vi;
                            p = mypos[ri];
                        }
                    }
                }
                {   // wave arg-max on the DPP crossbar: max |a|, then the lowest position among the lanes that attain it
                    const double vm = wave_max_f64(v);
                    p = wave_min_i32(v == vm ? p : 0x7fffffff);
                    v = vm;
                }
                if (lane == 0) {
                    s_v[k & 1][wave] = v;
                    s_p[k & 1][wave] = p;
                }
                __syncthreads();
                double bv = s_v[k & 1][0];
                int bp = s_p[k & 1][0];
#pragma unroll
                for (int qq = 1; qq < NW; ++qq) {
                    const double ov = s_v[k & 1][qq];
                    const int op = s_p[k & 1][qq];
                    if (ov > bv || (ov == bv && op < bp)) {
                        bv = ov;
                        bp = op;
                    }
                }
                bool own[R];
#pragma unroll
                for (int ri = 0; ri < R; ++ri) {
                    own[ri] = (step[ri] == BIG) && (mypos[ri] == bp);
                    if (own[ri]) {
                        unsigned zm = 0u;
#pragma unroll
                        for (int j = 0; j < NB; ++j)
                            if (j >= k) {
                                s_prow[k & 1][j] = a[ri][j];
                                if (j > k && a[ri][j] == 0.0) zm |= 1u << j;
                            }
                        s_prow[k & 1][NB] = 1.0 / a[ri][k];  // mult = a(k,k).recip()  (dense.rs:134)
                        s_pz[k & 1] = zm;
                    }
                }
                __syncthreads();
                const double pk = s_prow[k & 1][k];
                if (pk == 0.0) {  // zero pivot: Err(k+1)  (dense.rs:120-122)
                    if (t == 0) w.info[b] = kc + 1;
                    failed = true;
                } else {
                    if (t == 0) piv[kc] = (long long)bp;
                    const double mult = s_prow[k & 1][NB];
                    const unsigned zm = (unsigned)__builtin_amdgcn_readfirstlane((int)s_pz[k & 1]);
#pragma unroll
                    for (int ri = 0; ri < R; ++ri) {
                        if (own[ri]) {
                            step[ri] = kc;
                            mypos[ri] = kc;
                        } else if (step[ri] == BIG && row[ri] < n) {
                            if (mypos[ri] == kc) mypos[ri] = bp;  // displaced row takes the pivot's old position
                            a[ri][k] *= mult;
                            const double aik = a[ri][k];
                            if (zm == 0u) {
#pragma unroll
                                for (int j = 0; j < NB; ++j)
                                    if (j > k && j < wd) a[ri][j] -= s_prow[k & 1][j] * aik;  // dense.rs:151
                            } else {
#pragma unroll
                                for (int j = 0; j < NB; ++j)
                                    if (j > k && j < wd && !((zm >> j) & 1u)) a[ri][j] -= s_prow[k & 1][j] * aik;
                            }
                        }
                    }
                }
            }
        }
        // write the block column back (physical row order)
#pragma unroll
        for (int ri = 0; ri < R; ++ri)
#pragma unroll
            for (int j = 0; j < NB; ++j)
                if (row[ri] < n && j < wd) A[(long)(c0 + j) * n + row[ri]] = a[ri][j];
        __syncthreads();  // make the new multipliers visible to every wave of this workgroup
    }
    if (failed) return;
    if (t == 0) w.info[b] = 0;

    // ---------------------------------------------------------------- scatter rows to their pivoted positions
#pragma unroll
    for (int ri = 0; ri < R; ++ri) {
        if (row[ri] < n && w.perm) w.perm[(long)b * n + mypos[ri]] = row[ri];
    }
    __syncthreads();
    double* __restrict__ O = w.out + (long)b * w.ostride;
    // thread <-> row within a column (coalesced reads, writes permuted inside the same 8n-byte column)
#pragma unroll 1
    for (int j0 = 0; j0 < n; j0 += 16) {
        double v[R][16];
#pragma unroll
        for (int jj = 0; jj < 16; ++jj)
#pragma unroll
            for (int ri = 0; ri < R; ++ri) v[ri][jj] = (row[ri] < n && j0 + jj < n) ? A[(long)(j0 + jj) * n + row[ri]] : 0.0;
#pragma unroll
        for (int jj = 0; jj < 16; ++jj)
#pragma unroll
            for (int ri = 0; ri < R; ++ri)
                if (row[ri] < n && j0 + jj < n) O[(long)(j0 + jj) * n + mypos[ri]] = v[ri][jj];
    }
}

inline bool lu_left_supported(int n) { return n > TINY_N && n <= 1024; }

inline int lu_left_launch(idahip_ctx* c, double* work, long wstride, double* out, long ostride, long long* piv, long pstride, int* perm,
                          const int* d_idx, int nsys) {
    LuLeftArgs a;
    a.mats = work; a.mstride = wstride; a.out = out; a.ostride = ostride; a.idx = d_idx; a.n = c->n; a.piv = piv; a.pstride = pstride;
    a.perm = perm; a.info = c->lu_info;
    const int n = c->n;
    if (n <= 256)
        hipLaunchKernelGGL((lu_left_kernel<32, 1, 256>), dim3(nsys), dim3(256), 0, c->stream, a);
    else if (n <= 512)
        hipLaunchKernelGGL((lu_left_kernel<32, 2, 256>), dim3(nsys), dim3(256), 0, c->stream, a);
    else
        hipLaunchKernelGGL((lu_left_kernel<32, 2, 512>), dim3(nsys), dim3(512), 0, c->stream, a);
    return 0;
}

}  // namespace idahip
