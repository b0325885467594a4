// A whole 64-column super-panel of a large matrix (n > 1024: config 4's 4096 x 4096 Jacobians, ~100 matrices per call) in ONE
// launch (dense_get_rf, /root/reference/crates/linear/src/dense.rs:86-158, restricted to the super-panel's columns): one workgroup
// per matrix, R live rows per lane in registers (8 above 2048 live rows, 4 above 1024, 2 below), left-looking over eight blocks of
// 8 columns.
//
// Why: until round 4 such a super-panel was 8 launches of lu_panelr_kernel (8 columns each) with a narrow lu_trail_kernel launch
// after each to carry the panel's update to the rest of the super-panel -- 15 launches, and of a panel launch's 38 us at 4096
// live rows only 22 are the 8 pivot steps: 14 us are the dependent loads at the start (live list -> row ids -> positions ->
// entries) and the rest stores, L11 and the compaction of the live list at the end. A factorisation of config 4's 4096 x 4096
// Jacobians is a chain of ~640 such launches per matrix, and that chain, not the chip, bounds it. Here
// the row ids, positions and alive flags stay in registers across the eight blocks, the list is compacted once, and the narrow
// updates disappear: a block's columns are brought up to date when the block is loaded (left-looking), by
//   * the "U slot" of wave 0: lane k holds pivot row k of this super-panel (its entries in the block's 8 columns, loaded with the
//     block; prow says which physical row that is). Wave 0 solves U = L11^-1 A12 on it in ascending k -- the pivot's entries
//     broadcast with v_readlane, the multipliers of the later pivot rows from the transposed L11 kept in LDS --, writes the rows
//     back to the work matrix and leaves them in LDS;
//   * every wave then applies those rows to its live rows, ascending k, multipliers re-read from the work matrix.
// Both honour dense.rs:148 per entry (a_kj == 0: the column is left untouched) and skip pivot rows that are zero across the
// whole block -- for a banded matrix (the heat equation's Jacobian) that is all but one or two per block, which is what makes
// the left-looking form cheap there; for a dense matrix it re-reads the super-panel's multipliers as the narrow updates did.
// The 8 pivot steps of a block are lu_panelr_kernel's, unchanged (arg-max per lane, DPP per wave, LDS hand-off, one barrier).
// Every element receives a(i,j) -= a_kj * a_ik for the same k in the same ascending order as before, unfused.
#pragma once
#include "lu_kernels.hpp"

namespace idahip {

template <int R, int MAXT, int WPE>
__global__ __launch_bounds__(MAXT, WPE) void lu_superpanel_kernel(LuWs w, const int k0) {
    constexpr int NB = 8;            // columns of a block
    constexpr int NW = MAXT / 64;
    constexpr int LDR = NB + 2;      // row slot of the pivot hand-off: NB entries, [NB] = 1/pivot
    static_assert(NW <= 16, "candidate scan assumes <= 16 waves");
    if (w.cnt && (int)blockIdx.x >= ldc(w.cnt)) return;  // (list length on the device: surplus workgroups leave)
    const int b = w.idx[blockIdx.x];
    if (w.info[b] != 0) return;
    const int n = w.n;
    double* __restrict__ A = w.mats + (long)b * w.mstride;
    int* __restrict__ pos = w.pos + (long)b * n;
    int* __restrict__ live = w.live + (long)b * n;
    int* __restrict__ prow = w.prow + (long)b * n;
    long long* __restrict__ piv = w.piv + (long)b * w.pstride;
    double* __restrict__ l11 = w.l11 + (long)b * L11_STRIDE;  // transposed L11 of the super-panel, row length w.l11ld (64)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(A, 0, n * n * 8, 0x00020000);

    const int m = n - k0;      // live rows when the super-panel starts: >= 64 (lu_driver.hpp), so the super-panel has its 64 columns
    const int T = blockDim.x;  // R * T >= m
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);

    __shared__ __align__(16) double s_row[2][NW][LDR];
    __shared__ unsigned s_kh[2][16], s_kl[2][16];
    __shared__ __align__(16) int s_p[2][16];
    __shared__ int s_r[2][NW];
    __shared__ int s_cnt[R][NW];
    __shared__ __align__(16) double s_l11[64][64];  // [kk][k]: multiplier of pivot row k of the super-panel for its column kk < k
    __shared__ __align__(16) double s_u[64][NB];    // the solved rows of the U slot for the current block
    __shared__ unsigned long long s_umask;          // bit k: row k of s_u has a non-zero entry
    __shared__ int s_prow[64];                      // physical rows of the super-panel's pivots so far

    if (t < 32) {  // slots of waves that do not exist in this launch never win
        (&s_kh[0][0])[t] = 0u;
        (&s_kl[0][0])[t] = 0u;
        (&s_p[0][0])[t] = 0x7fffffff;
    }
    // which of the lane's R rows exist / are still live: bit i of one register each. (As bool arrays they are lane masks in
    // scalar register pairs; with 2 x R of them next to the kernel's other scalars the SGPRs ran out and were spilled into vector
    // registers, which the panel needs: 164 B of scratch per lane at R = 6.)
    unsigned valid_bits = 0u, alive_bits = 0u;
    auto is_valid = [&](const int i) { return ((valid_bits >> i) & 1u) != 0u; };
    auto is_alive = [&](const int i) { return ((alive_bits >> i) & 1u) != 0u; };
    int r[R], mypos[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int li = t + i * T;
        valid_bits |= (li < m) ? (1u << i) : 0u;
        r[i] = live[li < m ? li : m - 1];
    }
    alive_bits = valid_bits;
#pragma unroll
    for (int i = 0; i < R; ++i) mypos[i] = pos[r[i]];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        if (!is_valid(i)) {
            r[i] = 0;
            mypos[i] = 0x7fffffff;
        }
    }

#pragma unroll 1
    for (int lb = 0; lb < 64; lb += NB) {
        const int kb = k0 + lb;  // first column of the block
        __syncthreads();         // the previous block's stores (work matrix, s_l11, s_prow) are visible
        double a[R][NB];
        // every load is unconditional (indices are valid for every lane; results of rows that do not exist are masked)
#pragma unroll
        for (int i = 0; i < R; ++i) {
#pragma unroll
            for (int j = 0; j < NB; ++j) a[i][j] = buf_load_f64(rsrc, (unsigned)r[i] * 8u, (kb + j) * n * 8);
        }
        auto mask_rows = [&]() {  // (first use of the block's entries: everything above it is issued while they are in flight)
#pragma unroll
            for (int i = 0; i < R; ++i) {
#pragma unroll
                for (int j = 0; j < NB; ++j) a[i][j] = is_valid(i) ? a[i][j] : 0.0;
            }
        };
        // the U slot's rows (wave 0: lane k = pivot row k of the super-panel), requested with the block: one round trip less per block
        const bool uhas = wave == 0 && lane < lb;
        const int urow = s_prow[uhas ? lane : 0];
        double xu[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) xu[j] = buf_load_f64(rsrc, uhas ? (unsigned)urow * 8u : 0xfffffff0u, (kb + j) * n * 8);  // (other lanes: out of range, +0.0, no memory access)
        if (lb > NB) {
            // multipliers of the previous block's 8 pivot rows for the super-panel's columns left of that block: one entry per
            // thread (a serial read-back by the lanes that own those rows was a chain of lb / 8 round trips to memory per block)
            const int pb = lb - NB;  // first pivot of the previous block
            for (int e = t; e < NB * pb; e += T) {
                const int q = pb + e / pb, kk = e % pb;
                const double v = A[(long)(k0 + kk) * n + s_prow[q]];
                s_l11[kk][q] = v;
                l11[kk * w.l11ld + q] = v;
            }
        }
        if (lb > 0) {
            __syncthreads();  // s_l11 is complete for the pivots 0 .. lb-1
            // ---- U slot (wave 0): rows k < lb of U in this block's columns
            if (wave == 0) {
                int next = 0;
#pragma unroll 1
                for (;;) {
                    // the lowest pivot row k >= next that has a non-zero (or NaN) entry in the block: a row of zeros changes
                    // nothing below it (dense.rs:148)
                    bool rownz = false;
#pragma unroll
                    for (int j = 0; j < NB; ++j) rownz = rownz || !(xu[j] == 0.0);
                    const unsigned long long nz = __ballot(rownz && uhas) & ~((1ull << next) - 1ull);
                    if (nz == 0ull) break;
                    const int k = __builtin_ctzll(nz);
                    next = k + 1;
                    if (next >= lb) break;  // the last row has nobody below it
                    const double lk = s_l11[k][lane];  // multiplier of pivot row `lane` for column k (lanes <= k: never written, never used)
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        const double ukj = readlane_f64(xu[j], k);
                        const double tn = xu[j] - ukj * lk;  // a(i,j) -= a_kj * a_ik, unfused
                        xu[j] = (lane > k && uhas && ukj != 0.0) ? tn : xu[j];  // dense.rs:148: a_kj == 0 -> column untouched
                    }
                }
                bool rownz = false;
#pragma unroll
                for (int j = 0; j < NB; ++j) rownz = rownz || !(xu[j] == 0.0);
                const unsigned long long um = __ballot(rownz && uhas);
                if (lane == 0) s_umask = um;
                if (uhas) {
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        s_u[lane][j] = xu[j];
                        buf_store_f64(rsrc, (unsigned)urow * 8u, (kb + j) * n * 8, xu[j]);  // final: row k of U in these columns
                    }
                }
            }
            __syncthreads();
        }
        // (the block's registers are only touched outside of branches: a conditional around the panel registers makes the allocator
        // keep two copies of them)
        mask_rows();
        {
            // ---- every wave: the live rows receive the updates of pivots 0 .. lb-1 in ascending k
            unsigned long long um = lb > 0 ? s_umask : 0ull;
            um = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(um >> 32)) << 32) |
                 (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)um);
            if constexpr (R <= 4) {
                // The multipliers of the NEXT pivot row with work are requested before the arithmetic of the current one: on a dense
                // matrix every pivot has work, and a loop that loads, waits and computes per pivot is one round trip to memory per
                // pivot -- lb of them per block (a banded matrix has one pivot with work per block and does not care).
                double lcur[R], lnxt[R];
                auto load_l = [&](double (&l)[R], const int k) {
#pragma unroll
                    for (int i = 0; i < R; ++i) l[i] = buf_load_f64(rsrc, (unsigned)r[i] * 8u, (k0 + k) * n * 8);
                };
                load_l(lcur, um != 0ull ? __builtin_ctzll(um) : 0);
#pragma unroll 1
                for (; um != 0ull; um &= um - 1ull) {
                    const int k = __builtin_ctzll(um);
                    const unsigned long long rest = um & (um - 1ull);
                    load_l(lnxt, rest != 0ull ? __builtin_ctzll(rest) : k);  // (past the last pivot: the same column again, never used)
                    double u[NB];
                    bool uz[NB];
#pragma unroll
                    for (int j = 0; j < NB; j += 2) {
                        const double2 q = *reinterpret_cast<const double2*>(&s_u[k][j]);
                        u[j] = opaque_vgpr(q.x);
                        u[j + 1] = opaque_vgpr(q.y);
                        uz[j] = u[j] == 0.0;
                        uz[j + 1] = u[j + 1] == 0.0;
                    }
#pragma unroll
                    for (int i = 0; i < R; ++i) {
#pragma unroll
                        for (int j = 0; j < NB; ++j) {
                            const double tn = a[i][j] - u[j] * lcur[i];
                            a[i][j] = (is_alive(i) && !uz[j]) ? tn : a[i][j];  // dense.rs:148-151
                        }
                    }
#pragma unroll
                    for (int i = 0; i < R; ++i) lcur[i] = lnxt[i];
                }
            } else {
                // eight rows per lane: no registers to spare for a second set of multipliers (it spilled); four at a time
#pragma unroll 1
                for (; um != 0ull; um &= um - 1ull) {
                    const int k = __builtin_ctzll(um);
                    double u[NB];
                    bool uz[NB];
#pragma unroll
                    for (int j = 0; j < NB; j += 2) {
                        const double2 q = *reinterpret_cast<const double2*>(&s_u[k][j]);
                        u[j] = opaque_vgpr(q.x);
                        u[j + 1] = opaque_vgpr(q.y);
                        uz[j] = u[j] == 0.0;
                        uz[j + 1] = u[j + 1] == 0.0;
                    }
                    constexpr int RH = 4;
                    static_assert(R % RH == 0, "the passes cover the lane's rows exactly");
#pragma unroll
                    for (int i0 = 0; i0 < R; i0 += RH) {
                        double l[RH];
#pragma unroll
                        for (int i = 0; i < RH; ++i) l[i] = buf_load_f64(rsrc, (unsigned)r[i0 + i] * 8u, (k0 + k) * n * 8);
#pragma unroll
                        for (int i = 0; i < RH; ++i) {
#pragma unroll
                            for (int j = 0; j < NB; ++j) {
                                const double tn = a[i0 + i][j] - u[j] * l[i];
                                a[i0 + i][j] = (is_alive(i0 + i) && !uz[j]) ? tn : a[i0 + i][j];  // dense.rs:148-151
                            }
                        }
                    }
                }
            }
        }

        // ---- the block's 8 pivot steps (lu_panelr_kernel's step, with the block's first column as origin)
        bool failed = false;
        auto step = [&](auto kconst) -> bool {
            constexpr int k = decltype(kconst)::value;
            const int kc = kb + k;
            const int par = k & 1;
            int sel = 0;
            unsigned bkh = 0u, bkl = 0u;
            int bpos = mypos[0];
            double a0 = a[0][k];
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const double v = fabs(a[i][k]);
                const bool isnan_ = v != v;
                unsigned kh = (unsigned)__double2hiint(v) | 0x80000000u, kl = (unsigned)__double2loint(v);
                kh = isnan_ ? ((mypos[i] == kc) ? 0xfff00000u : 0u) : kh;
                kl = isnan_ ? 0u : kl;
                kh = is_alive(i) ? kh : 0u;
                kl = is_alive(i) ? kl : 0u;
                const bool better = i == 0 || kh > bkh || (kh == bkh && (kl > bkl || (kl == bkl && mypos[i] < bpos)));
                sel = better ? i : sel;
                bkh = better ? kh : bkh;
                bkl = better ? kl : bkl;
                bpos = better ? mypos[i] : bpos;
                a0 = better ? a[i][k] : a0;
            }
            const double myrecip = 1.0 / a0;  // mult = a(k,k).recip() (dense.rs:134), off the critical path
            const unsigned mh = wave_max_u32<false>(bkh);
            const unsigned ml = wave_max_u32<false>(bkh == mh ? bkl : 0u);
            const bool top = bkh != 0u && bkh == mh && bkl == ml;
            const int pm = wave_min_i32f<false>(top ? bpos : 0x7fffffff);
            const bool cand = top && bpos == pm;  // this wave's candidate row (one lane, or none)
            const int jb = k & ~1;                // row slots are written in aligned pairs
#pragma unroll
            for (int i = 0; i < R; ++i) {
                if (cand && sel == i) {
#pragma unroll
                    for (int j = jb; j < NB; j += 2) {
                        double2 q;
                        q.x = a[i][j];
                        q.y = a[i][j + 1];
                        *reinterpret_cast<double2*>(&s_row[par][wave][j]) = q;
                    }
                    s_r[par][wave] = r[i];
                }
            }
            if (cand) s_row[par][wave][NB] = myrecip;
            if (lane == 0) {
                s_kh[par][wave] = mh;  // 0 when the wave has no live row
                s_kl[par][wave] = ml;
                s_p[par][wave] = pm;
            }
            lds_barrier();
            int bp, bw;
            {
                const int q = lane & 15;
                const unsigned ch = s_kh[par][q], cl = s_kl[par][q];
                const int cp = s_p[par][q];
                const unsigned bh = wave_max_u32<true>(ch);
                const unsigned bl = wave_max_u32<true>(ch == bh ? cl : 0u);
                const int kmin = wave_min_i32f<true>((ch == bh && cl == bl) ? ((cp << 4) | q) : 0x7fffffff);
                bp = kmin >> 4;
                bw = kmin & 15;
            }
            const double pk = s_row[par][bw][k];
            if (pk == 0.0) {  // zero pivot: Err(k+1)  (dense.rs:120-122)
                if (t == 0) w.info[b] = kc + 1;
                return false;
            }
            if (t == 0) piv[kc] = (long long)bp;
            {
                bool anyp = false;
                int pr_mine = 0;
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    const bool isp = is_alive(i) && mypos[i] == bp;           // this row is the pivot
                    const bool mv = is_alive(i) && !isp && mypos[i] == kc;    // the row that sat at position k moves to the pivot's old position
                    anyp = anyp || isp;
                    pr_mine = isp ? r[i] : pr_mine;
                    mypos[i] = isp ? kc : (mv ? bp : mypos[i]);
                    alive_bits &= isp ? ~(1u << i) : ~0u;
                }
                if (anyp) {
                    prow[kc] = pr_mine;
                    s_prow[lb + k] = pr_mine;
                }
            }
            // the pivot row is final for the block's columns: one cooperative store of pivot + U entries from the LDS copy
            if (wave == 0 && lane >= k && lane < NB) A[(long)(kb + lane) * n + s_r[par][bw]] = s_row[par][bw][lane];
            const double recip = s_row[par][bw][NB];
            double u[NB];
            bool uz[NB];
#pragma unroll
            for (int j = jb; j < NB; j += 2) {
                const double2 q = *reinterpret_cast<const double2*>(&s_row[par][bw][j]);
                u[j] = opaque_vgpr(q.x);
                u[j + 1] = opaque_vgpr(q.y);
                uz[j] = u[j] == 0.0;
                uz[j + 1] = u[j + 1] == 0.0;
            }
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const double aik = a[i][k] * recip;  // dense.rs:134-137
                a[i][k] = aik;
#pragma unroll
                for (int j = k + 1; j < NB; ++j) a[i][j] = uz[j] ? a[i][j] : a[i][j] - u[j] * aik;  // dense.rs:148-151
            }
            return true;
        };
        failed = !static_steps<0, NB>(step, NB);
        if (failed) return;  // (uniform: every thread read the same pivot)

        // ---- the block's multipliers leave: a live row has one in every column of the block, a row that became a pivot in
        // this block in the columns left of its own step; rows that were pivots before the block have nothing here
        int lim[R];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int own = mypos[i] - kb;  // (pivot rows: their position is their pivot column)
            lim[i] = !is_valid(i) ? 0 : is_alive(i) ? NB : (own >= 0 && own < NB) ? own : 0;
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
#pragma unroll
            for (int i = 0; i < R; ++i)
                if (j < lim[i]) buf_store_f64(rsrc, (unsigned)r[i] * 8u, (kb + j) * n * 8, a[i][j]);
        }
        // transposed L11 rows of the pivots chosen in this block, the part inside the block (columns lb .. own - 1) from
        // registers -- into LDS for the U slots of the blocks to come and into LuWs::l11 for lu_trail64w_kernel. The part left of
        // the block (the super-panel's earlier columns, stored to the matrix when those blocks ended) is read back by the whole
        // workgroup, one entry per thread, when the next block starts.
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int own = mypos[i] - kb;
            if (is_valid(i) && !is_alive(i) && own >= 0 && own < NB) {
                const int kq = lb + own;  // index of this pivot row inside the super-panel
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    if (j < own) {
                        s_l11[lb + j][kq] = a[i][j];
                        l11[(lb + j) * w.l11ld + kq] = a[i][j];
                    }
            }
        }
    }
    // the last block's pivot rows: their multipliers left of the block, for lu_trail64w_kernel's L11
    __syncthreads();
    for (int e = t; e < NB * (64 - NB); e += T) {
        const int q = 64 - NB + e / (64 - NB), kk = e % (64 - NB);
        l11[kk * w.l11ld + q] = A[(long)(k0 + kk) * n + s_prow[q]];
    }

    // ---- positions and the compacted live list, once per super-panel
#pragma unroll
    for (int i = 0; i < R; ++i)
        if (is_valid(i)) pos[r[i]] = mypos[i];
    unsigned long long bal[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        bal[i] = __ballot(is_alive(i));
        if (lane == 0) s_cnt[i][wave] = __popcll(bal[i]);
    }
    __syncthreads();
    const int nwaves = T >> 6;
    int base = 0;
#pragma unroll
    for (int i = 0; i < R; ++i) {
        int mine = base;
        for (int q = 0; q < nwaves; ++q) {
            const int c = s_cnt[i][q];
            if (q < wave) mine += c;
            base += c;
        }
        if (is_alive(i)) live[mine + __popcll(bal[i] & ((1ull << lane) - 1ull))] = r[i];
    }
}

}  // namespace idahip
