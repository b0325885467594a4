// Trailing update of the batched LU with the U operands in SCALAR registers (dense_get_rf's inner update,
// /root/reference/crates/linear/src/dense.rs:142-154, for everything right of a 64-column super-panel).
//
// Same contract and the same prologue as lu_trail64w_kernel (one workgroup per (matrix, 64 trailing columns): gather of the
// 64 pivot rows, U12 = L11^-1 A12 in three stages, U12 stored straight into the factors). What changed is how the rank-64
// update A22 -= L21 U12 gets its operands. There a lane owned a 4 x 4 tile and read four multipliers and four U entries
// from LDS per pivot: four ds_read_b128 for 32 fp64 operations, which keeps the LDS pipe busier than the arithmetic allows
// (lu_kernels.hpp) and needs 49 KB of LDS and 162 VGPRs per workgroup (three waves per SIMD). Here
//   * a lane owns ROWS of the trailing matrix: one row (or two), 16 columns of it in registers;
//   * the pivot row's 16 entries are the same for every lane of the wave, so they are *scalar* operands: the solved U12 block
//     is written once, row-major, to a scratch area in global memory and comes back through the scalar cache with
//     s_load_dwordx16 into SGPRs -- v_mul_f64 takes an SGPR pair directly. No LDS traffic at all in the update loop;
//   * the multiplier l(row, k) is one coalesced 8-byte load per lane and pivot straight from the work matrix (column k0 + k),
//     the four waves of a workgroup sweep the same 64-row tiles at the same time (each with its own 16 columns), so three of
//     the four requests hit the CU's vector L1.
// Per pivot and wave: one vector load, two scalar loads, 32 VALU operations (mul + sub per column, unfused as in dense.rs:151).
// ~100 VGPRs and 33 KB of LDS (only the prologue uses it): four workgroups per CU, four waves per SIMD, so one workgroup's
// serial prologue overlaps three others' arithmetic. L11 is read with scalar loads as well (it is uniform in the column-per-
// lane solve), so the prologue needs no LDS staging either.
// Every element still receives a(i,j) -= a_kj * a_ik in ascending k with the reference's operands; dense.rs:148 (a_kj == 0
// leaves the column untouched) is honoured as in lu_trail64w_kernel: verified for the whole column block in the prologue, a
// select path with the all-zero pivot rows skipped otherwise.
#pragma once
#include "lu_kernels.hpp"
#include "lu_wavepanel.hpp"

namespace idahip {

#ifndef IDAHIP_TS_RING
#define IDAHIP_TS_RING 8  // multipliers in flight per lane (pivots ahead)
#endif
#ifndef IDAHIP_TS_EXP
#define IDAHIP_TS_EXP 0  // timing builds: 1 = no scalar loads in the loop, 2 = no multiplier loads, 3 = neither (results are wrong)
#endif
#ifndef IDAHIP_TS_RPL
#define IDAHIP_TS_RPL 1  // rows per lane in the update loop
#endif
constexpr int TS_BLOCK = 65 * 64;  // doubles per (matrix, column block) in the U12 scratch area: 64 rows + one that is only ever requested

template <bool FMA, int RPL>
__global__ __launch_bounds__(256, RPL >= 3 ? 2 : 3) void lu_trail64s_kernel(LuWs w, double* __restrict__ uscr, int k0, int nsys, int ncb) {
    constexpr int NB = 64, KC = 32, CT = 16, D = IDAHIP_TS_RING;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int cbi = slot % ncb, mi = (slot / ncb) * 8 + xcd;
    if (mi >= nsys) return;
    const int b = w.idx[mi];
    if (w.info[b] != 0) return;
    const int n = w.n;
    double* __restrict__ A = w.mats + (long)b * w.mstride;
    const int* __restrict__ live = w.live + (long)b * n;
    const int* __restrict__ prow = w.prow + (long)b * n + k0;
    const double* __restrict__ l11 = w.l11 + (long)b * L11_STRIDE;  // l11ld == 64 here

    const int mrem = n - k0 - NB;  // live rows after this panel (> 0)
    const int cb0 = k0 + NB + cbi * 64;
    const int ncols = (n - cb0) < 64 ? (n - cb0) : 64;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);

    constexpr int LD = 65;  // odd row stride: the column reads of the store to the factors (lane = row) are conflict-free
    __shared__ double Us[NB][LD];
    __shared__ __align__(16) double Ls[KC][64];  // L11 staging, 32 rows at a time (one scalar load per solve step was far slower)
    __shared__ int s_anyzero;
    __shared__ int s_nz[4];          // per wave: a non-zero entry among the pivot-row entries it gathered
    __shared__ unsigned s_kmask[2];  // slow path: bit k of word R0 / 32 set = pivot row R0 + k has a non-zero entry in this column block

    __shared__ int s_rank;
    phase_stagger(w, &s_rank);
    // ---- 1. gather the 64 pivot rows of this column block (wave-uniform k per pass: prow[k] is a scalar load)
    bool nz = false;
#pragma unroll
    for (int pass = 0; pass < NB / 4; ++pass) {
        const int k = pass * 4 + wave;
        const int pr = ldc(prow + k);
        const double g = (lane < ncols) ? A[(long)(cb0 + lane) * n + pr] : 0.0;
        nz = nz || (g != 0.0);
        Us[k][lane] = g;
    }
    if (lane == 0) s_nz[wave] = 0;
    if (__ballot(nz) != 0ull && lane == 0) s_nz[wave] = 1;
    if (t == 0) s_anyzero = 0;
    auto stage_l11 = [&](const int R0) {
#pragma unroll
        for (int i = 0; i < (KC * 64) / 256; ++i) {
            const int e = i * 256 + t;
            Ls[e >> 6][e & 63] = l11[(R0 + (e >> 6)) * NB + (e & 63)];
        }
    };
    stage_l11(0);
    lds_barrier();
    double* __restrict__ O = w.out + (long)b * w.ostride;
    auto store_factors = [&]() {
        // pivot k of this super-panel is row k0 + k of the reference layout: a column's 64 entries are one contiguous 512-byte
        // store (one row per lane); lu_finalize_kernel skips this region
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const int cc = wave * 16 + i;
            if (cc < ncols) O[(long)(cb0 + cc) * n + k0 + lane] = Us[lane][cc];
        }
    };
    if ((s_nz[0] | s_nz[1] | s_nz[2] | s_nz[3]) == 0) {
        // the 64 pivot rows are zero across this whole column block (banded matrices, off the band): the triangular solve
        // leaves them as they are (a_kj == 0: column untouched, dense.rs:148) and nothing is subtracted from the rows below
        store_factors();
        return;
    }

    // ---- 2. U12 = L11^-1 A12 in three stages (as lu_trail64w_kernel), one column per lane
    auto trsm32 = [&](const int R0) {
        double u[KC];
#pragma unroll
        for (int k = 0; k < KC; ++k) u[k] = Us[R0 + k][lane];
        const bool real = lane < ncols;
        bool anyz = false;
#pragma unroll
        for (int kk = 0; kk < KC; ++kk) {
            const double ukk = u[kk];
            const bool z = real && (ukk == 0.0);
            anyz = anyz || z;
            if (__ballot(z) == 0ull) {
#pragma unroll
                for (int k = 0; k < KC; ++k)
                    if (k > kk) u[k] = upd<FMA>(u[k], ukk, Ls[kk][R0 + k]);  // a(i,j) -= a_kj * a_ik, ascending kk
            } else {
#pragma unroll
                for (int k = 0; k < KC; ++k)
                    if (k > kk) {
                        const double tn = upd<FMA>(u[k], ukk, Ls[kk][R0 + k]);
                        u[k] = z ? u[k] : tn;  // dense.rs:148: a_kj == 0 -> column untouched
                    }
            }
        }
#pragma unroll
        for (int k = 0; k < KC; ++k) Us[R0 + k][lane] = u[k];
        // dense.rs:148 skips the whole row update when a_kj == 0: a pivot row that is zero across this column block
        // contributes nothing to it (banded Jacobians). The mask is only read on the select path.
        unsigned km = 0xffffffffu;
        if (__ballot(anyz) != 0ull || s_anyzero != 0) {
            km = 0u;
#pragma unroll
            for (int k = 0; k < KC; ++k) km |= (__ballot(real && u[k] != 0.0) != 0ull) ? (1u << k) : 0u;
        }
        if (lane == 0) {
            s_kmask[R0 / KC] = km;
            if (__ballot(anyz) != 0ull) s_anyzero = 1;
        }
    };
    if (wave == 0) trsm32(0);
    lds_barrier();
    {   // rows 32..63 receive the updates of pivot rows 0..31: 8 rows per wave, one column per lane
        const bool zpath = s_anyzero != 0;
        double v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = Us[KC + wave * 8 + i][lane];
        if (!zpath) {
#pragma unroll
            for (int kk = 0; kk < KC; ++kk) {
                const double ut = Us[kk][lane];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = upd<FMA>(v[i], ut, Ls[kk][KC + wave * 8 + i]);
            }
        } else {
#pragma unroll 4
            for (int kk = 0; kk < KC; ++kk) {
                const double ut = Us[kk][lane];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const double tn = upd<FMA>(v[i], ut, Ls[kk][KC + wave * 8 + i]);
                    v[i] = (ut != 0.0) ? tn : v[i];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) Us[KC + wave * 8 + i][lane] = v[i];
    }
    lds_barrier();
    stage_l11(KC);
    lds_barrier();
    if (wave == 0) trsm32(KC);
    lds_barrier();
    const bool slow = s_anyzero != 0;
    const unsigned kmask0 = (unsigned)__builtin_amdgcn_readfirstlane((int)s_kmask[0]);
    const unsigned kmask1 = (unsigned)__builtin_amdgcn_readfirstlane((int)s_kmask[1]);
    store_factors();
    if (slow && (kmask0 | kmask1) == 0u) return;  // (uniform over the workgroup) U12 of this block is all zeros: nothing to subtract

    // ---- 3. U12, row-major, to this workgroup's scratch block: the update loop reads it back through the scalar cache.
    // The block is written and read by this workgroup only and by no earlier wave of this launch (scalar caches are
    // invalidated at kernel start), so a scalar load can never see a stale line. A block has 65 rows: the loop requests
    // row k + 1 while it works on row k, and row 64 exists (never written, never used) so that the request needs no clamp.
    const double* up;
    {
        double* __restrict__ us = uscr + ((long)b * ((n + 63) >> 6) + cbi) * TS_BLOCK;
#pragma unroll 4
        for (int i = 0; i < 16; ++i) us[(wave * 16 + i) * 64 + lane] = Us[wave * 16 + i][lane];
        up = us;
    }
    __syncthreads();  // drains the stores above (vmcnt(0)) before any wave reads them back
    asm volatile("" : "+s"(up)::"memory");  // the loads below cannot be scheduled above this point
    if (IDAHIP_TS_EXP & 4) return;

    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(A, 0, n * n * 8, 0x00020000);
    constexpr int TR = 64 * RPL;  // rows of a tile
    const int nrt = (mrem + TR - 1) / TR;
    const int nct = (ncols + CT - 1) / CT;
    const int n8 = n * 8;
    const int lbase = k0 * n8;  // byte offset of the super-panel's first column

    // tasks: (row tile, 16-column tile). IDAHIP_TS_MAP 0: a wave keeps its column tile and the four waves sweep the row tiles
    // together (they share the multipliers through the vector L1); 1: a wave keeps its row tiles and the four waves sweep
    // the column tiles together (they share the U rows through the scalar cache)
#ifndef IDAHIP_TS_MAP
#define IDAHIP_TS_MAP 0
#endif
    const int ntask = nrt * 4;
#pragma unroll 1
    for (int task = wave; task < ntask; task += 4) {
        const int ct = IDAHIP_TS_MAP ? (task >> 2) % 4 : (task & 3);
        const int rt = IDAHIP_TS_MAP ? (task >> 4) * 4 + (task & 3) : (task >> 2);
        if (ct >= nct || rt >= nrt) continue;
        const int c0 = cb0 + ct * CT;
        const int ncw = (ncols - ct * CT) < CT ? (ncols - ct * CT) : CT;  // columns of the tile that exist
        const double* upw = up + ct * CT;
        bool rok[RPL];
        int row[RPL];
        unsigned roff[RPL];
        double c[RPL][CT];
#pragma unroll
        for (int r = 0; r < RPL; ++r) {
            const int li = rt * TR + r * 64 + lane;
            rok[r] = li < mrem;
            row[r] = live[rok[r] ? li : mrem - 1];
            roff[r] = (unsigned)row[r] * 8u;
        }
#pragma unroll
        for (int r = 0; r < RPL; ++r)
#pragma unroll
            for (int j = 0; j < CT; ++j) c[r][j] = buf_load_f64(rsrc, roff[r], (c0 + (j < ncw ? j : 0)) * n8);
        if (!slow) {
            // the tile has landed before the multiplier ring is primed: with loads of two kinds in flight at the loop's entry the
            // compiler's wait-count bookkeeping waits for *every* load at the top of every iteration (hardware lessons)
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
            double l[D][RPL];
            int lo = lbase;
#pragma unroll
            for (int d = 0; d < D; ++d) {
#pragma unroll
                for (int r = 0; r < RPL; ++r) l[d][r] = buf_load_f64(rsrc, roff[r], lo);
                lo += n8;
            }
            double ua[CT], ub[CT];
#pragma unroll
            for (int j = 0; j < CT; ++j) ua[j] = ldc(upw + j);
            const double* upk = upw + 64;  // row k + 1
#pragma unroll 1
            for (int k = 0; k < NB; k += D) {
                static_for<0, D>([&](auto dt) {
                    constexpr int d = decltype(dt)::value;
                    double(&ucur)[CT] = (d & 1) ? ub : ua;
                    double(&unext)[CT] = (d & 1) ? ua : ub;
                    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): ucur has landed; the request below is then the only one in flight
#pragma unroll
                    for (int j = 0; j < CT; ++j) unext[j] = (IDAHIP_TS_EXP & 1) ? ucur[j] : ldc(upk + d * 64 + j);
                    __builtin_amdgcn_sched_barrier(0);  // the request stays in front of this pivot's arithmetic
#pragma unroll
                    for (int r = 0; r < RPL; ++r) {
                        const double lk = l[d][r];
#pragma unroll
                        for (int j = 0; j < CT; ++j) c[r][j] = upd<FMA>(c[r][j], ucur[j], lk);  // dense.rs:151
                    }
                    // (past the super-panel's last column this reads trailing columns, or nothing beyond the matrix's end: the
                    // buffer descriptor bounds the access; the value is never used)
#pragma unroll
                    for (int r = 0; r < RPL; ++r)
                        if (!(IDAHIP_TS_EXP & 2)) l[d][r] = buf_load_f64(rsrc, roff[r], lo);
                    lo += n8;
                    __builtin_amdgcn_sched_barrier(0);
                });
                upk += D * 64;
            }
        } else {
            // select path: pivot rows that are zero across the block skipped, a_kj == 0 leaves the column untouched (dense.rs:148)
#pragma unroll 1
            for (int h = 0; h < 2; ++h) {
                for (unsigned mk = h == 0 ? kmask0 : kmask1; mk != 0u; mk &= mk - 1u) {
                    const int k = h * 32 + __builtin_ctz(mk);
                    double lk[RPL];
#pragma unroll
                    for (int r = 0; r < RPL; ++r) lk[r] = buf_load_f64(rsrc, roff[r], lbase + k * n8);
#pragma unroll
                    for (int j = 0; j < CT; ++j) {
                        const double uj = ldc(upw + k * 64 + j);
                        if (uj != 0.0) {
#pragma unroll
                            for (int r = 0; r < RPL; ++r) c[r][j] = upd<FMA>(c[r][j], uj, lk[r]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RPL; ++r)
            if (rok[r]) {
#pragma unroll
                for (int j = 0; j < CT; ++j)
                    if (j < ncw) A[(long)(c0 + j) * n + row[r]] = c[r][j];
            }
    }
}

}  // namespace idahip
