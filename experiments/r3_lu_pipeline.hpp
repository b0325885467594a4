// Batched LU, n <= 512: the trailing work of a super-panel step as a PIPELINE of pure kernels instead of one fused kernel
// (dense_get_rf, /root/reference/crates/linear/src/dense.rs:86-158; same arithmetic per element as lu_trail64w_kernel).
//
// lu_trail64w_kernel does three things per (matrix, 64 columns): it gathers the 64 pivot rows (8 useful bytes per 64-byte
// line of a column-major matrix), solves U12 = L11^-1 A12 on ONE of its four waves -- a 64-step dependent chain, 30-50 us during
// which the other three wait and the workgroup's registers and LDS sit idle -- and only then updates the live rows. Time
// stamps inside the kernel put that prologue at a third of a workgroup's life (tools/stamps.py on the experimental branch).
// Here the three are separate launches, each dense in the work it does:
//   lu_pack_rows_kernel  (first step only) the pivot rows of super-panel 0, row-major, into P;
//   lu_u12_kernel        U12 = L11^-1 P for every column block: one wave per block, one column per lane, all 64 entries of the
//                        column in registers, four blocks per workgroup share one LDS copy of L11; writes U12 row-major;
//   lu_update_kernel     A22 -= L21 U12 in wave-private 16-row strips with the 4 x 4 register tile of lu_trail64w_kernel, and
//                        nothing else: no gather, no solve, one barrier. It also writes U12 into the factors.
// The pivot rows of the NEXT super-panel never have to be gathered: the update of step s runs in two launches around the next
// panel factorisation (look-ahead),
//   update(s, column block 0) -> lu_wavepanel(s + 1) -> update(s, column blocks 1..)
// so the second launch knows the next 64 pivot rows. It processes them as four extra strips whose results go, row-major, to
// P instead of back into the matrix (nothing reads those entries in the matrix again): P(s + 1) falls out of the update as
// contiguous 32-byte pieces instead of a second sweep over every line of the trailing matrix.
// Every element still receives a(i,j) -= a_kj * a_ik in ascending k, unfused, and dense.rs:148 (a_kj == 0 leaves the column
// untouched) is honoured as before: the solve records whether any entry of a block's U12 is an exact zero and which pivot rows
// are zero across the block; the update takes its select path then.
#pragma once
#include "lu_kernels.hpp"

namespace idahip {

constexpr int U12_BLOCK = 64 * 64;  // doubles per (matrix, column block) of the row-major U12 scratch
constexpr int U12_FLAGS = 4;        // ints per (matrix, column block): any exact zero, kmask word 0, kmask word 1, unused

struct LuPipe {
    double* u12;   // [batch][ceil(n/64)][64][64]
    int* uflag;    // [batch][ceil(n/64)][U12_FLAGS]
    double* pbuf;  // [batch][64][n] pivot rows of the current super-panel, row-major (absolute column index)
};

// P[k][j] = work(prow[k0 + k], j) for the columns right of the super-panel (first step only)
__global__ __launch_bounds__(256) void lu_pack_rows_kernel(LuWs w, LuPipe p, int k0, int nsys, int ncb) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int cbi = slot % ncb, mi = (slot / ncb) * 8 + xcd;
    if (mi >= (w.cnt ? ldc(w.cnt) : nsys)) return;
    const int b = w.idx[mi];
    if (w.info[b] != 0) return;
    const int n = w.n;
    const double* __restrict__ A = w.mats + (long)b * w.mstride;
    const int* __restrict__ prow = w.prow + (long)b * n + k0;
    const int cb0 = k0 + 64 + cbi * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (cb0 + lane >= n) return;
    for (int k = wave; k < 64; k += 4) p.pbuf[((long)b * 64 + k) * n + cb0 + lane] = A[(long)(cb0 + lane) * n + prow[k]];
}

// U12 = L11^-1 P: one wave per column block, four blocks per workgroup
template <bool FMA>
__global__ __launch_bounds__(256, 2) void lu_u12_kernel(LuWs w, LuPipe p, int k0, int nsys, int ncb) {
    constexpr int NB = 64;
    const int ngrp = (ncb + 3) >> 2;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int grp = slot % ngrp, mi = (slot / ngrp) * 8 + xcd;
    if (mi >= (w.cnt ? ldc(w.cnt) : nsys)) return;
    const int b = w.idx[mi];
    if (w.info[b] != 0) return;
    const int n = w.n;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int cbi = grp * 4 + wave;
    const bool have = cbi < ncb;  // (wave-uniform) this wave's column block exists
    const int cb0 = k0 + NB + cbi * 64;
    const int ncols = have ? ((n - cb0) < 64 ? (n - cb0) : 64) : 0;
    const int nblk = (n + 63) >> 6;

    __shared__ __align__(16) double Ls[NB * NB + 64];  // Ls[kk * 64 + k] = multiplier of pivot row k for column kk (+ slack for the
                                                       // requests of the window's slots past the end)
    {
        const double* __restrict__ l11 = w.l11 + (long)b * L11_STRIDE;
#pragma unroll
        for (int i = 0; i < (NB * NB) / 256; ++i) Ls[i * 256 + t] = l11[i * 256 + t];
    }
    const bool real = lane < ncols;
    double u[NB];
    {
        const double* __restrict__ P = p.pbuf + (long)b * 64 * n + cb0 + lane;
#pragma unroll
        for (int k = 0; k < NB; ++k) u[k] = real ? P[(long)k * n] : 0.0;
    }
    __syncthreads();
    if (!have) return;
    double* __restrict__ U = p.u12 + ((long)b * nblk + cbi) * U12_BLOCK + lane;
    bool anyz = false;
    unsigned km0 = 0u, km1 = 0u;
    // The register file has no dynamic indexing, so the lane's window of entries rotates by one place per pivot: u[0] is the
    // entry of pivot row kk (final), u[j] the entry of row kk + j; the update writes u[j] = u[j+1] - ukk * l. One copy of the
    // step's code per window width (63, 47, 31, 15 candidates for pivots 0-15, 16-31, 32-47, 48-63).
    auto step = [&](auto wtag, const int kk) {
        constexpr int W = decltype(wtag)::value;
        const double ukk = u[0];
        U[kk * 64] = ukk;
        const bool z = real && (ukk == 0.0);
        anyz = anyz || z;
        // dense.rs:148 skips the whole row update when a_kj == 0: a pivot row that is zero across this column block contributes
        // nothing to it. The mask is only read on the select path of the update.
        const bool nzrow = __ballot(real && ukk != 0.0) != 0ull;
        if (kk < 32) km0 |= nzrow ? (1u << kk) : 0u;
        else km1 |= nzrow ? (1u << (kk - 32)) : 0u;
        const double* __restrict__ lrow = &Ls[kk * 64 + kk + 1];  // multipliers of rows kk + 1 + j for this column
        if (__ballot(z) == 0ull) {
#pragma unroll
            for (int j = 0; j < W; ++j) u[j] = upd<FMA>(u[j + 1], ukk, lrow[j]);  // a(i,j) -= a_kj * a_ik, ascending kk
        } else {
#pragma unroll
            for (int j = 0; j < W; ++j) {
                const double tn = upd<FMA>(u[j + 1], ukk, lrow[j]);
                u[j] = z ? u[j + 1] : tn;  // dense.rs:148: a_kj == 0 -> column untouched
            }
        }
    };
#pragma unroll 1
    for (int kk = 0; kk < 16; ++kk) step(std::integral_constant<int, 63>{}, kk);
#pragma unroll 1
    for (int kk = 16; kk < 32; ++kk) step(std::integral_constant<int, 47>{}, kk);
#pragma unroll 1
    for (int kk = 32; kk < 48; ++kk) step(std::integral_constant<int, 31>{}, kk);
#pragma unroll 1
    for (int kk = 48; kk < 64; ++kk) step(std::integral_constant<int, 15>{}, kk);
    const bool anyzw = __ballot(anyz) != 0ull;
    if (lane == 0) {
        int* f = p.uflag + ((long)b * nblk + cbi) * U12_FLAGS;
        f[0] = anyzw ? 1 : 0;
        f[1] = (int)km0;
        f[2] = (int)km1;
    }
}

// A22 -= L21 U12 for column blocks [cb_first, cb_first + cb_count) of this step.
// NEXTP: the launch runs after the next super-panel has been factored: its 64 pivot rows (no longer in the live list) are
// updated first, as four strips whose results go to P (row-major) instead of the matrix.
template <bool FMA, bool NEXTP>
__global__ __launch_bounds__(256, 3) void lu_update_kernel(LuWs w, LuPipe p, int k0, int nsys, int cb_first, int cb_count) {
    constexpr int NB = 64, KC = 32, MAXROWS = 512;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int cbi = cb_first + slot % cb_count, mi = (slot / cb_count) * 8 + xcd;
    if (mi >= (w.cnt ? ldc(w.cnt) : nsys)) return;
    const int b = w.idx[mi];
    if (w.info[b] != 0) return;
    const int n = w.n;
    double* __restrict__ A = w.mats + (long)b * w.mstride;
    const int* __restrict__ live = w.live + (long)b * n;
    const int nblk = (n + 63) >> 6;

    const int nextp = NEXTP ? NB : 0;               // rows of the next super-panel, handled first
    const int mlive = n - k0 - NB - nextp;          // rows of the live list (NEXTP: the list after the next panel's compaction)
    const int mrem = mlive + nextp;                 // rows this launch updates
    const int cb0 = k0 + NB + cbi * 64;
    const int ncols = (n - cb0) < 64 ? (n - cb0) : 64;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int pl = 4 * (lane & 15) + (lane >> 4);  // LDS slot of column `lane`: columns q, q+16, q+32, q+48 sit together

    __shared__ __align__(16) double Us[NB][66];       // U12, columns permuted by pl; rows padded (16-byte aligned, 8 bank groups)
    __shared__ __align__(16) double Ls[KC][64];       // 4 wave-private [KC][16] strips of multipliers
    __shared__ unsigned short s_live[MAXROWS];

    {
        const double* __restrict__ U = p.u12 + ((long)b * nblk + cbi) * U12_BLOCK;
#pragma unroll
        for (int pass = 0; pass < NB / 4; ++pass) {
            const int k = pass * 4 + wave;
            Us[k][pl] = U[k * 64 + lane];
        }
        if (NEXTP) {
            const int* __restrict__ pnext = w.prow + (long)b * n + k0 + NB;  // pivot rows of the super-panel just factored
            if (t < NB) s_live[t] = (unsigned short)pnext[t];
        }
        for (int i = t; i < mlive; i += 256) s_live[nextp + i] = (unsigned short)live[i];
    }
    const int* __restrict__ fl = p.uflag + ((long)b * nblk + cbi) * U12_FLAGS;
    const bool slow = ldc(fl) != 0;
    const unsigned kmask0 = (unsigned)ldc(fl + 1), kmask1 = (unsigned)ldc(fl + 2);
    __syncthreads();
    {   // The solved pivot rows are final: pivot k of this super-panel is row k0 + k of the reference layout, so a column's 64
        // entries are one contiguous 512-byte store (one row per lane); lu_finalize_kernel skips this region.
        double* __restrict__ O = w.out + (long)b * w.ostride;
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const int cc = wave * 16 + i;  // column of the block; its LDS slot is 4 * (cc & 15) + (cc >> 4)
            if (cc < ncols) O[(long)(cb0 + cc) * n + k0 + lane] = Us[lane][4 * (cc & 15) + (cc >> 4)];
        }
    }
    if (mrem <= 0) return;
    if (slow && (kmask0 | kmask1) == 0u) {
        if (!NEXTP) return;  // U12 of this block is all zeros: nothing to subtract anywhere
        // (NEXTP: the next pivot rows still have to reach P, unchanged: the strips below copy them -- the select path with an
        // empty mask subtracts nothing)
    }

    // ---- this lane's share of a strip: rows a + 4i, columns q + 16j
    const int a = lane & 3, q = lane >> 2;
    // (NEXTP with an all-zero U12: only the four strips of the next pivot rows, which are copied to P)
    const int nstrips = (NEXTP && slow && (kmask0 | kmask1) == 0u) ? 4 : (mrem + 15) >> 4;
    int coff[4];
    bool cok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int cj = q + 16 * j;
        cok[j] = cj < ncols;
        coff[j] = (cb0 + (cok[j] ? cj : 0)) * n;
    }
    constexpr int LPT = KC / 4;  // multipliers per lane per k-chunk: lane (kq = lane >> 4, row = lane & 15) loads k = 4i + kq
    const int kq = lane >> 4, lr16 = lane & 15;
    const int lslot = 4 * (lr16 & 3) + (lr16 >> 2);  // rows a, a+4, a+8, a+12 of the strip sit together
    double lreg[LPT], creg[4][4];
    int crow[4];
    bool rok[4];
    auto load_L = [&](int s, int h) {  // one k-chunk of the strip's multipliers
        const int lr = s * 16 + lr16;
        const int lrow = s_live[lr < mrem ? lr : mrem - 1];
#pragma unroll
        for (int i = 0; i < LPT; ++i) lreg[i] = A[(k0 + h * KC + 4 * i + kq) * n + lrow];
    };
    auto load_C = [&](int s) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ri = s * 16 + a + 4 * i;
            rok[i] = ri < mrem;
            crow[i] = s_live[rok[i] ? ri : mrem - 1];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) creg[i][j] = A[coff[j] + crow[i]];
    };
    double (*__restrict__ Lw)[16] = reinterpret_cast<double (*)[16]>(&Ls[0][0] + wave * (KC * 16));

    auto chunk = [&](double (&c)[4][4], const int kbase) {
        if (!slow) {
            // software-pipelined by hand: the operands of step k + 1 are requested from LDS before the arithmetic of step k
            double lvA[4], uvA[4], lvB[4], uvB[4];
            auto rd = [&](const int k, double (&lv)[4], double (&uv)[4]) {
#pragma unroll
                for (int i = 0; i < 4; ++i) lv[i] = Lw[k][4 * a + i];
#pragma unroll
                for (int j = 0; j < 4; ++j) uv[j] = Us[kbase + k][4 * q + j];
            };
            auto mac = [&](const double (&lv)[4], const double (&uv)[4]) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) c[i][j] = upd<FMA>(c[i][j], uv[j], lv[i]);  // dense.rs:151
            };
            rd(0, lvA, uvA);
#pragma unroll
            for (int k = 0; k < KC; k += 2) {
                rd(k + 1, lvB, uvB);
                __builtin_amdgcn_sched_barrier(0);
                mac(lvA, uvA);
                if (k + 2 < KC) rd(k + 2, lvA, uvA);
                __builtin_amdgcn_sched_barrier(0);
                mac(lvB, uvB);
            }
        } else {
            for (unsigned mk = kbase == 0 ? kmask0 : kmask1; mk != 0u; mk &= mk - 1u) {  // ascending k, all-zero pivot rows skipped
                const int k = __builtin_ctz(mk);
                double lv[4], uv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) lv[i] = Lw[k][4 * a + i];
#pragma unroll
                for (int j = 0; j < 4; ++j) uv[j] = Us[kbase + k][4 * q + j];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const double tn = upd<FMA>(c[i][j], uv[j], lv[i]);
                        c[i][j] = (uv[j] != 0.0) ? tn : c[i][j];  // dense.rs:148
                    }
            }
        }
    };

    if (wave < nstrips) {
        load_C(wave);
        load_L(wave, 0);
    }
    double* __restrict__ Pb = p.pbuf + (long)b * 64 * n + cb0;
#pragma unroll 1
    for (int s = wave; s < nstrips; s += 4) {
#pragma unroll
        for (int i = 0; i < LPT; ++i) Lw[4 * i + kq][lslot] = lreg[i];
        load_L(s, 1);  // the strip's second k-chunk, in flight behind the first chunk's arithmetic
        double c[4][4];
        int srow[4];
        bool sok[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            srow[i] = crow[i];
            sok[i] = rok[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) c[i][j] = creg[i][j];
        }
        chunk(c, 0);
#pragma unroll
        for (int i = 0; i < LPT; ++i) Lw[4 * i + kq][lslot] = lreg[i];  // same wave, program order: chunk 0's reads are done
        if (s + 4 < nstrips) {  // next strip in flight behind the second chunk
            load_C(s + 4);
            load_L(s + 4, 0);
        }
        chunk(c, KC);
        if (NEXTP && s < 4) {
            // rows of the next super-panel: pivot index 16 s + a + 4 i; their updated entries are the next step's P
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (cok[j]) Pb[(long)(s * 16 + a + 4 * i) * n + q + 16 * j] = c[i][j];
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (cok[j] && sok[i]) A[coff[j] + srow[i]] = c[i][j];
        }
    }
}

}  // namespace idahip
