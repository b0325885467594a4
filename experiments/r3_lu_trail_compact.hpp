// Trailing update of the batched LU, 64-wide super-panels, COMPACT code (dense_get_rf's update,
// /root/reference/crates/linear/src/dense.rs:142-154, for everything right of a super-panel).
//
// Same algorithm, data layout and result as lu_trail64w_kernel (lu_kernels.hpp): one workgroup per (matrix, 64 trailing
// columns): gather of the 64 pivot rows, U12 = L11^-1 A12 in three stages, U12 stored straight into the factors, then the
// rank-64 update of the live rows in wave-private 16-row strips with a 4 x 4 register tile per lane.
//
// What is different is the size of the code. lu_trail64w_kernel unrolls everything -- both 32-step triangular solves (each
// twice: with and without the zero test of dense.rs:148), the middle stage, both 32-pivot chunks of the update -- into 108 KB of
// instructions; the instruction cache of a compute-unit pair holds 64 KB. Time stamps taken inside that kernel (tools/stamps.py)
// show what this costs: the three-stage solve -- 64 dependent steps, about 4000 instructions on ONE wave while the
// workgroup's other three wait -- takes 49 us of a workgroup's 161 us (33 us when that wave outranks all others in the issue
// arbiter, s_setprio: it is not waiting for issue slots, it is waiting for its instructions), and the update waves of the
// neighbouring workgroups lose their loop to the same evictions. Here
//   * a triangular stage is a ROLLED loop over its 32 pivots. The register file has no dynamic indexing, so the lane's 32
//     entries rotate by one place per step (the update writes u[j] = u[j+1] - ukk * l, the same trick as the panel kernels'
//     column rotation): every step runs the same ~70 instructions, on 31 candidates for the first 16 pivots and on 15 for
//     the last 16 (about a third of the multiplications work on slots past the end, whose values are never stored);
//   * both stages share one copy of that code (the stage index is a loop variable), the middle stage is unrolled 4 times
//     instead of 32, the update runs its two 32-pivot chunks through one copy of a loop unrolled 8 times.
// About 15 KB in all: the whole kernel stays resident in the instruction cache.
// Arithmetic per element is unchanged -- a(i,j) -= a_kj * a_ik in ascending k, unfused, dense.rs:148 honoured by the same
// verify-then-select scheme -- so results are bit-identical to lu_trail64w_kernel (tests/test_gpu_lsolver.py).
#pragma once
#include "lu_kernels.hpp"

namespace idahip {

template <int MAXROWS, bool FMA = false>
__global__ __launch_bounds__(256, MAXROWS <= 1024 ? 3 : 2) void lu_trail64c_kernel(LuWs w, int k0, int nsys, int ncb) {
    constexpr int NB = 64, KC = 32;
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
    const int pl = 4 * (lane & 15) + (lane >> 4);  // LDS slot of column `lane`: columns q, q+16, q+32, q+48 sit together

    constexpr int ULD = 66;  // row stride of Us (doubles): 16-byte aligned rows, column reads spread over 8 bank groups
    // U12, columns permuted by pl. One row more than the 64 pivots and 32 doubles more than the staging area: the software
    // pipelines of the loops below request the operands of one step past their last (values never used)
    __shared__ __align__(16) double Us[NB + 1][ULD];
    __shared__ __align__(16) double Ls[KC * 64 + 32];  // prologue: 32 rows of L11; update loop: 4 wave-private [KC][16] strips
    __shared__ unsigned short s_live[MAXROWS];
    __shared__ int s_anyzero;
    __shared__ int s_nz[4];          // per wave: a non-zero entry among the pivot-row entries it gathered
    __shared__ unsigned s_kmask[2];  // bit k of word R0 / 32 set = pivot row R0 + k has a non-zero entry in this column block

    // the prologue is a serial chain that three of the workgroup's four waves wait for: it outranks the update arithmetic of the
    // other workgroups on this compute unit in the issue arbiter
#ifdef IDAHIP_STAMPS
    unsigned long long* st = (k0 == 0 && w.stamps) ? w.stamps + (size_t)blockIdx.x * 8 : nullptr;
#define STAMP(i, wv) do { if (st && lane == 0 && wave == (wv)) st[i] = wall_clock64(); } while (0)
    if (st && t == 0) st[6] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | __builtin_amdgcn_s_getreg(63492);
#else
#define STAMP(i, wv) do { } while (0)
#endif
    STAMP(0, 0);
    __builtin_amdgcn_s_setprio(IDAHIP_TRAIL_PRIO);
    for (int i = t; i < mrem; i += 256) s_live[i] = (unsigned short)live[i];
    // ---- 1. gather the 64 pivot rows of this column block (wave-uniform k per pass: prow[k] is a scalar load)
    bool nz = false;
#pragma unroll 4
    for (int pass = 0; pass < NB / 4; ++pass) {
        const int k = pass * 4 + wave;
        const int pr = ldc(prow + k);
        const double g = (lane < ncols) ? A[(long)(cb0 + lane) * n + pr] : 0.0;
        nz = nz || (g != 0.0);
        Us[k][pl] = g;
    }
    if (lane == 0) s_nz[wave] = 0;
    if (__ballot(nz) != 0ull && lane == 0) s_nz[wave] = 1;
    if (t == 0) s_anyzero = 0;
    auto stage_l11 = [&](const int R0) {  // Ls[kk * 64 + c] = multiplier of pivot row c for column R0 + kk
#pragma unroll
        for (int i = 0; i < (KC * 64) / 256; ++i) {
            const int e = i * 256 + t;
            Ls[e] = l11[R0 * NB + e];
        }
    };
    stage_l11(0);
    lds_barrier();
    STAMP(1, 0);
    double* __restrict__ O = w.out + (long)b * w.ostride;
    auto store_factors = [&]() {
        // The solved pivot rows are final and nothing reads them in the work matrix again: pivot k of this super-panel is row
        // k0 + k of the reference layout, so a column's 64 entries are one contiguous 512-byte store (one row per lane);
        // lu_finalize_kernel skips this region.
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const int cc = wave * 16 + i;  // column of the block; its LDS slot is 4 * (cc & 15) + (cc >> 4)
            if (cc < ncols) O[(long)(cb0 + cc) * n + k0 + lane] = Us[lane][4 * (cc & 15) + (cc >> 4)];
        }
    };
    if ((s_nz[0] | s_nz[1] | s_nz[2] | s_nz[3]) == 0) {
        // the 64 pivot rows are zero across this whole column block (banded matrices, off the band): the triangular solve
        // leaves them as they are (a_kj == 0: column untouched, dense.rs:148) and nothing is subtracted from the rows below
        store_factors();
        return;
    }

    // ---- 2. U12 = L11^-1 A12, one column per lane: wave 0 solves rows 0..31, all four waves apply those rows to rows
    //         32..63 (8 rows per wave), wave 0 solves rows 32..63. Every element receives its updates in ascending pivot order.
    const bool real = lane < ncols;
    // one step of a triangular stage on the rotating register window: u[0] is the entry of pivot row R0 + kk (final), u[j] the
    // entry of row R0 + kk + j; W = candidates that may still exist
    // (the first NPF multipliers of a step are requested one step ahead: the next pivot's entry u[0] is the chain every later step
    // waits for, and it must not wait for an LDS round trip at the head of each step)
    constexpr int NPF = 8;
    auto tri_step = [&](double (&u)[KC], double (&lpf)[NPF], auto wtag, const int R0, const int kk, bool& anyz, unsigned& km) {
        constexpr int W = decltype(wtag)::value;
        const double ukk = u[0];
        Us[R0 + kk][pl] = ukk;
        const bool z = real && (ukk == 0.0);
        anyz = anyz || z;
        // dense.rs:148 skips the whole row update when a_kj == 0: a pivot row that is zero across this column block contributes
        // nothing to it (banded Jacobians). The mask is only read on the select path of the update.
        km |= (__ballot(real && ukk != 0.0) != 0ull) ? (1u << kk) : 0u;
        const double* __restrict__ lrow = &Ls[kk * 64 + R0 + kk + 1];  // multipliers of rows R0 + kk + 1 + j for this column
        // groups of 8 multipliers: a group is requested while the previous one is multiplied (all 31 at once would not fit the
        // register budget next to the 32 entries of the window)
        double lg[2][NPF];
#pragma unroll
        for (int j = 0; j < NPF; ++j) lg[0][j] = lpf[j];
        const bool anyzw = __ballot(z) != 0ull;
#pragma unroll
        for (int g = 0; g < W; g += NPF) {
            const int cur = (g / NPF) & 1;
#pragma unroll
            for (int j = 0; j < NPF; ++j)
                if (g + NPF + j < W) lg[cur ^ 1][j] = lrow[g + NPF + j];
            if (g == 0) {
#pragma unroll
                for (int j = 0; j < NPF; ++j) lpf[j] = lrow[(kk < KC - 1 ? 65 : 0) + j];  // next step: row kk + 1 of the staging area, one column further
            }
            if (!anyzw) {
#pragma unroll
                for (int j = 0; j < NPF; ++j)
                    if (g + j < W) u[g + j] = upd<FMA>(u[g + j + 1], ukk, lg[cur][j]);  // a(i,j) -= a_kj * a_ik, ascending kk
            } else {
#pragma unroll
                for (int j = 0; j < NPF; ++j)
                    if (g + j < W) {
                        const double tn = upd<FMA>(u[g + j + 1], ukk, lg[cur][j]);
                        u[g + j] = z ? u[g + j + 1] : tn;  // dense.rs:148: a_kj == 0 -> column untouched
                    }
            }
        }
    };
#pragma unroll 1
    for (int stage = 0; stage < 2; ++stage) {
        const int R0 = stage * KC;
        if (wave == 0) {
            double u[KC];
#pragma unroll
            for (int k = 0; k < KC; ++k) u[k] = Us[R0 + k][pl];
            bool anyz = false;
            unsigned km = 0u;
            double lpf[NPF];
#pragma unroll
            for (int j = 0; j < NPF; ++j) lpf[j] = Ls[R0 + 1 + j];
#pragma unroll 1
            for (int kk = 0; kk < 16; ++kk) tri_step(u, lpf, std::integral_constant<int, 31>{}, R0, kk, anyz, km);
#pragma unroll 1
            for (int kk = 16; kk < KC; ++kk) tri_step(u, lpf, std::integral_constant<int, 15>{}, R0, kk, anyz, km);
            if (lane == 0) {
                s_kmask[stage] = km;
                if (__ballot(anyz) != 0ull) s_anyzero = 1;
            }
        }
        lds_barrier();
        if (stage == 0) {
            // rows 32..63 receive the updates of pivot rows 0..31: 8 rows per wave, one column per lane
            const bool zpath = s_anyzero != 0;
            double v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = Us[KC + wave * 8 + i][pl];
            const double* __restrict__ lcol = &Ls[KC + wave * 8];
            if (!zpath) {
#pragma unroll 4
                for (int kk = 0; kk < KC; ++kk) {
                    const double ut = Us[kk][pl];
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = upd<FMA>(v[i], ut, lcol[kk * 64 + i]);
                }
            } else {
#pragma unroll 2
                for (int kk = 0; kk < KC; ++kk) {
                    const double ut = Us[kk][pl];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const double tn = upd<FMA>(v[i], ut, lcol[kk * 64 + i]);
                        v[i] = (ut != 0.0) ? tn : v[i];
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) Us[KC + wave * 8 + i][pl] = v[i];
            lds_barrier();
            stage_l11(KC);
            lds_barrier();
        }
    }
    // last workgroup barrier passed: from here on a wave touches only Us (read-only) and its own strip of Ls
    __builtin_amdgcn_s_setprio(0);
    STAMP(2, 0);
    const bool slow = s_anyzero != 0;
    store_factors();
    STAMP(3, 0);
    const unsigned kmask0 = (unsigned)__builtin_amdgcn_readfirstlane((int)s_kmask[0]);
    const unsigned kmask1 = (unsigned)__builtin_amdgcn_readfirstlane((int)s_kmask[1]);
    if (slow && (kmask0 | kmask1) == 0u) return;  // (uniform over the workgroup) U12 of this block is all zeros: nothing to subtract

    // ---- 3. rank-64 update in wave-private strips: this lane's share of a strip is rows a + 4i, columns q + 16j
    const int a = lane & 3, q = lane >> 2;
    const int nstrips = (mrem + 15) >> 4;
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
    double* __restrict__ Lw = &Ls[wave * (KC * 16)];  // [KC][16]

    // 32 pivots of the strip: operands of step k + 1 requested from LDS before the arithmetic of step k is issued
    auto chunk = [&](double (&c)[4][4], const int kbase) {
        if (!slow) {
            double lvA[4], uvA[4], lvB[4], uvB[4];
            auto rd = [&](const int k, double (&lv)[4], double (&uv)[4]) {
#pragma unroll
                for (int i = 0; i < 4; ++i) lv[i] = Lw[k * 16 + 4 * a + i];
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
#pragma unroll 1
            for (int k8 = 0; k8 < KC; k8 += 8) {
#pragma unroll
                for (int k = 0; k < 8; k += 2) {
                    rd(k8 + k + 1, lvB, uvB);
                    __builtin_amdgcn_sched_barrier(0);
                    mac(lvA, uvA);
                    rd(k8 + k + 2, lvA, uvA);  // (one step past the chunk's end at the very last: within the padding, never used)
                    __builtin_amdgcn_sched_barrier(0);
                    mac(lvB, uvB);
                }
            }
        } else {
            for (unsigned mk = kbase == 0 ? kmask0 : kmask1; mk != 0u; mk &= mk - 1u) {  // ascending k, all-zero pivot rows skipped
                const int k = __builtin_ctz(mk);
                double lv[4], uv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) lv[i] = Lw[k * 16 + 4 * a + i];
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
#pragma unroll 1
    for (int s = wave; s < nstrips; s += 4) {
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
#pragma unroll 1
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int i = 0; i < LPT; ++i) Lw[(4 * i + kq) * 16 + lslot] = lreg[i];  // same wave, program order: the previous chunk's reads are done
            if (h == 0) {
                load_L(s, 1);  // the strip's second k-chunk, in flight behind the first chunk's arithmetic
            } else if (s + 4 < nstrips) {  // next strip in flight behind the second chunk
                load_C(s + 4);
                load_L(s + 4, 0);
            }
            chunk(c, h * KC);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (cok[j] && sok[i]) A[coff[j] + srow[i]] = c[i][j];
    }
    STAMP(4, 0);
    STAMP(5, 3);
#undef STAMP
}

}  // namespace idahip
