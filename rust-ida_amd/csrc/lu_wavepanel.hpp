// Super-panel factorisation by ONE wavefront per matrix (dense_get_rf, /root/reference/crates/linear/src/dense.rs:86-158,
// restricted to a super-panel of 64 columns; the columns right of it are updated by lu_trail64w_kernel).
//
// Why one wave: a pivot step (dense.rs:107-155) is a dependent chain -- arg-max over the column, pivot row, reciprocal,
// multipliers, update of the next column. Spread over several waves every step pays an LDS hand-off and a workgroup
// barrier, and every wave repeats the chain's bookkeeping (lu_panel2_kernel: ~450 instructions per step on each of four
// waves, 3 us per matrix at N = 512). Here all live rows of the panel sit in the registers of one wave -- lane l holds rows
// l, l + 64, ... of the live list ("slots", <= 8 of them: <= 512 live rows) -- so the arg-max is one pass on the DPP
// crossbar, the pivot row is broadcast with v_readlane into SGPRs (scalar operands of the update, no LDS), and there is
// no barrier and no redundant work. Thousands of such waves run side by side (one per matrix of the batch), two per SIMD.
//
// Blocking: the super-panel is factored left-looking in blocks of 8 columns, because 8 slots x 8 columns x 8 bytes is
// what a lane can hold (128 VGPRs). For block b the wave loads its 8 columns, applies the updates of the super-panel's
// earlier pivots k = 0 .. 8b-1 in ascending k (multipliers of column k re-read from the work matrix in coalesced column
// segments, U entries broadcast from the lane that holds pivot row k), then runs the block's own 8 pivot steps, and
// stores the block: multipliers and U entries are in place. Every element receives a(i,j) -= a_kj * a_ik for the same k
// in the same ascending order as the reference's column sweep, unfused (-ffp-contract=off), multipliers a_ik * (1/a_kk);
// a row never moves (implicit pivoting as in lu_kernels.hpp: `rpos` is the row's position in the reference's matrix, ties
// in |a| go to the lowest position, dense.rs:113).
//
// Two modes per block. FAST assumes what holds for all but sparse or broken matrices: no NaN or infinity among the pivot
// candidates and no exact zero among the U entries the block multiplies with (dense.rs:148 leaves such a column
// untouched, which differs from subtracting 0 * l in the sign of a zero and when l is not finite). It verifies both after
// the fact, before anything is stored; if one fails the block is loaded again and redone in SLOW mode (NaN-aware keys,
// zero test per column). The two modes never share live registers: each ends in its own store.
#pragma once
#include <type_traits>

#include "lu_kernels.hpp"

namespace idahip {

constexpr int WP_MAX_ROWS = 512;
// depth (in pivots) of the multiplier prefetch ring of lu_wavepanel_kernel for a given slot count
// (measured, 1240 matrices per call: rings of 4 and 8 change nothing for 1-5 slots and cost 4-8 % for 6-7 -- the loop waits
// for instruction issue, not for the loads; -DIDAHIP_TIMING_BUILD -DIDAHIP_WP_RING=1 builds the deep rings for another look)
constexpr int WP_RING(int ns) { return !tb::WP_DEEP_RING ? 2 : (ns >= 1 && ns <= 2) ? 8 : (ns >= 3 && ns <= 7) ? 4 : 2; }  // 8 slots of 64 lanes

// NS > 0: the number of slots (= ceil(live rows / 64)) as a compile-time constant -- the per-slot guards fold away and the
// registers of the unused slots are never allocated (eight instantiations of the FAST kernel, one per super-panel of an
// N = 512 factorisation); NS = 0: taken from the matrix size at run time (the SLOW kernel).
template <bool SLOWK, int NS>
__global__ __launch_bounds__(64, SLOWK ? 1 : (NS > 0 && NS <= 3 ? 4 : 2)) void lu_wavepanel_kernel(LuWs w, const int k0) {
    constexpr int BIG = 1 << 20;  // pstep of a row that is still live
    if (w.cnt && (int)blockIdx.x >= *w.cnt) return;
    const int b = w.idx[blockIdx.x];
    if (w.info[b] != 0) return;
    const int n = w.n;
    double* __restrict__ A = w.mats + (long)b * w.mstride;
    int* __restrict__ pos = w.pos + (long)b * n;
    int* __restrict__ live = w.live + (long)b * n;
    int* __restrict__ prow = w.prow + (long)b * n;
    long long* __restrict__ piv = w.piv + (long)b * w.pstride;
    double* __restrict__ l11 = w.l11 + (long)b * L11_STRIDE;

    // buffer descriptor of this system's work matrix: loads with a 32-bit per-lane row offset + a scalar column offset, no
    // vector address arithmetic (the matrix is n * n * 8 bytes <= 2 MiB)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(A, 0, n * n * 8, 0x00020000);
    const int m = n - k0;                 // live rows, <= WP_MAX_ROWS
    const int wsp = m < 64 ? m : 64;      // columns of this super-panel
    const int lane = threadIdx.x;
    const int nslots = NS > 0 ? NS : (m + 63) >> 6;
    constexpr int NSC = NS > 0 ? NS : 8;  // slots that can hold rows: every loop over slots stops there

    __shared__ int ptab[64];  // owner of pivot k of this super-panel: lane | slot << 6

    int rowid[8], rpos[8], pstep[8];  // physical row, reference position, step at which the row became a pivot (BIG: live, -1: no row)
    unsigned roff[8];                 // byte offset of the row inside a column of the work matrix
    static_for<0, NSC>([&](auto st) {
        constexpr int S = decltype(st)::value;
        const int li = S * 64 + lane;
        const bool has = li < m;
        rowid[S] = has ? live[li] : 0;
        rpos[S] = has ? pos[rowid[S]] : 0x7fffffff;
        pstep[S] = has ? BIG : -1;
        roff[S] = (unsigned)rowid[S] * 8u;
    });

    // ------------------------------------------------------------------------------------------------------------------
    // one block of `WB` (= 8 unless PARTIAL) columns starting at column b8 of the super-panel; x = the block, in registers.
    // returns 0 = done, 1 = zero pivot (info set), 2 = FAST assumptions violated (nothing stored)
    auto load_block = [&](double (&x)[8][8], const int b8, const int wb) {
        static_for<0, NSC>([&](auto st) {
            constexpr int S = decltype(st)::value;
            if (S < nslots) {
                const double* __restrict__ src = A + (long)(k0 + b8) * n + rowid[S];
#pragma unroll
                for (int j = 0; j < 8; ++j) x[S][j] = (j < wb) ? src[(long)j * n] : 0.0;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) x[S][j] = 0.0;
            }
        });
    };
    auto store_block = [&](const double (&x)[8][8], const int b8, const int wb) {
        static_for<0, NSC>([&](auto st) {
            constexpr int S = decltype(st)::value;
            if (S < nslots && pstep[S] >= 0) {
                double* __restrict__ dst = A + (long)(k0 + b8) * n + rowid[S];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (j < wb) dst[(long)j * n] = x[S][j];
            }
        });
    };

    auto process = [&](auto slow_tag, double (&x)[8][8], const int b8, const int wb) -> int {
        constexpr bool SLOW = decltype(slow_tag)::value;
        // ---- left-looking: the updates of the super-panel's earlier pivots, ascending k
        if (b8 > 0) {
            // multipliers of column k0 + k for the rows of this lane: scalar column base + 32-bit row offset (no vector
            // address arithmetic per load); two register sets used alternately, each loaded one step ahead of its use
            auto load_l = [&](double (&l)[8], const int k) {
                const int coloff = (k0 + k) * n * 8;  // byte offset of the column inside this system's matrix (< 2^31: n <= 512)
                static_for<0, NSC>([&](auto st) {
                    constexpr int S = decltype(st)::value;
                    // (unconditional: a slot without rows has offset 0 and reads row 0 -- one cached line for the whole wave, the
                    // value is never used. A guarded load would leave the register's old value live on one path, and the compiler
                    // then waits for every outstanding load before issuing the next one.)
                    l[S] = buf_load_f64(rsrc, roff[S], coloff);
                });
            };
            auto apply = [&](const double (&l)[8], const int k, const int ow) {
                const int pl = __builtin_amdgcn_readfirstlane(ow) & 63, ps = __builtin_amdgcn_readfirstlane(ow) >> 6;
                double uk[8];  // U entries of pivot row k in this block's columns, as they stand after the updates 0 .. k-1
                switch (ps) {
#define IDAHIP_WP_CASE(SV)                                                                     \
    case SV:                                                                                   \
        if constexpr (SV < NSC) {                                                              \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) uk[j] = readlane_f64(x[SV][j], pl); \
        }                                                                                      \
        break;
                    IDAHIP_WP_CASE(1)
                    IDAHIP_WP_CASE(2)
                    IDAHIP_WP_CASE(3)
                    IDAHIP_WP_CASE(4)
                    IDAHIP_WP_CASE(5)
                    IDAHIP_WP_CASE(6)
                    IDAHIP_WP_CASE(7)
                    default:
                        IDAHIP_WP_CASE(0)
#undef IDAHIP_WP_CASE
                }
                double ukv[8];
                if (SLOW) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) ukv[j] = opaque_vgpr(uk[j]);
                }
                static_for<0, NSC>([&](auto st) {
                    constexpr int S = decltype(st)::value;
                    if (S < nslots) {
                        if (pstep[S] > k) {  // the row was still live at step k (it is live now, or became a pivot later)
#pragma unroll
                            for (int j = 0; j < 8; ++j) {
                                if (SLOW) {
                                    const double tn = upd(x[S][j], ukv[j], l[S]);
                                    x[S][j] = (ukv[j] == 0.0) ? x[S][j] : tn;  // dense.rs:148: a_kj == 0 -> column untouched
                                } else {
                                    x[S][j] = upd(x[S][j], uk[j], l[S]);
                                }
                            }
                        }
                    }
                });
            };
            // the block's loads are complete before the loop starts: the compiler's wait-count bookkeeping would otherwise carry
            // "x may still be in flight" around the loop and wait for the freshly issued multiplier loads at every use of x
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
            // D register sets used in turn, each reloaded right after its use: D - 1 pivots' worth of arithmetic covers a load.
            // A step is 16 * slots fp64 operations, a load from HBM about a microsecond: few slots want a deep ring, and have the
            // registers for it. (A reload past the end of the range re-reads the last column; the value is never used.)
            constexpr int D = WP_RING(NS);
            double l[D][8];
            int ow[D];
            static_for<0, D>([&](auto dt) {
                constexpr int d = decltype(dt)::value;
                load_l(l[d], d);
                ow[d] = ptab[d];
            });
#pragma unroll 1
            for (int k = 0; k < b8; k += D) {  // b8 is a multiple of 8, D divides 8
                static_for<0, D>([&](auto dt) {
                    constexpr int d = decltype(dt)::value;
                    apply(l[d], k + d, ow[d]);
                    const int kn = (k + d + D < b8) ? k + d + D : b8 - 1;
                    load_l(l[d], kn);
                    ow[d] = ptab[kn];
                });
            }
        }
        // ---- the block's own pivot steps
        bool special = false;  // FAST: a NaN or an infinity was among the candidates of some step
        int status = 0, failcol = 0;
        static_for<0, 8>([&](auto kt) {
            constexpr int kk = decltype(kt)::value;
            // SLOW: a step beyond the block's width, or after a zero pivot, runs with `active` = false and changes nothing
            // (FAST sees full blocks only, and leaves a zero pivot to the SLOW rerun)
            const bool active = !SLOW || (status == 0 && kk < wb);
            const int kstep = b8 + kk, kc = k0 + kstep;
            // candidate key of a live row: the bit pattern of |a| with the always-clear sign bit set (0 = no candidate);
            // NaN only wins at position kc (dense.rs:111-117 scan semantics)
            unsigned kh[8], kl[8];
            static_for<0, NSC>([&](auto st) {
                constexpr int S = decltype(st)::value;
                const double v = x[S][kk];
                unsigned h = (unsigned)__double2hiint(v) | 0x80000000u, l = (unsigned)__double2loint(v);
                if (SLOW) {
                    if (v != v) {
                        h = (rpos[S] == kc) ? 0xfff00000u : 0u;
                        l = 0u;
                    }
                }
                kh[S] = (pstep[S] == BIG) ? h : 0u;
                kl[S] = l;
            });
            unsigned lm = kh[0];
#pragma unroll
            for (int s = 1; s < NSC; ++s) lm = kh[s] > lm ? kh[s] : lm;
            const unsigned mh = wave_max_u32<false>(lm);
            if (!SLOW) special = special || mh >= 0xfff00000u;
            // is the maximum of the high words attained by one row only? (the common case: the arg-max is found)
            int cnt = 0, slot = 0;
#pragma unroll
            for (int s = 0; s < NSC; ++s) {
                const bool e = kh[s] == mh;
                cnt += e ? 1 : 0;
                slot = e ? s : slot;
            }
            const unsigned long long bal = __ballot(cnt > 0);
            int pl = (int)__ffsll((unsigned long long)bal) - 1;
            int ps;
            if (__popcll(bal) == 1 && __builtin_amdgcn_readlane(cnt, pl) == 1) {
                ps = __builtin_amdgcn_readlane(slot, pl);
            } else {  // several rows share the high word: low word, then lowest position (dense.rs:113)
                unsigned bl = 0u;
                int bp_ = 0x7fffffff, bs = 0;
#pragma unroll
                for (int s = 0; s < NSC; ++s) {
                    const bool e = kh[s] == mh;
                    const bool better = e && (kl[s] > bl || (kl[s] == bl && rpos[s] < bp_) || bp_ == 0x7fffffff);
                    bl = better ? kl[s] : bl;
                    bp_ = better ? rpos[s] : bp_;
                    bs = better ? s : bs;
                }
                const bool e1 = bp_ != 0x7fffffff;
                const unsigned ml = wave_max_u32<false>(e1 ? bl : 0u);
                const bool top = e1 && bl == ml;
                const int pm = wave_min_i32f<false>(top ? bp_ : 0x7fffffff);
                const unsigned long long b2 = __ballot(top && bp_ == pm);
                pl = (int)__ffsll((unsigned long long)b2) - 1;
                ps = __builtin_amdgcn_readlane(bs, pl);
            }
            // pivot row: block entries, position, physical row -- broadcast from lane pl, slot ps; the owner marks its row
            double u[8];
            int bp = 0, pr = 0;
            switch (ps) {
#define IDAHIP_WP_CASE(SV)                                                                             \
    case SV:                                                                                           \
        if constexpr (SV < NSC) {                                                                      \
            _Pragma("unroll") for (int j = kk; j < 8; ++j) u[j] = readlane_f64(x[SV][j], pl);         \
            bp = __builtin_amdgcn_readlane(rpos[SV], pl);                                              \
            pr = __builtin_amdgcn_readlane(rowid[SV], pl);                                             \
        }                                                                                              \
        break;
                IDAHIP_WP_CASE(1)
                IDAHIP_WP_CASE(2)
                IDAHIP_WP_CASE(3)
                IDAHIP_WP_CASE(4)
                IDAHIP_WP_CASE(5)
                IDAHIP_WP_CASE(6)
                IDAHIP_WP_CASE(7)
                default:
                    IDAHIP_WP_CASE(0)
#undef IDAHIP_WP_CASE
            }
            bool go = active;
            if (active && u[kk] == 0.0) {  // zero pivot: Err(k+1)  (dense.rs:120-122)
                status = 1;
                failcol = kc + 1;
                go = !SLOW;
            }
            const double recip = 1.0 / u[kk];  // a(k,k).recip() (dense.rs:134)
            if (go) {
                if (lane == 0) {
                    piv[kc] = (long long)bp;
                    prow[kc] = pr;
                    ptab[kstep] = pl | (ps << 6);
                }
                // ptab is written by lane 0 and read by every lane in the left-looking loop of the later blocks: one wave, LDS
                // operations in program order -- the fence and the wave barrier only keep the compiler from reordering them
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                static_for<0, NSC>([&](auto st) {
                    constexpr int S = decltype(st)::value;
                    if (pstep[S] == BIG && rpos[S] == kc) rpos[S] = bp;  // the row that sat at position k moves to the pivot's old position
                    if (lane == pl && ps == S) {
                        pstep[S] = kstep;
                        rpos[S] = kc;
                    }
                });
            }
            double uv[8];
            if (SLOW) {
#pragma unroll
                for (int j = kk + 1; j < 8; ++j) uv[j] = opaque_vgpr(u[j]);
            }
            static_for<0, NSC>([&](auto st) {
                constexpr int S = decltype(st)::value;
                if (S < nslots) {
                    if (pstep[S] == BIG && go) {
                        const double l = x[S][kk] * recip;  // dense.rs:134-137
                        x[S][kk] = l;
#pragma unroll
                        for (int j = kk + 1; j < 8; ++j) {
                            if (SLOW) {
                                const double tn = upd(x[S][j], uv[j], l);
                                x[S][j] = (uv[j] == 0.0) ? x[S][j] : tn;  // dense.rs:148
                            } else {
                                x[S][j] = upd(x[S][j], u[j], l);
                            }
                        }
                    }
                }
            });
        });
        if (!SLOW) {
            // verify the assumptions: no special value among the candidates, no exact zero in a pivot row of this block's
            // columns (conservative: the multipliers of the block's own pivot rows are looked at too)
            bool z = false;
            static_for<0, NSC>([&](auto st) {
                constexpr int S = decltype(st)::value;
                bool zz = false;
#pragma unroll
                for (int j = 0; j < 8; ++j) zz = zz || (x[S][j] == 0.0 && j < wb);
                z = z || (zz && pstep[S] >= 0 && pstep[S] != BIG);
            });
            if (special || status != 0 || __ballot(z) != 0ull) return 2;
        }
        if (status == 1 && lane == 0) w.info[b] = failcol;
        return status;
    };

    int bstart = 0;
    if (SLOWK) {
        // this launch only finishes the systems the FAST launch handed over: redo = 1 + first block to be done in SLOW mode
        const int rd = w.redo[b];
        if (rd == 0) return;
        bstart = (rd - 1) * 8;
        // replay the pivots of the blocks the FAST launch completed (positions, pivot steps, owner table)
        for (int k = 0; k < bstart; ++k) {
            const int kc = k0 + k;
            const int bp = (int)piv[kc], pr = prow[kc];
            static_for<0, NSC>([&](auto st) {
                constexpr int S = decltype(st)::value;
                if (pstep[S] == BIG && rpos[S] == kc) rpos[S] = bp;
                const bool mine = pstep[S] >= 0 && rowid[S] == pr;
                if (mine) {
                    pstep[S] = k;
                    rpos[S] = kc;
                }
                const unsigned long long bm = __ballot(mine);
                if (bm != 0ull && lane == 0) ptab[k] = ((int)__ffsll((unsigned long long)bm) - 1) | (S << 6);
            });
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (lane == 0) w.redo[b] = 0;
    }
    for (int b8 = bstart; b8 < wsp; b8 += 8) {
        const int wb = (wsp - b8) < 8 ? (wsp - b8) : 8;
        double x[8][8];
        int st;
        if (SLOWK) {
            load_block(x, b8, wb);
            st = process(std::true_type{}, x, b8, wb);
        } else {
            if (wb != 8) {
                st = 2;
            } else {
                load_block(x, b8, 8);
                st = process(std::false_type{}, x, b8, 8);
            }
            if (st == 2) {  // hand the rest of this super-panel to the SLOW launch; nothing of this block has been stored
                if (lane == 0) w.redo[b] = 1 + (b8 >> 3);
                return;
            }
        }
        if (st != 0) return;
        store_block(x, b8, wb);
    }

    // ---- positions, transposed L11 (multipliers of the pivot rows, read back from the matrix), compacted live list
    __syncthreads();  // the stores above (work matrix, prow) are visible to every lane of the wave
    static_for<0, NSC>([&](auto st) {
        constexpr int S = decltype(st)::value;
        if (pstep[S] >= 0) pos[rowid[S]] = rpos[S];
    });
    if (lane < wsp) {  // lane = pivot index k: l11[kk * 64 + k] = multiplier of pivot row k for column kk < k
        const double* __restrict__ src = A + (long)k0 * n + prow[k0 + lane];
        for (int j0 = 0; j0 < lane; j0 += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = (j0 + u < lane) ? src[(long)(j0 + u) * n] : 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (j0 + u < lane) l11[(j0 + u) * 64 + lane] = v[u];
        }
    }
    int base = 0;
    static_for<0, NSC>([&](auto st) {
        constexpr int S = decltype(st)::value;
        const bool al = pstep[S] == BIG;
        const unsigned long long bal = __ballot(al);
        if (al) live[base + __popcll(bal & ((1ull << lane) - 1ull))] = rowid[S];
        base += __popcll(bal);
    });
}

}  // namespace idahip
