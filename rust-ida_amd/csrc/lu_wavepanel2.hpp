// Super-panel factorisation by TWO wavefronts per matrix (dense_get_rf, /root/reference/crates/linear/src/dense.rs:86-158,
// restricted to a super-panel of 64 columns) -- the FAST kernel of lu_wavepanel.hpp with a matrix's live rows dealt to two
// waves of one workgroup.
//
// Why: with one wave per matrix a config-3 round gives 1312 matrices to 1024 SIMDs; a wave that has its SIMD to itself issues
// an fp64 instruction every 8-9 cycles where the pipe takes one every 4, and the launch lasts as long as
// one wave needs for all eight slots. Two waves per matrix halve the rows a wave carries and put 2.5 waves on a SIMD.
//
// What the split costs, and how it is kept small:
//   * the left-looking updates of a block need, for every earlier pivot k of the super-panel, that pivot row's entries in
//     the block's columns AS UPDATED BY THE PIVOTS BEFORE IT. In the one-wave kernel the row sits in some lane's registers and
//     is read with v_readlane. Here either wave may own it -- so BOTH waves carry a copy: an extra register slot ("U slot")
//     in which lane k holds pivot row k of this super-panel (loaded with the block, prow[] says which physical row that is)
//     and which receives the same updates, from the same multipliers, as any other row. No exchange inside the left-looking
//     loop; the price is one slot's worth of redundant arithmetic per wave (5 slots of work instead of 4 at 512 live rows).
//   * the block's own 8 pivot steps need the arg-max over both waves: each wave finds its best candidate on the DPP crossbar
//     as before and puts key, position and the candidate row's 8 entries into LDS; one LDS barrier; both waves read both
//     candidates and take the same decision (|a|, then lowest position, dense.rs:113). One barrier per pivot step, 64 per
//     super-panel; two more per block (the verification verdict, and the stores before the next block's loads).
// Everything else -- implicit pivoting with reference positions, FAST assumptions verified before a block is stored, hand-over
// of the rest of a super-panel to the SLOW launch (lu_wavepanel_kernel<true, 0>, unchanged: it replays the pivots from piv /
// prow), transposed L11, compacted live list -- is lu_wavepanel.hpp's, and so is every bit of the result: an element receives
// a(i,j) -= a_kj * a_ik for the same k in the same ascending order, unfused, multipliers a_ik * (1 / a_kk).
#pragma once
#include "lu_wavepanel.hpp"

namespace idahip {

struct Wp2Cand {      // a wave's best pivot candidate of one step, as it hands it to the other wave
    unsigned kh, kl;  // key: bit pattern of |a| with the (always clear) sign bit set; kh == 0: the wave has no live row
    int rpos, rowid;  // the candidate's position in the reference's matrix, its physical row
    double u[8];      // its entries in the block's columns (those left of the step's column are not used)
};

__device__ __forceinline__ double uniform_f64(double v) {  // a wave-uniform value into scalar registers
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// NSG = ceil(live rows / 64), 2..8: global slot g (rows g * 64 + lane of the live list) belongs to wave g & 1, local slot g >> 1
template <int NSG>
__global__ __launch_bounds__(128) void lu_wavepanel2_kernel(LuWs w, const int k0) {
    static_assert(NSG >= 2 && NSG <= 8, "two waves share 65 .. 512 live rows");
    constexpr int BIG = 1 << 20;         // pstep of a row that is still live
    constexpr int NSL = (NSG + 1) / 2;   // local slots of a wave
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
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(A, 0, n * n * 8, 0x00020000);
    const int m = n - k0;             // live rows, 65 .. WP_MAX_ROWS
    const int wsp = m < 64 ? m : 64;  // columns of this super-panel (64: m > 64)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));

    __shared__ Wp2Cand s_c[2][2];  // [step parity][wave]
    __shared__ int s_flag[2];      // FAST verification: this wave saw a violated assumption in the block
    __shared__ int s_cnt[8];       // live rows left in global slot g (for the compacted live list)
    if (threadIdx.x < 8) s_cnt[threadIdx.x] = 0;

    int rowid[NSL], rpos[NSL], pstep[NSL];  // physical row, reference position, BIG = live / step at which it became a pivot / -1 = no row
    unsigned roff[NSL];
    static_for<0, NSL>([&](auto st) {
        constexpr int S = decltype(st)::value;
        const int li = (2 * S + wave) * 64 + lane;
        const bool has = li < m;
        rowid[S] = has ? live[li] : 0;
        rpos[S] = has ? pos[rowid[S]] : 0x7fffffff;
        pstep[S] = has ? BIG : -1;
        roff[S] = (unsigned)rowid[S] * 8u;
    });

    for (int b8 = 0; b8 < wsp; b8 += 8) {
        if (wsp - b8 < 8) {  // a partial last block (n not a multiple of 8): the SLOW launch takes it
            if (threadIdx.x == 0) w.redo[b] = 1 + (b8 >> 3);
            return;
        }
        __syncthreads();  // the previous block's stores (work matrix, prow) are visible to both waves
        // ---- load the block: own rows, and the super-panel's earlier pivot rows in the U slot (lane k = pivot k)
        const bool uhas = lane < b8;
        const int urow = uhas ? prow[k0 + lane] : 0;
        const unsigned uoff = (unsigned)urow * 8u;
        double x[NSL][8], xu[8];
        static_for<0, NSL>([&](auto st) {
            constexpr int S = decltype(st)::value;
            const double* __restrict__ src = A + (long)(k0 + b8) * n + rowid[S];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[S][j] = src[(long)j * n];
        });
        {
            const double* __restrict__ src = A + (long)(k0 + b8) * n + urow;
#pragma unroll
            for (int j = 0; j < 8; ++j) xu[j] = src[(long)j * n];
        }
        // ---- left-looking: the updates of the super-panel's earlier pivots, ascending k
        if (b8 > 0) {
            auto load_l = [&](double (&l)[NSL + 1], const int k) {
                const int coloff = (k0 + k) * n * 8;
                static_for<0, NSL>([&](auto st) {
                    constexpr int S = decltype(st)::value;
                    l[S] = buf_load_f64(rsrc, roff[S], coloff);  // (unconditional: see lu_wavepanel.hpp)
                });
                l[NSL] = buf_load_f64(rsrc, uoff, coloff);
            };
            auto apply = [&](const double (&l)[NSL + 1], const int k) {
                double uk[8];  // pivot row k in this block's columns, as it stands after the updates 0 .. k-1
#pragma unroll
                for (int j = 0; j < 8; ++j) uk[j] = readlane_f64(xu[j], k);
                if (lane > k && uhas) {  // the pivot rows after k
#pragma unroll
                    for (int j = 0; j < 8; ++j) xu[j] = upd(xu[j], uk[j], l[NSL]);
                }
                static_for<0, NSL>([&](auto st) {
                    constexpr int S = decltype(st)::value;
                    if (pstep[S] == BIG) {  // live rows (a row that became a pivot in an earlier block lives on in the U slot)
#pragma unroll
                        for (int j = 0; j < 8; ++j) x[S][j] = upd(x[S][j], uk[j], l[S]);
                    }
                });
            };
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the block is in its registers before the loop starts (lu_wavepanel.hpp)
            double l[2][NSL + 1];
            load_l(l[0], 0);
            load_l(l[1], 1);
#pragma unroll 1
            for (int k = 0; k < b8; k += 2) {  // b8 is a multiple of 8
                apply(l[0], k);
                load_l(l[0], (k + 2 < b8) ? k + 2 : b8 - 1);
                apply(l[1], k + 1);
                load_l(l[1], (k + 3 < b8) ? k + 3 : b8 - 1);
            }
        }
        // ---- the block's own pivot steps
        bool special = false;
        int status = 0;
        static_for<0, 8>([&](auto kt) {
            constexpr int kk = decltype(kt)::value;
            const int kstep = b8 + kk, kc = k0 + kstep;
            unsigned kh[NSL], kl[NSL];
            static_for<0, NSL>([&](auto st) {
                constexpr int S = decltype(st)::value;
                const double v = x[S][kk];
                kh[S] = (pstep[S] == BIG) ? ((unsigned)__double2hiint(v) | 0x80000000u) : 0u;
                kl[S] = (unsigned)__double2loint(v);
            });
            unsigned lm = kh[0];
#pragma unroll
            for (int s = 1; s < NSL; ++s) lm = kh[s] > lm ? kh[s] : lm;
            const unsigned mh = wave_max_u32<false>(lm);
            special = special || mh >= 0xfff00000u;
            int pl = 0, ps = 0;
            Wp2Cand* __restrict__ mine = &s_c[kk & 1][wave];
            if (mh != 0u) {  // (uniform) this wave has a live row
                int cnt = 0, slot = 0;
#pragma unroll
                for (int s = 0; s < NSL; ++s) {
                    const bool e = kh[s] == mh;
                    cnt += e ? 1 : 0;
                    slot = e ? s : slot;
                }
                const unsigned long long bal = __ballot(cnt > 0);
                pl = (int)__ffsll((unsigned long long)bal) - 1;
                if (__popcll(bal) == 1 && __builtin_amdgcn_readlane(cnt, pl) == 1) {
                    ps = __builtin_amdgcn_readlane(slot, pl);
                } else {  // several rows share the high word: low word, then lowest position (dense.rs:113)
                    unsigned bl = 0u;
                    int bp_ = 0x7fffffff, bs = 0;
#pragma unroll
                    for (int s = 0; s < NSL; ++s) {
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
                // the lane that holds the candidate puts it into LDS itself (no broadcast through scalar registers needed here)
                static_for<0, NSL>([&](auto st) {
                    constexpr int S = decltype(st)::value;
                    if (lane == pl && ps == S) {
                        mine->kh = kh[S];
                        mine->kl = kl[S];
                        mine->rpos = rpos[S];
                        mine->rowid = rowid[S];
#pragma unroll
                        for (int j = kk; j < 8; ++j) mine->u[j] = x[S][j];
                    }
                });
            } else if (lane == 0) {
                mine->kh = 0u;
                mine->kl = 0u;
                mine->rpos = 0x7fffffff;
                mine->rowid = 0;
            }
            lds_barrier();
            // both waves take the same decision: larger |a| (high word, then low word), then the lower position
            const Wp2Cand* __restrict__ c01 = &s_c[kk & 1][0];
            const unsigned h0 = c01[0].kh, h1 = c01[1].kh, l0 = c01[0].kl, l1 = c01[1].kl;
            const int q0 = c01[0].rpos, q1 = c01[1].rpos;
            const int w1 = __builtin_amdgcn_readfirstlane((h1 > h0 || (h1 == h0 && (l1 > l0 || (l1 == l0 && q1 < q0)))) ? 1 : 0);
            const Wp2Cand* __restrict__ cw = c01 + w1;  // only the winner's row is read
            double u[8];
#pragma unroll
            for (int j = kk; j < 8; ++j) u[j] = uniform_f64(cw->u[j]);
            const int bp = __builtin_amdgcn_readfirstlane(cw->rpos), pr = __builtin_amdgcn_readfirstlane(cw->rowid);
            if (u[kk] == 0.0) status = 1;  // zero pivot (dense.rs:120-122): the SLOW rerun reports it
            const double recip = 1.0 / u[kk];
            if (threadIdx.x == 0) {
                piv[kc] = (long long)bp;
                prow[kc] = pr;
            }
            const bool iwon = w1 == wave;
            static_for<0, NSL>([&](auto st) {
                constexpr int S = decltype(st)::value;
                if (pstep[S] == BIG && rpos[S] == kc) rpos[S] = bp;  // the row that sat at position k moves to the pivot's old position
                if (iwon && lane == pl && ps == S) {
                    pstep[S] = kstep;
                    rpos[S] = kc;
                }
            });
            static_for<0, NSL>([&](auto st) {
                constexpr int S = decltype(st)::value;
                if (pstep[S] == BIG) {
                    const double l = x[S][kk] * recip;  // dense.rs:134-137
                    x[S][kk] = l;
#pragma unroll
                    for (int j = kk + 1; j < 8; ++j) x[S][j] = upd(x[S][j], u[j], l);
                }
            });
        });
        // ---- verify the FAST assumptions: no special value among the candidates, no zero pivot, no exact zero among the
        // entries of the super-panel's pivot rows in this block's columns (conservative, as in lu_wavepanel.hpp)
        bool z = false;
#pragma unroll
        for (int j = 0; j < 8; ++j) z = z || (uhas && xu[j] == 0.0);
        static_for<0, NSL>([&](auto st) {
            constexpr int S = decltype(st)::value;
            bool zz = false;
#pragma unroll
            for (int j = 0; j < 8; ++j) zz = zz || (x[S][j] == 0.0);
            z = z || (zz && pstep[S] >= b8 && pstep[S] != BIG);
        });
        const bool viol = special || status != 0 || __ballot(z) != 0ull;
        if (lane == 0) s_flag[wave] = viol ? 1 : 0;
        lds_barrier();
        if ((s_flag[0] | s_flag[1]) != 0) {  // hand the rest of this super-panel to the SLOW launch; nothing of this block has been stored
            if (threadIdx.x == 0) w.redo[b] = 1 + (b8 >> 3);
            return;
        }
        // ---- store the block: live rows and this block's pivot rows from their slots, the earlier pivot rows from wave 0's U slot
        static_for<0, NSL>([&](auto st) {
            constexpr int S = decltype(st)::value;
            if (pstep[S] == BIG || (pstep[S] >= b8 && pstep[S] != -1)) {
                double* __restrict__ dst = A + (long)(k0 + b8) * n + rowid[S];
#pragma unroll
                for (int j = 0; j < 8; ++j) dst[(long)j * n] = x[S][j];
            }
        });
        if (wave == 0 && uhas) {
            double* __restrict__ dst = A + (long)(k0 + b8) * n + urow;
#pragma unroll
            for (int j = 0; j < 8; ++j) dst[(long)j * n] = xu[j];
        }
    }

    // ---- positions, transposed L11 (multipliers of the pivot rows, read back from the matrix), compacted live list
    __syncthreads();
    static_for<0, NSL>([&](auto st) {
        constexpr int S = decltype(st)::value;
        if (pstep[S] >= 0) pos[rowid[S]] = rpos[S];
    });
    if (wave == 0 && lane < wsp) {  // lane = pivot index k: l11[kk * 64 + k] = multiplier of pivot row k for column kk < k
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
    unsigned long long balv[NSL];
    static_for<0, NSL>([&](auto st) {
        constexpr int S = decltype(st)::value;
        balv[S] = __ballot(pstep[S] == BIG);
        if (lane == 0) s_cnt[2 * S + wave] = __popcll(balv[S]);
    });
    lds_barrier();
    static_for<0, NSL>([&](auto st) {
        constexpr int S = decltype(st)::value;
        int base = 0;
        for (int g = 0; g < 2 * S + wave; ++g) base += s_cnt[g];
        if (pstep[S] == BIG) live[base + __popcll(balv[S] & ((1ull << lane) - 1ull))] = rowid[S];
    });
}

}  // namespace idahip
