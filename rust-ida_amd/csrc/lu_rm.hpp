// Batched dense LU with partial pivoting for gfx950, second generation: ROW-MAJOR work matrix, 16-column sub-panels,
// rank-64 trailing update. Replaces dense_get_rf (/root/reference/crates/linear/src/dense.rs:86-158) for a list of
// independent matrices; same bit-exactness contract as before:
//   * every element receives a -= a_kj * a_ik in ascending k, unfused (-ffp-contract=off), skipped when a_kj == 0
//     (dense.rs:148); multipliers are a_ik * (1/a_kk) (dense.rs:134-137);
//   * the pivot is the first row in the REFERENCE's current row order attaining max |a_ik| (dense.rs:111-117):
//     rows never move here (implicit pivoting); each row carries the position the reference would have put it at.
//
// Why this shape on MI355X (measured, see DESIGN.md section 4 and profiles/):
//   * column-major + implicit pivoting makes the pivot rows a strided gather with 8-16x line amplification and forces
//     8-byte-per-lane accesses; with the work matrix stored row-major every access of every kernel is a contiguous
//     row segment (rows are the unit that pivoting permutes), pivot rows included, and the row indirection is free.
//   * the right-looking trailing update is HBM-traffic bound (read+write of the trailing matrix once per panel):
//     64-column super-panels halve that traffic versus 32.
//   * the panel factorisation is a chain of 512 dependent pivot steps; what hides its latency is several workgroups
//     per CU, i.e. few registers: 16-column sub-panels (16 doubles per row) factored left-looking INSIDE the
//     super-panel (each sub-panel first applies the <= 48 pending pivots of its super-panel from LDS).
// Launch sequence per super-panel k0: lu_sub(s = 0..3) -> lu_trsm (U12 = L11^-1 A12, one column per lane, L11 through
// the scalar cache) -> lu_trail (64x64 tiles, 4x4 register tile per thread, operands from LDS). A final pass
// transposes rows (at their pivoted positions) into the column-major reference layout the solve kernels stream.
#pragma once
#include "common.hpp"

namespace idahip {

constexpr int RM_NBS = 16;  // sub-panel width
constexpr int RM_SB = 64;   // super-panel width = rank of the trailing update

struct RmWs {
    double* W;         // work matrices, ROW-major n x n per system: W[r*n + c]
    long wstride;
    const int* idx;    // [nsys] system ids (device)
    int n;
    int* pos;          // [batch][n] physical row -> position in the reference's row order
    int* live;         // [batch][n] rows not pivoted before the current super-panel, ascending physical index
    int* prow;         // [batch][n] pivot step -> physical row
    long long* piv;    // [batch][pstride] reference pivots
    long pstride;
    int* info;         // [batch] 0 | 1-based zero-pivot column
    double* l11;       // [batch][SB*SB] transposed unit-lower factor of the current super-panel: l11[kk*SB + k], k > kk
    int* uz;           // [batch][n/64+1] trailing 64-column block of U12 holds an exact zero
};

__device__ __forceinline__ void lds_barrier_rm() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ void rm_init_kernel(RmWs w) {
    const int b = w.idx[blockIdx.x];
    for (int i = threadIdx.x; i < w.n; i += blockDim.x) {
        w.pos[(long)b * w.n + i] = i;
        w.live[(long)b * w.n + i] = i;
    }
    if (threadIdx.x == 0) w.info[b] = 0;
}

// ------------------------------------------------------------------------------------------------ sub-panel
// One workgroup per matrix; thread t <-> row live[t] of the super-panel (rows pivoted by earlier sub-panels of the same
// super-panel keep their thread: they are the pending U rows). NBS columns of the row in registers.
template <int MAXT, int WPE>
__global__ __launch_bounds__(MAXT, WPE) void rm_sub_kernel(RmWs w, int k0, int s, int last) {
    constexpr int NBS = RM_NBS, SB = RM_SB, NW = MAXT / 64;
    constexpr int PMAX = SB - NBS;  // pending pivots at most
    static_assert(NW <= 16, "candidate scan assumes <= 16 waves");
    const int b = w.idx[blockIdx.x];
    if (w.info[b] != 0) return;
    const int n = w.n;
    double* __restrict__ W = w.W + (long)b * w.wstride;
    int* __restrict__ pos = w.pos + (long)b * n;
    int* __restrict__ live = w.live + (long)b * n;
    int* __restrict__ prow = w.prow + (long)b * n;
    long long* __restrict__ piv = w.piv + (long)b * w.pstride;
    double* __restrict__ l11 = w.l11 + (long)b * (SB + 1) * SB;

    const int c0 = k0 + s * NBS;
    const int wd = (n - c0) < NBS ? (n - c0) : NBS;
    const int P = s * NBS;   // pivots of this super-panel already chosen
    const int m = n - k0;    // rows of this super-panel's live list
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int nwaves = (int)(blockDim.x >> 6);

    __shared__ __align__(16) double s_c[PMAX][NBS];      // pending pivot rows (this sub-panel's columns), solved in place
    __shared__ double s_l[PMAX][PMAX + 1];               // pending L11: s_l[kk][k] = l(pivot k, column kk)
    __shared__ unsigned s_zm[PMAX];                      // per pending pivot row: bit j set when u(k, j) == 0
    __shared__ __align__(16) double s_row[2][NW][NBS + 2];  // candidate pivot rows; [NBS] = 1/pivot
    __shared__ __align__(16) double s_v[2][16];
    __shared__ __align__(16) int s_p[2][16];
    __shared__ unsigned s_cz[2][NW];
    __shared__ int s_cnt[NW];

    if (t < 32) {  // slots of waves that do not exist in this launch never win
        (&s_v[0][0])[t] = -2.0;
        (&s_p[0][0])[t] = 0x7fffffff;
    }
    const bool valid = t < m;
    const int r = valid ? live[t] : 0;
    int mypos = valid ? pos[r] : 0x7fffffff;
    double* __restrict__ Wr = W + (long)r * n;
    double a[NBS];
#pragma unroll
    for (int j = 0; j < NBS; ++j) a[j] = (valid && j < wd) ? Wr[c0 + j] : 0.0;

    // ---------------------------------------------------------------- pending pivots of this super-panel (left-looking)
    if (P > 0) {
        const bool pend = valid && mypos >= k0 && mypos < c0;
        const int kq = mypos - k0;
        if (pend) {
#pragma unroll
            for (int j = 0; j < NBS; ++j) s_c[kq][j] = a[j];
        }
        for (int e = t; e < P * P; e += (int)blockDim.x) {
            const int kk = e / P, k = e - kk * P;
            if (k > kk) s_l[kk][k] = l11[kk * SB + k];
        }
        if (t < P) s_zm[t] = 0u;
        __syncthreads();
        // U = L11^-1 C: one column per wave pass, lane <-> pending pivot index k, source row broadcast by readlane
        for (int j = wave; j < NBS; j += nwaves) {
            double val = (lane < P) ? s_c[lane][j] : 0.0;
#pragma unroll 1
            for (int kk = 0; kk + 1 < P; ++kk) {
                const double u = readlane_f64(val, kk);
                const double lk = (lane > kk && lane < P) ? s_l[kk][lane] : 0.0;
                const double tn = val - u * lk;                       // a(i,j) -= a_kj * a_ik, ascending kk
                val = (lane > kk && lane < P && u != 0.0) ? tn : val;  // dense.rs:148 skip
            }
            if (lane < P) {
                s_c[lane][j] = val;
                if (val == 0.0) atomicOr(&s_zm[lane], 1u << j);
            }
        }
        __syncthreads();
        if (pend) {
#pragma unroll
            for (int j = 0; j < NBS; ++j) a[j] = s_c[kq][j];  // final U entries of a pending pivot row
        } else if (valid && mypos >= c0) {
            // live row: subtract L(row, pending pivots) * U, multipliers are P contiguous doubles of the row
#pragma unroll 1
            for (int kb = 0; kb < P; kb += 8) {
                double l[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) l[u] = Wr[k0 + kb + u];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const unsigned zm = (unsigned)__builtin_amdgcn_readfirstlane((int)s_zm[kb + u]);
                    if (zm == 0u) {
#pragma unroll
                        for (int j = 0; j < NBS; j += 2) {
                            const double2 q = *reinterpret_cast<const double2*>(&s_c[kb + u][j]);
                            a[j] -= q.x * l[u];  // dense.rs:151
                            a[j + 1] -= q.y * l[u];
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < NBS; ++j)
                            if (!((zm >> j) & 1u)) a[j] -= s_c[kb + u][j] * l[u];
                    }
                }
            }
        }
    }
    __syncthreads();

    // ---------------------------------------------------------------- factor the NBS columns (rolled, registers rotate)
    bool alive = valid && mypos >= c0;
    int ownk = -1;
    bool failed = false;
    double o[NBS];  // the row's final entries for these columns, pushed in column order
#pragma unroll
    for (int j = 0; j < NBS; ++j) o[j] = 0.0;

#pragma unroll 1
    for (int k = 0; k < wd; ++k) {
        const int kc = c0 + k;
        const int par = k & 1;
        double v = -1.0;
        int p = 0x7fffffff;
        if (alive) {
            v = fabs(a[0]);
            p = mypos;
            if (v != v) v = (mypos == kc) ? __builtin_huge_val() : -1.0;  // NaN: dense.rs:111-117 scan semantics
        }
        const double myrecip = 1.0 / a[0];  // every lane, overlapping the reduction (dense.rs:134: a(k,k).recip())
        const double vm = wave_max_f64(v);
        const int pm = wave_min_i32(v == vm ? p : 0x7fffffff);
        const bool cand = alive && p == pm && v == vm;  // this wave's candidate (one lane or none)
        if (cand) {
#pragma unroll
            for (int j = 0; j < NBS; j += 2) {
                double2 q;
                q.x = a[j];
                q.y = a[j + 1];
                *reinterpret_cast<double2*>(&s_row[par][wave][j]) = q;
            }
            s_row[par][wave][NBS] = myrecip;
        }
        {   // zero mask of the candidate row (dense.rs:148) by one ballot over a re-read
            const double e = (lane < NBS) ? s_row[par][wave][lane] : 1.0;
            const unsigned zm = (unsigned)(__ballot(lane > 0 && lane < NBS && e == 0.0) & 0xffffffffull);
            if (lane == 0) {
                s_cz[par][wave] = zm;
                s_v[par][wave] = vm;  // -1 when the wave has no live row
                s_p[par][wave] = pm;
            }
        }
        lds_barrier_rm();
        double bv;
        int bp, bw;
        {   // global winner among <= 16 wave candidates: lane q takes candidate q, 4-step DPP fold inside the row
            const int q = lane & 15;
            const double cv = s_v[par][q];
            const int cp = s_p[par][q];
            double mv = cv, od;
            od = dpp_mov_f64<0x111, 0xf>(mv); mv = od > mv ? od : mv;
            od = dpp_mov_f64<0x112, 0xf>(mv); mv = od > mv ? od : mv;
            od = dpp_mov_f64<0x114, 0xf>(mv); mv = od > mv ? od : mv;
            od = dpp_mov_f64<0x118, 0xf>(mv); mv = od > mv ? od : mv;
            bv = readlane_f64(mv, 15);
            int key = (cv == bv) ? ((cp << 4) | q) : 0x7fffffff;  // positions < 2^27
            int oi;
            oi = dpp_mov_i32<0x111, 0xf>(key); key = oi < key ? oi : key;
            oi = dpp_mov_i32<0x112, 0xf>(key); key = oi < key ? oi : key;
            oi = dpp_mov_i32<0x114, 0xf>(key); key = oi < key ? oi : key;
            oi = dpp_mov_i32<0x118, 0xf>(key); key = oi < key ? oi : key;
            const int kmin = __builtin_amdgcn_readlane(key, 15);
            bp = kmin >> 4;
            bw = kmin & 15;
        }
        const double pk = s_row[par][bw][0];
        if (pk == 0.0) {  // zero pivot: Err(k+1)  (dense.rs:120-122)
            if (t == 0) w.info[b] = kc + 1;
            failed = true;
            break;
        }
        if (t == 0) piv[kc] = (long long)bp;
        const bool owner = alive && (mypos == bp);
        if (owner) {
            prow[kc] = r;
            ownk = k;
            alive = false;
            mypos = kc;
        }
        double push;
        if (alive) {  // live, not the pivot: multiplier + rank-1 update, result rotated one column to the left
            if (mypos == kc) mypos = bp;  // the row that sat at position k moves to the pivot's old position
            const double aik = a[0] * s_row[par][bw][NBS];
            push = aik;
            const unsigned zm = (unsigned)__builtin_amdgcn_readfirstlane((int)s_cz[par][bw]);
            if (zm == 0u) {
                double u[NBS];
#pragma unroll
                for (int j = 0; j < NBS; j += 2) {
                    const double2 q = *reinterpret_cast<const double2*>(&s_row[par][bw][j]);
                    u[j] = q.x;
                    u[j + 1] = q.y;
                }
#pragma unroll
                for (int j = 1; j < NBS; ++j) a[j - 1] = a[j] - u[j] * aik;  // dense.rs:151
            } else {
#pragma unroll
                for (int j = 1; j < NBS; ++j) a[j - 1] = ((zm >> j) & 1u) ? a[j] : a[j] - s_row[par][bw][j] * aik;
            }
        } else {      // the pivot row itself, or a row pivoted earlier: its entry of column kc is final; plain rotation
            push = a[0];
#pragma unroll
            for (int j = 1; j < NBS; ++j) a[j - 1] = a[j];
        }
        a[NBS - 1] = 0.0;
#pragma unroll
        for (int j = 1; j < NBS; ++j) o[j - 1] = o[j];
        o[NBS - 1] = push;
    }
    if (failed) return;
    // align: after wd pushes the row's entries sit in o[NBS-wd .. NBS-1]
#pragma unroll 1
    for (int x = wd; x < NBS; ++x) {
#pragma unroll
        for (int j = 1; j < NBS; ++j) o[j - 1] = o[j];
        o[NBS - 1] = 0.0;
    }
    if (valid) {
#pragma unroll
        for (int j = 0; j < NBS; ++j)
            if (j < wd) Wr[c0 + j] = o[j];
        pos[r] = mypos;
        if (ownk >= 0) {  // this row is pivot P+ownk of the super-panel: publish its multipliers (transposed L11)
            const int kk1 = P + ownk;
            for (int kk = 0; kk < P; ++kk) l11[kk * SB + kk1] = Wr[k0 + kk];
#pragma unroll
            for (int j = 0; j < NBS; ++j)
                if (j < ownk) l11[(P + j) * SB + kk1] = o[j];
        }
    }
    if (last) {  // compact the live list for the next super-panel / the trailing update
        const bool keep = valid && mypos >= c0 + wd;
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) s_cnt[wave] = __popcll(bal);
        __syncthreads();
        int base = 0;
        for (int q = 0; q < wave; ++q) base += s_cnt[q];
        if (keep) live[base + __popcll(bal & ((1ull << lane) - 1ull))] = r;
    }
}

// ------------------------------------------------------------------------------------------------ U12 = L11^-1 A12
// One workgroup per (matrix, 256 trailing columns), one column per lane; pivot rows are contiguous row segments; the
// 64x64 unit-lower L11 sits in LDS and is read as broadcasts. Rolled over the source row with register rotation:
// rr[0] is always the current source row, rr[i-1] <- rr[i] - u * l(kk, kk+i) (slots past the last row hold junk that
// is never stored) -- ~1 KB of code instead of a 2016-term unrolled triangle.
__global__ __launch_bounds__(256) void rm_trsm_kernel(RmWs w, int k0) {
    constexpr int SB = RM_SB;
    const int b = w.idx[blockIdx.x];
    if (w.info[b] != 0) return;
    const int n = w.n;
    double* __restrict__ W = w.W + (long)b * w.wstride;
    const int* __restrict__ prow = w.prow + (long)b * n + k0;
    const double* __restrict__ l11 = w.l11 + (long)b * (SB + 1) * SB;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int jc = k0 + SB + blockIdx.y * 256 + t;
    const bool real = jc < n;

    __shared__ double s_L[SB * SB + SB];
    __shared__ double s_out[16][256];  // finished U rows, flushed to the matrix 16 rows at a time (no global store in the
                                       // inner loop: a store there makes every compiler-placed vmcnt wait an HBM round trip)
    __shared__ int s_pr[SB];
    for (int e = t; e < SB * SB + SB; e += 256) s_L[e] = (e < SB * SB) ? l11[e] : 0.0;
    if (t < SB) s_pr[t] = prow[t];
    __syncthreads();
    if (k0 + SB + blockIdx.y * 256 + wave * 64 >= n) return;  // whole wave past the last column (no barrier below)
    const int jcl = real ? jc : n - 1;  // clamp: lanes past the last column load a valid address and never store

    double rr[SB];
#pragma unroll
    for (int k = 0; k < SB; ++k) rr[k] = W[(long)__builtin_amdgcn_readfirstlane(s_pr[k]) * n + jcl];
    bool anyz = false;
#pragma unroll 1
    for (int ch = 0; ch < SB; ch += 16) {
#pragma unroll 1
        for (int kk = ch; kk < ch + 16; ++kk) {
            const double ukk = rr[0];
            const bool z = real && (ukk == 0.0);
            anyz = anyz || z;
            s_out[kk - ch][t] = ukk;
            const int lb = kk * SB + kk;  // s_L[lb + i] = l(pivot kk+i, column kk)
            if (__ballot(z) == 0ull) {
#pragma unroll
                for (int i = 1; i < SB; ++i) rr[i - 1] = rr[i] - ukk * s_L[lb + i];  // a(i,j) -= a_kj * a_ik, ascending kk
            } else {
#pragma unroll
                for (int i = 1; i < SB; ++i) {
                    const double tn = rr[i] - ukk * s_L[lb + i];
                    rr[i - 1] = z ? rr[i] : tn;  // dense.rs:148
                }
            }
        }
        if (real) {
#pragma unroll
            for (int q = 0; q < 16; ++q) W[(long)__builtin_amdgcn_readfirstlane(s_pr[ch + q]) * n + jc] = s_out[q][t];
        }
    }
    const unsigned long long bal = __ballot(anyz);
    if (lane == 0) w.uz[(long)b * (n / 64 + 2) + blockIdx.y * 4 + wave] = bal != 0ull;
}

// ------------------------------------------------------------------------------------------------ rank-64 update
// One workgroup per (matrix, 64 trailing columns): U12 block (64 pivot rows x 64 columns) into LDS, then every 64-row
// tile of live rows: multipliers (64 contiguous doubles per row) transposed into LDS, 4 rows x 4 contiguous columns per
// thread in registers, 16-byte coalesced loads/stores (VEC = 2; n even) or 8-byte (VEC = 1).
template <int VEC>
__global__ __launch_bounds__(256) void rm_trail_kernel(RmWs w, int k0) {
    constexpr int SB = RM_SB, LD = 66;
    const int b = w.idx[blockIdx.x];
    if (w.info[b] != 0) return;
    const int n = w.n;
    double* __restrict__ W = w.W + (long)b * w.wstride;
    const int* __restrict__ live = w.live + (long)b * n;
    const int* __restrict__ prow = w.prow + (long)b * n + k0;
    const int mrem = n - k0 - SB;
    const int cb0 = k0 + SB + blockIdx.y * 64;
    const int ncols = (n - cb0) < 64 ? (n - cb0) : 64;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);

    __shared__ __align__(16) double Us[SB][LD];
    __shared__ __align__(16) double Ls[SB][LD];
    __shared__ unsigned short s_live[LU_MAX_N];

    for (int i = t; i < mrem; i += 256) s_live[i] = (unsigned short)live[i];
#pragma unroll
    for (int pass = 0; pass < SB / 4; ++pass) {
        const int k = pass * 4 + wave;
        Us[k][lane] = (lane < ncols) ? W[(long)ldc(prow + k) * n + cb0 + lane] : 0.0;
    }
    const bool slow = ldc(w.uz + (long)b * (n / 64 + 2) + blockIdx.y) != 0;
    __syncthreads();

    // thread (tx, ty): rows ty + 16 i, columns {2tx, 2tx+1} and {32+2tx, 32+2tx+1}: 16-byte accesses that are contiguous
    // across the 16 lanes of a row group, in global memory and in LDS (no bank conflicts on the U reads)
    const int tx = t & 15, ty = t >> 4;
    const int cxa = 2 * tx, cxb = 32 + 2 * tx;
    const int ntiles = (mrem + 63) >> 6;
    double lreg[16], creg[4][4];
    int crow[4];
    bool rok[4];
    auto load_tile = [&](int rt) {
        const int lr = rt * 64 + lane;
        const double* __restrict__ lsrc = W + (long)s_live[lr < mrem ? lr : mrem - 1] * n + k0 + wave * 16;
#pragma unroll
        for (int i = 0; i < 16; ++i) lreg[i] = lsrc[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ri = rt * 64 + ty + 16 * i;
            rok[i] = ri < mrem;
            crow[i] = s_live[rok[i] ? ri : mrem - 1];
            const double* __restrict__ src = W + (long)crow[i] * n + cb0;
            if (VEC == 2) {
                const double2 q0 = *reinterpret_cast<const double2*>(src + (cxa + 1 < ncols ? cxa : 0));
                const double2 q1 = *reinterpret_cast<const double2*>(src + (cxb + 1 < ncols ? cxb : 0));
                creg[i][0] = q0.x; creg[i][1] = q0.y; creg[i][2] = q1.x; creg[i][3] = q1.y;
            } else {
                creg[i][0] = (cxa < ncols) ? src[cxa] : 0.0;
                creg[i][1] = (cxa + 1 < ncols) ? src[cxa + 1] : 0.0;
                creg[i][2] = (cxb < ncols) ? src[cxb] : 0.0;
                creg[i][3] = (cxb + 1 < ncols) ? src[cxb + 1] : 0.0;
            }
        }
    };
    load_tile(0);
#pragma unroll 1
    for (int rt = 0; rt < ntiles; ++rt) {
#pragma unroll
        for (int i = 0; i < 16; ++i) Ls[wave * 16 + i][lane] = lreg[i];
        lds_barrier_rm();
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
        if (rt + 1 < ntiles) load_tile(rt + 1);  // in flight behind the arithmetic below
        if (!slow) {
#pragma unroll 8
            for (int k = 0; k < SB; ++k) {
                double lv[4], uv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) lv[i] = Ls[k][ty + 16 * i];
                {
                    const double2 q0 = *reinterpret_cast<const double2*>(&Us[k][cxa]);
                    const double2 q1 = *reinterpret_cast<const double2*>(&Us[k][cxb]);
                    uv[0] = q0.x; uv[1] = q0.y; uv[2] = q1.x; uv[3] = q1.y;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) c[i][j] -= uv[j] * lv[i];  // dense.rs:151, unfused
            }
        } else {
#pragma unroll 4
            for (int k = 0; k < SB; ++k) {
                double lv[4], uv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) lv[i] = Ls[k][ty + 16 * i];
                uv[0] = Us[k][cxa]; uv[1] = Us[k][cxa + 1]; uv[2] = Us[k][cxb]; uv[3] = Us[k][cxb + 1];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const double tn = c[i][j] - uv[j] * lv[i];
                        c[i][j] = (uv[j] != 0.0) ? tn : c[i][j];  // dense.rs:148
                    }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (!sok[i]) continue;
            double* __restrict__ dst = W + (long)srow[i] * n + cb0;
            if (VEC == 2 && cxa + 1 < ncols) {
                double2 q0;
                q0.x = c[i][0]; q0.y = c[i][1];
                *reinterpret_cast<double2*>(dst + cxa) = q0;
            } else {
                if (cxa < ncols) dst[cxa] = c[i][0];
                if (cxa + 1 < ncols) dst[cxa + 1] = c[i][1];
            }
            if (VEC == 2 && cxb + 1 < ncols) {
                double2 q1;
                q1.x = c[i][2]; q1.y = c[i][3];
                *reinterpret_cast<double2*>(dst + cxb) = q1;
            } else {
                if (cxb < ncols) dst[cxb] = c[i][2];
                if (cxb + 1 < ncols) dst[cxb + 1] = c[i][3];
            }
        }
        lds_barrier_rm();  // everyone is done reading Ls before the next tile's multipliers overwrite it
    }
}

// ------------------------------------------------------------------------------------------------ layout changes
// out(col-major)[j*n + p] = W[prow[p]*n + j]: 64 positions x 64 columns per workgroup through an LDS tile, both sides
// coalesced; perm[p] = prow[p].
__global__ __launch_bounds__(256) void rm_finalize_kernel(RmWs w, double* __restrict__ out, long ostride, int* __restrict__ perm) {
    const int b = w.idx[blockIdx.x];
    if (w.info[b] != 0) return;
    const int n = w.n;
    const double* __restrict__ W = w.W + (long)b * w.wstride;
    const int* __restrict__ prow = w.prow + (long)b * n;
    double* __restrict__ O = out + (long)b * ostride;
    const int p0 = blockIdx.y * 64, j0 = blockIdx.z * 64;
    __shared__ double tile[64][65];
    const int t = threadIdx.x, lane = t & 63, grp = t >> 6;
    for (int pp = grp; pp < 64; pp += 4) {
        const int p = p0 + pp;
        if (p < n && j0 + lane < n) tile[pp][lane] = W[(long)prow[p] * n + j0 + lane];
    }
    if (perm && blockIdx.z == 0 && t < 64 && p0 + t < n) perm[(long)b * n + p0 + t] = prow[p0 + t];
    __syncthreads();
    for (int jj = grp; jj < 64; jj += 4) {
        const int j = j0 + jj;
        if (j < n && p0 + lane < n) O[(long)j * n + p0 + lane] = tile[lane][jj];
    }
}

// W(row-major)[r*n + c] = in(col-major)[c*n + r]
__global__ __launch_bounds__(256) void rm_transpose_in_kernel(RmWs w, const double* __restrict__ in, long istride) {
    const int b = w.idx[blockIdx.x];
    const int n = w.n;
    double* __restrict__ W = w.W + (long)b * w.wstride;
    const double* __restrict__ I = in + (long)b * istride;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.z * 64;
    __shared__ double tile[64][65];
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    for (int cc = grp; cc < 64; cc += 4)
        if (c0 + cc < n && r0 + lane < n) tile[cc][lane] = I[(long)(c0 + cc) * n + r0 + lane];
    __syncthreads();
    for (int rr = grp; rr < 64; rr += 4)
        if (r0 + rr < n && c0 + lane < n) W[(long)(r0 + rr) * n + c0 + lane] = tile[lane][rr];
}

// ------------------------------------------------------------------------------------------------ host driver
// Factor the ROW-major work matrices W[b] of the listed systems into out[b] (column-major reference layout).
inline int rm_factor_batched(idahip_ctx* c, double* Wm, long wstride, double* out, long ostride, long long* piv, long pstride,
                             int* perm, const int* d_idx, int nsys) {
    const int n = c->n;
    RmWs w;
    w.W = Wm; w.wstride = wstride; w.idx = d_idx; w.n = n; w.pos = c->lu_pos; w.live = c->lu_live; w.prow = c->lu_prow;
    w.piv = piv; w.pstride = pstride; w.info = c->lu_info; w.l11 = c->lu_l11; w.uz = c->lu_uz;
    hipLaunchKernelGGL(rm_init_kernel, dim3(nsys), dim3(256), 0, c->stream, w);
    for (int k0 = 0; k0 < n; k0 += RM_SB) {
        const int m = n - k0;
        const int threads = ((m + 63) / 64) * 64;
        const int nsub = ((m < RM_SB ? m : RM_SB) + RM_NBS - 1) / RM_NBS;
        for (int s = 0; s < nsub; ++s) {
            const int last = (s == nsub - 1);
            if (threads <= 512)
                hipLaunchKernelGGL((rm_sub_kernel<512, 2>), dim3(nsys), dim3(threads), 0, c->stream, w, k0, s, last);
            else
                hipLaunchKernelGGL((rm_sub_kernel<1024, 4>), dim3(nsys), dim3(threads), 0, c->stream, w, k0, s, last);
        }
        const int ntrail = n - k0 - RM_SB;
        if (ntrail > 0) {
            const int nblk = (ntrail + 63) / 64;
            hipLaunchKernelGGL(rm_trsm_kernel, dim3(nsys, (ntrail + 255) / 256), dim3(256), 0, c->stream, w, k0);
            if (n % 2 == 0)
                hipLaunchKernelGGL(rm_trail_kernel<2>, dim3(nsys, nblk), dim3(256), 0, c->stream, w, k0);
            else
                hipLaunchKernelGGL(rm_trail_kernel<1>, dim3(nsys, nblk), dim3(256), 0, c->stream, w, k0);
        }
    }
    const int nb64 = (n + 63) / 64;
    hipLaunchKernelGGL(rm_finalize_kernel, dim3(nsys, nb64, nb64), dim3(256), 0, c->stream, w, out, ostride, perm);
    return 0;
}

}  // namespace idahip
