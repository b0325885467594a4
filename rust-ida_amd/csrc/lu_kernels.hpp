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
//   * right-looking blocked algorithm over the whole list of matrices (lock-step batch, thousands of workgroups per
//     launch), 64-column super-panels. Default pipeline (LU variant 4), per super-panel:
//       lu_wavepanel (lu_wavepanel.hpp): one wavefront per matrix factors the whole super-panel (<= 512 live rows);
//       lu_trail64w              : per (matrix, 64 trailing columns): pivot-row gather, U12 = L11^-1 A12 in LDS, then
//                                  A22 -= L21 U12 in wave-private 16-row strips, rank 64, 4x4 register tile per lane.
//     Variant 3 (cross-check in the tests, and the path for more than 512 live rows) factors the super-panel with
//       lu_panel2 (k0) -> lu_trail<32> narrow (the first panel's update of the other 32 columns) -> lu_panel2 (k0 + 32):
//       one workgroup per matrix, two live rows per lane, 32 panel entries per row in registers, fused-DPP + LDS arg-max.
//     More than 1024 live rows: eight 8-column panels with eight rows per lane (lu_panelr) per super-panel.
//   * a final pass scatters rows to their pivoted positions (the reference layout the solve kernels stream).
// Blocking changes neither the per-element operation order nor any operand, only when each update is applied.
#pragma once
#include <type_traits>
#include "common.hpp"

namespace idahip {

constexpr int LU_SPLIT_BLOCKS = 2;  // lu_trail64w_kernel, n > 1024: column blocks per matrix whose live rows several workgroups may share
constexpr int LU_SPARSE_COLS = 8;   // trailing kernels, n > 1024: at most this many non-zero columns of U12 -> the row-per-thread update

struct LuWs {
    double* mats;      // work matrices (physical row order), column-major n x n
    long mstride;      // elements between consecutive systems
    const int* idx;    // [nsys] system ids (device)
    const int* cnt;    // null, or the length of idx on the device: launches are then sized for the worst case and the surplus
                       // workgroups leave at once (the device-resident lock-step stepper does not know the list's length on the host)
    int n;
    int* pos;          // [batch][n] physical row -> current reference position
    int* live;         // [batch][n] sorted physical indices of not-yet-pivoted rows
    int* prow;         // [batch][n] pivot step -> physical row
    long long* piv;    // [batch][n] reference pivots (position chosen at step k)
    long pstride;      // elements between consecutive systems in piv
    int* info;         // [batch]   0 | 1-based zero-pivot column
    int* nzb;          // [batch]   lu_trail64w_kernel, n > 1024: column blocks of the first super-panel's update that had work
    int* bz;           // [batch][64] n > 1024: per column block of the current super-panel's update, 1 = its pivot rows have a non-zero entry
    unsigned char* zmap;  // null, or [batch][64][64] (n >= 2048): zmap[K][I] = 1 when block (rows I, columns K) of the factors may be non-zero
    unsigned char* dirty; // null, or [batch][64][64] with zmap: dirty[K][I] = 1 once block (rows I, columns K) of `out` has held a value with non-zero bits; never reset
    int* jwzero;       // null, or [batch] (with dirty, work == the ctx's work matrix, every super-panel 64 columns wide): 1 = the factorisation has left the
                       // work matrix all +0.0 (lu_finalize_kernel clears what it reads and the pivot rows' entries right of their super-panel); heat_jac_kernel
                       // then writes the band only. lu_init_kernel resets it, lu_finalize_kernel sets it.
    int* redo;         // [batch]   lu_wavepanel_kernel: 0 | 1 + first 8-column block of the super-panel left to its SLOW launch
    double* l11;       // [batch][L11_STRIDE] transposed L11: l11[kk*l11ld + k] = multiplier of the k-th pivot row for column kk
    int l11ld;         // row length of l11: the (super-)panel width, 32 or 64
    unsigned long long* stamps;  // timing builds (-DIDAHIP_STAMPS): time stamps of one launch, null otherwise
    double* out;       // factors in the reference layout (rows at their pivoted positions), column-major n x n
    long ostride;      // elements between consecutive systems in out
};
constexpr int L11_STRIDE = 64 * 64;  // elements per system in LuWs::l11

__global__ void lu_init_kernel(LuWs w) {
    if (w.cnt && (int)blockIdx.x >= *w.cnt) return;
    const int b = w.idx[blockIdx.x];
    for (int i = threadIdx.x; i < w.n; i += blockDim.x) {
        w.pos[(long)b * w.n + i] = i;
        w.live[(long)b * w.n + i] = i;
    }
    if (threadIdx.x == 0) {
        w.info[b] = 0;
        w.nzb[b] = 0;
        if (w.jwzero) w.jwzero[b] = 0;
    }
    if (w.zmap)
        for (int i = threadIdx.x; i < 4096 / 8; i += blockDim.x) reinterpret_cast<unsigned long long*>(w.zmap + (long)b * 4096)[i] = 0ull;
}

// A wave-uniform value as a per-lane value the optimiser cannot see through: `u == 0 ? a : b` then stays a pair of
// v_cndmask instead of becoming a scalar branch around the update (a branch around 64 panel registers makes the
// register allocator keep two copies of them).
__device__ __forceinline__ double opaque_vgpr(double u) {
    asm volatile("" : "+v"(u));
    return u;
}

// LDS-only workgroup barrier: unlike __syncthreads() it does not drain the vector-memory counter, so global stores issued
// inside a latency-critical loop stay in flight across it.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ------------------------------------------------------------------------------------------------ panel, two rows per thread
// Same contract as lu_panel_kernel. A step of the panel is one long dependent chain (arg-max -> LDS -> barrier ->
// winner scan -> LDS -> update) of ~450 instructions per wave whatever the number of rows a lane carries; with two
// live rows per lane (row t and row t + T of the live list) the same chain serves twice the rows, the wave count per
// matrix halves, and two matrices share a CU where one used to sit (~190 VGPRs: 2 x 32 panel entries per lane).
template <int NB, int MAXT, int WPE>
__global__ __launch_bounds__(MAXT, WPE) void lu_panel2_kernel(LuWs w, int k0, int lbase) {
    constexpr int R = 2;
    constexpr int NW = MAXT / 64;
    constexpr int LDR = NB + 2;  // row slot: NB entries, [NB] = 1/pivot
    static_assert(NW <= 16 && NB <= 64, "candidate scan assumes <= 16 waves, zero mask assumes NB <= 64");
    if (w.cnt && (int)blockIdx.x >= ldc(w.cnt)) return;  // (list length on the device: surplus workgroups leave)
    const int b = w.idx[blockIdx.x];
    if (w.info[b] != 0) return;
    const int n = w.n;
    double* __restrict__ A = w.mats + (long)b * w.mstride;
    int* __restrict__ pos = w.pos + (long)b * n;
    int* __restrict__ live = w.live + (long)b * n;
    int* __restrict__ prow = w.prow + (long)b * n;
    long long* __restrict__ piv = w.piv + (long)b * w.pstride;
    double* __restrict__ l11 = w.l11 + (long)b * L11_STRIDE;

    const int m = n - k0;
    const int wd = m < NB ? m : NB;
    const int T = blockDim.x;  // R * T >= m
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);

    __shared__ __align__(16) double s_row[2][NW][LDR];
    __shared__ unsigned s_kh[2][16], s_kl[2][16];
    __shared__ __align__(16) int s_p[2][16];
    __shared__ unsigned long long s_zm[2][NW];
    __shared__ int s_r[2][NW];
    __shared__ int s_cnt[R][NW];

    if (t < 32) {  // slots of waves that do not exist in this launch never win
        (&s_kh[0][0])[t] = 0u;
        (&s_kl[0][0])[t] = 0u;
        (&s_p[0][0])[t] = 0x7fffffff;
    }
    bool valid[R], alive[R];
    int r[R], mypos[R], ownk[R];
    double a[R][NB];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int li = t + i * T;
        valid[i] = li < m;
        alive[i] = valid[i];
        ownk[i] = -1;
        r[i] = valid[i] ? live[li] : 0;
        mypos[i] = valid[i] ? pos[r[i]] : 0x7fffffff;
#pragma unroll
        for (int j = 0; j < NB; ++j) a[i][j] = (valid[i] && j < wd) ? A[(long)(k0 + j) * n + r[i]] : 0.0;
    }
    __syncthreads();

    bool failed = false;
    auto step = [&](const int k) -> bool {
        const int kc = k0 + k;
        const int par = k & 1;
        // candidate key of a live row: the bit pattern of |a| with the always-clear sign bit set (0 = no candidate);
        // ties go to the lowest position; NaN only wins at position kc (dense.rs:111-117 scan semantics)
        unsigned kh[R], kl[R];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            kh[i] = 0u;
            kl[i] = 0u;
            if (alive[i]) {
                const double v = fabs(a[i][0]);
                kh[i] = (unsigned)__double2hiint(v) | 0x80000000u;
                kl[i] = (unsigned)__double2loint(v);
                if (v != v) {
                    kh[i] = (mypos[i] == kc) ? 0xfff00000u : 0u;
                    kl[i] = 0u;
                }
            }
        }
        // the better of the lane's own two rows
        const bool s1 = kh[1] > kh[0] || (kh[1] == kh[0] && (kl[1] > kl[0] || (kl[1] == kl[0] && mypos[1] < mypos[0])));
        const unsigned bkh = s1 ? kh[1] : kh[0], bkl = s1 ? kl[1] : kl[0];
        const int bpos = s1 ? mypos[1] : mypos[0];
        const double myrecip = 1.0 / (s1 ? a[1][0] : a[0][0]);  // mult = a(k,k).recip() (dense.rs:134), off the critical path
        const unsigned mh = wave_max_u32<false>(bkh);
        const unsigned ml = wave_max_u32<false>(bkh == mh ? bkl : 0u);
        const bool top = bkh != 0u && bkh == mh && bkl == ml;
        const int pm = wave_min_i32f<false>(top ? bpos : 0x7fffffff);
        const bool cand = top && bpos == pm;  // this wave's candidate row (one lane, or none)
        if (cand && !s1) {
#pragma unroll
            for (int j = 0; j < NB; j += 2) {
                double2 q;
                q.x = a[0][j];
                q.y = a[0][j + 1];
                *reinterpret_cast<double2*>(&s_row[par][wave][j]) = q;
            }
            s_r[par][wave] = r[0];
        }
        if (cand && s1) {
#pragma unroll
            for (int j = 0; j < NB; j += 2) {
                double2 q;
                q.x = a[1][j];
                q.y = a[1][j + 1];
                *reinterpret_cast<double2*>(&s_row[par][wave][j]) = q;
            }
            s_r[par][wave] = r[1];
        }
        if (cand) s_row[par][wave][NB] = myrecip;
        {   // zero mask of the candidate row (dense.rs:148): one entry per lane, one ballot
            const double e = (lane < NB) ? s_row[par][wave][lane] : 1.0;
            const unsigned long long zm = __ballot(lane > 0 && lane < NB && e == 0.0);
            if (lane == 0) {
                s_zm[par][wave] = zm;
                s_kh[par][wave] = mh;  // 0 when the wave has no live row
                s_kl[par][wave] = ml;
                s_p[par][wave] = pm;
            }
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
        const double pk = s_row[par][bw][0];
        if (pk == 0.0) {  // zero pivot: Err(k+1)  (dense.rs:120-122)
            if (t == 0) w.info[b] = kc + 1;
            return false;
        }
        if (t == 0) piv[kc] = (long long)bp;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            if (alive[i] && mypos[i] == bp) {  // this row is the pivot
                prow[kc] = r[i];
                ownk[i] = k;
                alive[i] = false;
                mypos[i] = kc;
            } else if (alive[i] && mypos[i] == kc) {
                mypos[i] = bp;  // the row that sat at position k moves to the pivot's old position
            }
        }
        // the pivot row is final for the panel columns: one cooperative store of pivot + U entries from the LDS copy
        if (wave == 0 && lane < NB && k + lane < wd)
            A[(long)(kc + lane) * n + s_r[par][bw]] = s_row[par][bw][lane];
        const double recip = s_row[par][bw][NB];
        double aik[R];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            aik[i] = a[i][0] * recip;
            if (alive[i]) A[(long)kc * n + r[i]] = aik[i];  // the multiplier is final (coalesced column store)
        }
        const unsigned long long zmv = s_zm[par][bw];
        const unsigned long long zm = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(zmv >> 32)) << 32) |
                                      (unsigned)__builtin_amdgcn_readfirstlane((int)(zmv & 0xffffffffull));
        // the arithmetic runs on every lane (rows that are not live compute values nobody reads)
        if (zm == 0ull) {
#pragma unroll
            for (int jc = 0; jc < NB; jc += 8) {
                double u[8];
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    const double2 q = *reinterpret_cast<const double2*>(&s_row[par][bw][jc + j]);
                    u[j] = q.x;
                    u[j + 1] = q.y;
                }
#pragma unroll
                for (int i = 0; i < R; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (jc + j >= 1) a[i][jc + j - 1] = a[i][jc + j] - u[j] * aik[i];  // dense.rs:151, rotated one column
                __builtin_amdgcn_sched_barrier(0);  // keep the pivot-row reads of later chunks from piling up in registers
            }
        } else {
#pragma unroll
            for (int j = 1; j < NB; ++j) {
                const double uj = s_row[par][bw][j];
#pragma unroll
                for (int i = 0; i < R; ++i) a[i][j - 1] = ((zm >> j) & 1ull) ? a[i][j] : a[i][j] - uj * aik[i];
            }
        }
#pragma unroll
        for (int i = 0; i < R; ++i) a[i][NB - 1] = 0.0;
        return true;
    };
#pragma unroll 1
    for (int k = 0; k < wd; ++k)
        if (!step(k)) { failed = true; break; }
    if (failed) return;

    // positions, transposed L11 (multipliers of the pivot rows, read back from the matrix) and the compacted live list
#pragma unroll
    for (int i = 0; i < R; ++i)
        if (valid[i]) pos[r[i]] = mypos[i];
    __syncthreads();  // the multipliers stored above are visible to the whole workgroup
#pragma unroll
    for (int i = 0; i < R; ++i) {
        if (ownk[i] >= 0) {
            const int kq = lbase + ownk[i];  // index of this pivot row inside the enclosing super-panel
            const double* __restrict__ src = A + (long)(k0 - lbase) * n + r[i];
            for (int j0 = 0; j0 < kq; j0 += 8) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = (j0 + u < kq) ? src[(long)(j0 + u) * n] : 0.0;
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (j0 + u < kq) l11[(j0 + u) * w.l11ld + kq] = v[u];  // [kk][k]: a TRSM step reads a contiguous run
            }
        }
    }
    // live list: the surviving rows of set 0 in thread order, then those of set 1 (the list stays sorted)
    unsigned long long bal[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        bal[i] = __ballot(alive[i]);
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
        if (alive[i]) live[mine + __popcll(bal[i] & ((1ull << lane) - 1ull))] = r[i];
    }
}

// f(integral_constant<K>) for K = 0 .. min(N, wd) - 1, stopping at the first false: a loop the compiler cannot decline to unroll
template <int K, int N, class F>
__device__ __forceinline__ bool static_steps(F& f, int wd) {
    if constexpr (K < N) {
        if (K >= wd) return true;
        if (!f(std::integral_constant<int, K>{})) return false;
        return static_steps<K + 1, N>(f, wd);
    }
    return true;
}

// ------------------------------------------------------------------------------------------------ panel, R rows per thread
// lu_panel2_kernel for any number R of live rows per lane (rows t + i T of the live list): with narrow panels (NB = 8) and
// R = 8 a 512-thread workgroup factors a panel of up to 4096 rows (R * NB * 2 = 128 VGPRs of panel entries). Used for the
// leading super-panels of matrices with more than 1024 rows; the two-row kernel takes over below that.
template <int NB, int R, int MAXT, int WPE>
__global__ __launch_bounds__(MAXT, WPE) void lu_panelr_kernel(LuWs w, int k0, int lbase) {
    constexpr int NW = MAXT / 64;
    constexpr int LDR = NB + 2;  // row slot: NB entries, [NB] = 1/pivot
    static_assert(NW <= 16 && NB <= 64, "candidate scan assumes <= 16 waves, zero mask assumes NB <= 64");
    if (w.cnt && (int)blockIdx.x >= ldc(w.cnt)) return;  // (list length on the device: surplus workgroups leave)
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

    const int m = n - k0;
    const int wd = m < NB ? m : NB;
    const int T = blockDim.x;  // R * T >= m
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);

    __shared__ __align__(16) double s_row[2][NW][LDR];
    __shared__ unsigned s_kh[2][16], s_kl[2][16];
    __shared__ __align__(16) int s_p[2][16];
    __shared__ int s_r[2][NW];
    __shared__ int s_cnt[R][NW];

    // (timing builds only, exp_switches.hpp: tb::STAMPS is -1 in the product and everything below folds away)
    unsigned long long* st = (tb::STAMPS >= 0 && k0 == 0 && w.stamps) ? w.stamps + (size_t)blockIdx.x * 8 : nullptr;
#define STAMP(i) do { if (tb::STAMPS >= 0 && st && t == 0) st[i] = wall_clock64(); } while (0)
    STAMP(0);
    if (t < 32) {  // slots of waves that do not exist in this launch never win
        (&s_kh[0][0])[t] = 0u;
        (&s_kl[0][0])[t] = 0u;
        (&s_p[0][0])[t] = 0x7fffffff;
    }
    bool valid[R], alive[R];
    int r[R], mypos[R];
    double a[R][NB];
    // The lane's R rows of the panel stay in registers for the whole kernel, column j of the panel in a[.][j]: step k turns
    // a[.][k] of the live rows into multipliers and updates the columns right of it, so nothing is written to the matrix
    // inside the step loop except the pivot row (from its LDS copy); the multipliers leave in one burst after the last step.
    // (The first version rotated the columns and stored each step's multipliers at once: with the panel registers at the
    // limit the row offsets were spilled, and every reload waited (vmcnt(0)) for the store before it -- R write round trips
    // per step, 11 us of a 14 us step at 4096 live rows; tools/stamps_panelr.py.)
    // Every load is unconditional (indices clamped, results masked afterwards): a guarded load makes the compiler wait for
    // each one before it issues the next, and the R * NB column segments then arrive one round trip at a time.
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int li = t + i * T;
        valid[i] = li < m;
        alive[i] = valid[i];
        r[i] = live[valid[i] ? li : m - 1];
    }
#pragma unroll
    for (int i = 0; i < R; ++i) {
        mypos[i] = pos[r[i]];
#pragma unroll
        for (int j = 0; j < NB; ++j) a[i][j] = buf_load_f64(rsrc, (unsigned)r[i] * 8u, (k0 + (j < wd ? j : wd - 1)) * n * 8);
    }
#pragma unroll
    for (int i = 0; i < R; ++i) {
        if (!valid[i]) {
            r[i] = 0;
            mypos[i] = 0x7fffffff;
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) a[i][j] = (valid[i] && j < wd) ? a[i][j] : 0.0;
    }
    STAMP(1);
    __syncthreads();

    bool failed = false;
    auto step = [&](auto kconst) -> bool {  // one instantiation per pivot step: k is a constant (static register indices)
        constexpr int k = decltype(kconst)::value;
        const int kc = k0 + k;
        const int par = k & 1;
        // candidate key of a live row: the bit pattern of |a| with the always-clear sign bit set (0 = no candidate);
        // ties go to the lowest position; NaN only wins at position kc (dense.rs:111-117 scan semantics)
        // the best of the lane's own rows, one row at a time (selects, not branches)
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
            kh = alive[i] ? kh : 0u;
            kl = alive[i] ? kl : 0u;
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
                const bool isp = alive[i] && mypos[i] == bp;           // this row is the pivot
                const bool mv = alive[i] && !isp && mypos[i] == kc;    // the row that sat at position k moves to the pivot's old position
                anyp = anyp || isp;
                pr_mine = isp ? r[i] : pr_mine;
                mypos[i] = isp ? kc : (mv ? bp : mypos[i]);
                alive[i] = alive[i] && !isp;
            }
            if (anyp) prow[kc] = pr_mine;
        }
        // the pivot row is final for the panel columns: one cooperative store of pivot + U entries from the LDS copy
        if (wave == 0 && lane >= k && lane < wd)
            A[(long)(k0 + lane) * n + s_r[par][bw]] = s_row[par][bw][lane];
        const double recip = s_row[par][bw][NB];
        // the pivot row right of the pivot, and which of its entries are zero (dense.rs:148 skips those columns): the test is
        // made on a per-lane copy the optimiser cannot see through, so the skip stays a pair of v_cndmask per entry -- a
        // uniform branch around the 2 R NB panel registers makes the register allocator keep two copies of them and spill
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
        // multipliers (kept in a[.][k]) and the updates right of column k. The arithmetic runs on every lane: a row that is a
        // pivot already only overwrites entries that were stored when it was chosen (columns >= its own step) and are not
        // read again; its multipliers sit left of that and are not touched.
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const double aik = a[i][k] * recip;  // dense.rs:134-137
            a[i][k] = aik;
#pragma unroll
            for (int j = k + 1; j < NB; ++j) a[i][j] = uz[j] ? a[i][j] : a[i][j] - u[j] * aik;  // dense.rs:148-151
        }
        if (k < 4) STAMP(4 + k);
        return true;
    };
    failed = !static_steps<0, NB>(step, wd);
    STAMP(2);
    if (failed) return;

    // multipliers: a live row has one in every panel column, a pivot row in the columns left of its own step
    // (its position is its pivot column, so that step is mypos - k0)
    int lim[R];
#pragma unroll
    for (int i = 0; i < R; ++i) lim[i] = !valid[i] ? 0 : alive[i] ? wd : mypos[i] - k0;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
#pragma unroll
        for (int i = 0; i < R; ++i)
            if (j < lim[i]) buf_store_f64(rsrc, (unsigned)r[i] * 8u, (k0 + j) * n * 8, a[i][j]);
    }
    // positions, transposed L11 (multipliers of the pivot rows: earlier panels' from the matrix, this panel's from registers)
    // and the compacted live list
#pragma unroll
    for (int i = 0; i < R; ++i)
        if (valid[i]) pos[r[i]] = mypos[i];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        if (valid[i] && !alive[i]) {
            const int kq = lbase + lim[i];  // index of this pivot row inside the enclosing super-panel
            const double* __restrict__ src = A + (long)(k0 - lbase) * n + r[i];
            for (int j0 = 0; j0 < lbase; j0 += 8) {  // lbase is a multiple of NB
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = (j0 + u < lbase) ? src[(long)(j0 + u) * n] : 0.0;
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (j0 + u < lbase) l11[(j0 + u) * w.l11ld + kq] = v[u];  // [kk][k]: a TRSM step reads a contiguous run
            }
#pragma unroll
            for (int j = 0; j < NB; ++j)
                if (j < lim[i]) l11[(lbase + j) * w.l11ld + kq] = a[i][j];
        }
    }
    // live list: the surviving rows of set 0 in thread order, then those of set 1 (the list stays sorted)
    unsigned long long bal[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        bal[i] = __ballot(alive[i]);
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
        if (alive[i]) live[mine + __popcll(bal[i] & ((1ull << lane) - 1ull))] = r[i];
    }
    STAMP(3);
#undef STAMP
}

// ------------------------------------------------------------------------------------------------ zero column blocks (n > 1024)
// One workgroup per (matrix, block of 64 trailing columns) of a 64-column super-panel's update: gathers the block's entries
// of the 64 pivot rows (one pivot row per lane, a column per load: pivot rows that are neighbours in memory share cache
// lines) and, if all are zero, writes them to the factors as they are -- a_kj == 0 leaves the column untouched in the
// triangular solve and subtracts nothing from the rows below (dense.rs:148). LuWs::bz tells lu_trail64w_kernel which blocks
// are left. A banded matrix in dense storage (the heat equation's Jacobian) has almost only such blocks; this kernel holds
// no LDS to speak of and few registers, so the chip is full of gathers, where the update kernel fits two workgroups per CU.
// k0 == 0 also counts the blocks with work (LuWs::nzb: the row split of lu_trail64w_kernel).
__global__ __launch_bounds__(256) void lu_u12_zero_kernel(LuWs w, int k0, int nsys, int ncb) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int cbi = slot % ncb, mi = (slot / ncb) * 8 + xcd;  // same dealing as lu_trail64w_kernel
    if (mi >= (w.cnt ? ldc(w.cnt) : nsys)) return;
    const int b = w.idx[mi];
    if (w.info[b] != 0) return;
    const int n = w.n;
    const double* __restrict__ A = w.mats + (long)b * w.mstride;
    double* __restrict__ O = w.out + (long)b * w.ostride;
    const int cb0 = k0 + 64 + cbi * 64;
    const int ncols = (n - cb0) < 64 ? (n - cb0) : 64;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    __shared__ int s_nz[4], s_bits[4];
    const int pr = w.prow[(long)b * n + k0 + lane];
    double g[16];
    bool nz = false, bits = false;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int cc = wave * 16 + i;
        g[i] = A[(long)(cb0 + (cc < ncols ? cc : 0)) * n + pr];
        nz = nz || (cc < ncols && g[i] != 0.0);
        bits = bits || (cc < ncols && __double_as_longlong(g[i]) != 0ll);  // (a -0.0 is content)
    }
    const bool wnz = __ballot(nz) != 0ull, wbits = __ballot(bits) != 0ull;
    if (lane == 0) { s_nz[wave] = wnz ? 1 : 0; s_bits[wave] = wbits ? 1 : 0; }
    __syncthreads();
    const int any = s_nz[0] | s_nz[1] | s_nz[2] | s_nz[3];
    const int anyb = s_bits[0] | s_bits[1] | s_bits[2] | s_bits[3];
    const long blk = (long)b * 4096 + ((k0 >> 6) + 1 + cbi) * 64 + (k0 >> 6);  // U block (rows k0 / 64, these columns)
    if (t == 0) {
        w.bz[b * 64 + cbi] = any;
        if (any && k0 == 0) atomicAdd(w.nzb + b, 1);
        if (any && w.zmap) w.zmap[blk] = 1;
        if (anyb && w.dirty) w.dirty[blk] = 1;  // (`any`: lu_trail64w_kernel will write this block's U12)
    }
    if (any) return;
    // all-zero-bits rows bound for a block of the ctx's factors that has only ever held zeros are there already (lu_finalize_kernel)
    if (w.dirty && !anyb && w.dirty[blk] == 0) return;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int cc = wave * 16 + i;
        if (cc < ncols) O[(long)(cb0 + cc) * n + k0 + lane] = g[i];
    }
}

// ------------------------------------------------------------------------------------------------ trailing (fused)
// One workgroup per (matrix, block of 64 trailing columns): gathers the NB pivot rows of the block, solves
// U12 = L11^-1 A12 in LDS (one wave, one column per lane, L11 staged in LDS), writes the final U rows back,
// then sweeps every 64-row tile of live rows with an LDS-tiled, register-blocked rank-NB update:
//   256 threads = 16 x 16, each owning a 4 x 4 tile (rows tx+16i, columns ty+16j); per k the lane reads 4 multipliers
//   and 4 U entries from LDS (conflict-free b64 reads) for 16 updates -> VALU-bound, ~100 VGPRs, 3 workgroups per CU.
// No Ubuf round trip, no per-lane operand broadcast; the matrix is touched in coalesced column segments only
// (the pivot-row gather/scatter is the one strided access: NB x 64 elements per workgroup).
template <int NB, int MAXROWS, int CJ>
__global__ __launch_bounds__(256) void lu_trail_kernel(LuWs w, int k0, int nsys, int ncb, int climit, int l11off, int nsplit) {
    // XCD-aware 1-D grid: workgroup ids are dealt round-robin to the 8 XCDs, so the column blocks of one matrix are
    // given ids with equal (id & 7) and adjacent (id >> 3): they run on one XCD at about the same time and share the
    // multiplier panel L21 (read by every column block) through that XCD's L2 instead of re-reading it from HBM.
    // nsplit > 1: the row tiles of one (matrix, column block) are dealt to nsplit workgroups -- with one column block per matrix
    // (the narrow update inside a super-panel) a launch would otherwise be one workgroup per matrix, a fraction of the chip.
    // Every workgroup of a split repeats the (small) pivot-row gather and U12 solve; split 0 writes U12 to the factors.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int split = slot % nsplit, cbi = (slot / nsplit) % ncb, mi = (slot / (nsplit * ncb)) * 8 + xcd;
    if (mi >= (w.cnt ? ldc(w.cnt) : nsys)) return;
    const int b = w.idx[mi];
    if (w.info[b] != 0) return;
    const int n = w.n;
    double* __restrict__ A = w.mats + (long)b * w.mstride;
    const int* __restrict__ live = w.live + (long)b * n;
    const int* __restrict__ prow = w.prow + (long)b * n + k0;
    const double* __restrict__ l11 = w.l11 + (long)b * L11_STRIDE + l11off;  // diagonal block of this (sub-)panel

    const int mrem = n - k0 - NB;  // live rows after this panel (> 0)
    const int cb0 = k0 + NB + cbi * 64;
    const int ncols = (climit - cb0) < 64 ? (climit - cb0) : 64;  // climit = n, or the end of the super-panel for the narrow update
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);

    constexpr int LD = 66;  // padded row (doubles): rows stay 16-byte aligned, column reads conflict-free
    __shared__ __align__(16) double Us[NB][LD];
    __shared__ __align__(16) double Ls[2][NB][LD];
    __shared__ unsigned short s_live[MAXROWS];  // row indices < 65536
    __shared__ int s_anyzero;
    __shared__ unsigned s_kmask;  // slow path: bit k set = pivot row k has a non-zero entry in this column block
    __shared__ unsigned long long s_cmask;  // slow path: bit c set = column c of the block has a non-zero entry of U12
    __shared__ int s_nz[4];       // per wave: a non-zero entry among the pivot-row entries it gathered

    for (int i = t; i < mrem; i += 256) s_live[i] = (unsigned short)live[i];
    // 1. gather the pivot rows of this column block (wave-uniform k per pass -> prow[k] is a scalar load)
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
    // transposed L11 into the (still idle) tile buffer: a triangular-solve step reads its multipliers as LDS broadcasts
    // instead of waiting on one scalar load per step
    for (int e = t; e < NB * NB; e += 256) Ls[0][e / NB][e % NB] = l11[(e / NB) * w.l11ld + (e % NB)];
    __syncthreads();
    if ((s_nz[0] | s_nz[1] | s_nz[2] | s_nz[3]) == 0) {
        // the pivot rows are zero across this whole column block (banded matrices, off the band): the triangular solve
        // leaves them as they are (a_kj == 0: column untouched, dense.rs:148) and nothing is subtracted from the rows below
        double* __restrict__ O = w.out + (long)b * w.ostride;
        if (split == 0) {
#pragma unroll 4
            for (int i = 0; i < 16; ++i) {
                const int cc = wave * 16 + i;
                if (cc < ncols && lane < NB) O[(long)(cb0 + cc) * n + k0 + lane] = Us[lane][cc];
            }
        }
        return;
    }

    // 3. rank-NB update of every 64-row tile of live rows, software-pipelined: while tile rt is computed from LDS the
    //    multipliers and the C tile of tile rt+1 are already in flight; one raw barrier per tile (LDS visibility only --
    //    a __syncthreads() would also drain the tile stores, which nobody in this launch reads).
    const int tx = t & 15, ty = t >> 4;
    const int ntiles = (mrem + 63) >> 6;
    int col[CJ];  // CJ = 4: a full block of 64 columns; CJ = 2: at most 32 (the narrow update inside a super-panel)
    bool cok[CJ];
#pragma unroll
    for (int j = 0; j < CJ; ++j) {
        const int cj = ty + 16 * j;
        cok[j] = cj < ncols;
        col[j] = cb0 + (cok[j] ? cj : 0);
    }
    constexpr int LPT = NB / 4;  // multipliers per thread per tile
    double lreg[LPT], creg[4][CJ];
    int crow[4];
    bool rok[4];

    auto load_tile = [&](int rt, double (&lr_)[LPT], double (&cr_)[4][CJ], int (&crow_)[4], bool (&rok_)[4]) {
        const int lr = rt * 64 + lane;
        const int lrow = s_live[lr < mrem ? lr : mrem - 1];
#pragma unroll
        for (int i = 0; i < LPT; ++i) lr_[i] = A[(long)(k0 + wave * LPT + i) * n + lrow];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ri = rt * 64 + tx + 16 * i;
            rok_[i] = ri < mrem;
            crow_[i] = s_live[rok_[i] ? ri : mrem - 1];
        }
#pragma unroll
        for (int j = 0; j < CJ; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) cr_[i][j] = A[(long)col[j] * n + crow_[i]];
    };

    // 2. U12 = L11^-1 A12 by wave 0, one column per lane, right-looking over the source row kk
    if (wave == 0) {
        double u[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) u[k] = Us[k][lane];
        const bool real = lane < ncols;
        bool anyz = false;
#pragma unroll
        for (int kk = 0; kk < NB; ++kk) {
            const double ukk = u[kk];
            const bool z = real && (ukk == 0.0);
            anyz = anyz || z;
            if (__ballot(z) == 0ull) {
#pragma unroll
                for (int k = 0; k < NB; ++k)
                    if (k > kk) u[k] -= ukk * Ls[0][kk][k];  // a(i,j) -= a_kj * a_ik, ascending kk
            } else {
#pragma unroll
                for (int k = 0; k < NB; ++k)
                    if (k > kk) {
                        const double tn = u[k] - ukk * Ls[0][kk][k];
                        u[k] = z ? u[k] : tn;  // dense.rs:148: a_kj == 0 -> column untouched
                    }
            }
        }
#pragma unroll
        for (int k = 0; k < NB; ++k) Us[k][lane] = u[k];
        if (__ballot(anyz) != 0ull) {
            // dense.rs:148 skips the whole row update when a_kj == 0: a pivot row that is zero across this column block
            // contributes nothing here -- banded Jacobians (heat equation) have almost only such rows
            unsigned km = 0u;
            bool cnz = false;
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const bool e = real && u[k] != 0.0;
                cnz = cnz || e;
                km |= (__ballot(e) != 0ull) ? (1u << k) : 0u;
            }
            const unsigned long long cm = __ballot(cnz);
            if (lane == 0) {
                s_anyzero = 1;
                s_kmask = km;
                s_cmask = cm;
            }
        }
        load_tile(split, lreg, creg, crow, rok);
    } else {
        load_tile(split, lreg, creg, crow, rok);  // waves 1-3: first tile in flight while wave 0 solves for U12
    }
    __syncthreads();
    const bool slow = s_anyzero != 0;
    const unsigned kmask = slow ? (unsigned)__builtin_amdgcn_readfirstlane((int)s_kmask) : 0xffffffffu;
    // The solved pivot rows are final and nothing reads them in the work matrix again: they go straight to their place in
    // the factors -- pivot k of this panel is row k0 + k of the reference layout, so a column's NB entries are one
    // contiguous store (one row per lane) instead of NB eight-byte stores into NB different sectors of the work matrix;
    // lu_finalize_kernel skips this region.
    if (split == 0) {
        double* __restrict__ O = w.out + (long)b * w.ostride;
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const int cc = wave * 16 + i;
            if (cc < ncols && lane < NB) O[(long)(cb0 + cc) * n + k0 + lane] = Us[lane][cc];
        }
    }

    if (kmask == 0u) return;  // (uniform over the workgroup) nothing to subtract anywhere in this column block
    if (MAXROWS > 1024 && slow) {
        // Large n, U12 nearly empty (a banded matrix): one live row per thread, only the columns and pivot rows with a
        // non-zero entry are touched (see lu_trail64w_kernel). Same arithmetic per entry: ascending k, a_kj == 0 skipped.
        const unsigned long long cmask = s_cmask;
        if (__popcll(cmask) <= LU_SPARSE_COLS) {
            for (int ri = split * 256 + t; ri < mrem; ri += 256 * nsplit) {
                const int row = s_live[ri];
                for (unsigned long long cmm = cmask; cmm != 0ull; cmm &= cmm - 1ull) {
                    const int c = __builtin_ctzll(cmm);
                    double v = A[(long)(cb0 + c) * n + row];
                    for (unsigned mk = kmask; mk != 0u; mk &= mk - 1u) {
                        const int kk = __builtin_ctz(mk);
                        const double u = Us[kk][c];
                        if (u != 0.0) v = v - u * A[(long)(k0 + kk) * n + row];
                    }
                    A[(long)(cb0 + c) * n + row] = v;
                }
            }
            return;
        }
    }
#pragma unroll 1
    for (int rt = split, it = 0; rt < ntiles; rt += nsplit, ++it) {
        const int buf = it & 1;
#pragma unroll
        for (int i = 0; i < LPT; ++i) Ls[buf][wave * LPT + i][lane] = lreg[i];
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        double c[4][CJ];
        int srow[4];
        bool sok[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            srow[i] = crow[i];
            sok[i] = rok[i];
#pragma unroll
            for (int j = 0; j < CJ; ++j) c[i][j] = creg[i][j];
        }
        if (rt + nsplit < ntiles) load_tile(rt + nsplit, lreg, creg, crow, rok);  // in flight behind the arithmetic below
        if (!slow) {
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                double lv[4], uv[CJ];
#pragma unroll
                for (int i = 0; i < 4; ++i) lv[i] = Ls[buf][k][tx + 16 * i];
#pragma unroll
                for (int j = 0; j < CJ; ++j) uv[j] = Us[k][ty + 16 * j];
#pragma unroll
                for (int j = 0; j < CJ; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) c[i][j] -= uv[j] * lv[i];  // dense.rs:151, unfused
            }
        } else {
            for (unsigned mk = kmask; mk != 0u; mk &= mk - 1u) {  // ascending k, rows that are zero across the block skipped
                const int k = __builtin_ctz(mk);
                double lv[4], uv[CJ];
#pragma unroll
                for (int i = 0; i < 4; ++i) lv[i] = Ls[buf][k][tx + 16 * i];
#pragma unroll
                for (int j = 0; j < CJ; ++j) uv[j] = Us[k][ty + 16 * j];
#pragma unroll
                for (int j = 0; j < CJ; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const double tn = c[i][j] - uv[j] * lv[i];
                        c[i][j] = (uv[j] != 0.0) ? tn : c[i][j];  // dense.rs:148
                    }
            }
        }
#pragma unroll
        for (int j = 0; j < CJ; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (cok[j] && sok[i]) A[(long)col[j] * n + srow[i]] = c[i][j];
    }
}

// ------------------------------------------------------------------------------------------------ trailing, 64-wide super-panels
// Same contract as lu_trail_kernel for super-panels of 64 columns: half as many sweeps over the trailing matrix, so about
// half its HBM traffic. One workgroup per (matrix, 64 trailing columns):
//   1. gather the 64 pivot rows of the block into LDS;
//   2. U12 = L11^-1 A12 in three stages so that only two short chains are serial: wave 0 solves rows 0..31, all four
//      waves apply those rows to rows 32..63 (8 rows per wave), wave 0 solves rows 32..63. Every element still receives
//      its updates in ascending pivot order; L11 is staged in LDS (one scalar load per solve step was 4x slower);
//   3. rank-64 update of the live rows in wave-private strips: a wave owns 16-row strips of the block and stages the
//      multipliers of ITS rows in its own 4 KB of LDS, so the update loop has no workgroup barrier at all -- waves only
//      meet in the U12 prologue. Per lane a 4 x 4 register tile (rows a + 4i of the strip, columns q + 16j); L and U are
//      stored in LDS with the lane's four rows / four columns adjacent, so a k-step is four ds_read_b128 for 32 VALU
//      ops; two k-chunks of 32 through the strip buffer, the next strip's operands in flight behind the second chunk.
//      49 KB of LDS per workgroup -> three workgroups per CU.
template <int MAXROWS>
__global__ __launch_bounds__(256, MAXROWS <= 1024 ? 3 : 2) void lu_trail64w_kernel(LuWs w, int k0, int nsys, int ncb, int nsplit_arg) {
    constexpr int NB = 64, KC = 32;
    constexpr bool NOPRO = tb::NOPRO;  // timing build (exp_switches.hpp): no gather, no U12 solve (U12 = constants); false in the product
    const int nsplit = MAXROWS > 1024 ? nsplit_arg : 1;  // the row split exists for large n only
    // workgroups of one matrix: its ncb column blocks, then (nsplit > 1) nsplit - 1 helpers for each of the first
    // LU_SPLIT_BLOCKS column blocks
    const int nbs = nsplit > 1 ? (ncb < LU_SPLIT_BLOCKS ? ncb : LU_SPLIT_BLOCKS) : 0;
    const int hs = nsplit > 1 ? nsplit - 1 : 1;  // helpers per split column block
    const int wpm = ncb + hs * nbs;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int sm = slot % wpm, mi = (slot / wpm) * 8 + xcd;
    const int cbi = sm < ncb ? sm : (sm - ncb) / hs;
    const int split = sm < ncb ? 0 : 1 + (sm - ncb) % hs;
    if (mi >= (w.cnt ? ldc(w.cnt) : nsys)) return;
    const int b = w.idx[mi];
    if (w.info[b] != 0) return;
    // nsplit > 1: the strips of live rows of one (matrix, column block) are dealt to nsplit workgroups when the matrix has
    // shown itself banded -- at most a quarter of the column blocks of the first super-panel's update (k0 == 0, counted in
    // nzb by lu_u12_zero_kernel) had a non-zero pivot-row entry -- and the block is one of the first few, next to the panel, where a band has its
    // work. Every workgroup of a split repeats the gather and the U12 solve; split 0 writes U12 to the factors. Otherwise
    // (dense matrices: every column block has work and the launch fills the chip by itself) split 0 does it all and the
    // helpers leave. All workgroups of a launch read the same, already final, count.
    if (MAXROWS > 1024 && ldc(w.bz + b * 64 + cbi) == 0) return;  // zero block: lu_u12_zero_kernel has dealt with it
    int nsp = 1;
    if (nsplit > 1) {
        const int nzb = ldc(w.nzb + b);
        const bool banded = k0 > 0 && nzb * 4 <= (w.n - 1) / 64;
        int eff = nsplit;  // helpers in use: as many as keep the launch at two workgroups per CU (lu_driver.hpp)
        if (w.cnt) {
            const int cntv = ldc(w.cnt);
            while (eff > 1 && cntv * eff > 512) eff >>= 1;
        }
        if (banded && cbi < nzb && cbi < nbs && split < eff) nsp = eff;
        else if (split != 0) return;
    }
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
    unsigned long long* tst = (tb::STAMPS >= 0 && k0 == tb::STAMPS && w.stamps) ? w.stamps + (size_t)blockIdx.x * 8 : nullptr;
#define TSTAMP(i) do { if (tb::STAMPS >= 0 && tst && t == 0) tst[i] = wall_clock64(); } while (0)
    TSTAMP(0);

    // U12, columns permuted by pl. Rows padded by two doubles (still 16-byte aligned for ds_read_b128): the column reads of
    // the store to the factors below (lane = row) then spread over 8 bank groups instead of hitting one
    __shared__ __align__(16) double Us[NB][64 + tb::US_PAD];
    __shared__ __align__(16) double Ls[KC][64];      // prologue: L11 staging; update loop: 4 wave-private [KC][16] strips
    __shared__ unsigned short s_live[MAXROWS];
    __shared__ int s_anyzero;
    __shared__ int s_nz[4];          // per wave: a non-zero entry among the pivot-row entries it gathered
    __shared__ unsigned s_kmask[2];  // slow path: bit k of word R0 / 32 set = pivot row R0 + k has a non-zero entry in this column block
    __shared__ unsigned long long s_cmask[2];  // slow path: bit c set = column c of the block has a non-zero entry among pivot rows R0 .. R0 + 31

    if (MAXROWS <= 1024)
        for (int i = t; i < mrem; i += 256) s_live[i] = (unsigned short)live[i];
    const int s0 = wave + 4 * split, sstep = 4 * nsp;  // this wave's strips: s0, s0 + sstep, ...

    // ---- this lane's share of a strip: rows a + 4i, columns q + 16j
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


    // ---- FAST prologue (n <= 1024): U12 = L11^-1 A12 solved in registers by all four waves at once, no branch and no
    // barrier inside the solve. Wave w owns columns 16w .. 16w+15 of the block, FOUR lanes per column: lane `part` of a
    // quad holds rows k = 4i + part (i = 0..15) of its column. Step kk: the quad lane that owns row kk broadcasts it on
    // the DPP crossbar (quad_perm), every lane updates its rows k > kk: a(k,j) -= a(kk,j) * l(k,kk), ascending kk, unfused
    // (dense.rs:142-154) -- 504 updates per lane instead of 2016 on one lane of one wave. L11 (transposed, as lu_wavepanel
    // left it) is staged once, whole, in the memory of Us -- U12 is not needed there before the solve has finished -- in a
    // layout that gives a lane its rows as consecutive doubles (ds_read_b128, conflict-free across the quad).
    // The path assumes that no pivot-row entry a(kk,j) is an exact zero (dense.rs:148 would leave such a column untouched,
    // which differs from subtracting 0 * l in the sign of a zero and for a non-finite l); it checks every such entry on the
    // way and, if one was zero, nothing has been stored: the prologue below starts over with the rule applied per entry.
    __shared__ int s_fz[4];
    bool fast_ok = false;
    if constexpr (MAXROWS <= 1024 && tb::TRAIL_QUAD) {
      if (!NOPRO) {
        const int part = lane & 3, qc = wave * 16 + (lane >> 2);
        const bool real = qc < ncols;
        double l11r[16], u[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) l11r[i] = l11[i * 256 + t];
        {
            const double* __restrict__ colp = A + (long)(cb0 + (real ? qc : 0)) * n;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int p0 = ldc(prow + 4 * i), p1 = ldc(prow + 4 * i + 1), p2 = ldc(prow + 4 * i + 2), p3 = ldc(prow + 4 * i + 3);
                const int pr = part == 0 ? p0 : part == 1 ? p1 : part == 2 ? p2 : p3;
                if constexpr (tb::NOGATHER) u[i] = 1.0e-3 * (double)(1 + ((i + pr) & 7));  // timing build
                else u[i] = colp[pr];
            }
        }
        double* __restrict__ Lq = &Us[0][0];  // [64][66]: row kk, entry of pivot row k at (k & 3) * 16 + (k >> 2) + 2 * ((k & 3) >> 1)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int e = i * 256 + t, kk = e >> 6, k = e & 63;
            Lq[kk * 66 + (k & 3) * 16 + (k >> 2) + 2 * ((k & 3) >> 1)] = l11r[i];
        }
        if (t == 0) s_anyzero = 0;
        lds_barrier();  // L11 and the live list are in LDS
        TSTAMP(1);
        if (s0 < nstrips) {  // the first strip of every wave: in flight behind the solve
            load_C(s0);
            load_L(s0, 0);
        }
        const double* __restrict__ Lp = Lq + part * 16 + 2 * (part >> 1);
        bool anyz = false;
        static_for<0, 64>([&](auto kt) {
            constexpr int kk = decltype(kt)::value;
            constexpr int p0 = kk & 3, i0 = kk >> 2;
            const double ukk = dpp_mov_f64<p0 * 0x55, 0xf>(u[i0]);  // quad_perm: [p0, p0, p0, p0]
            anyz = anyz || (ukk == 0.0);
            if constexpr (p0 < 3) {  // row 4 i0 + part is below row kk for the quad's lanes part > p0 only
                const double tn = upd(u[i0], ukk, Lp[kk * 66 + i0]);
                u[i0] = (part > p0) ? tn : u[i0];
            }
            static_for<i0 + 1, 16>([&](auto it) {
                constexpr int i = decltype(it)::value;
                u[i] = upd(u[i], ukk, Lp[kk * 66 + i]);
            });
        });
        TSTAMP(2);
        const unsigned long long zb = __ballot(anyz && real);
        if (lane == 0) s_fz[wave] = zb != 0ull ? 1 : 0;
        lds_barrier();  // every wave has finished reading L11 from the memory of Us
        fast_ok = (s_fz[0] | s_fz[1] | s_fz[2] | s_fz[3]) == 0;
        if (fast_ok) {
            const int qpl = 4 * (qc & 15) + (qc >> 4);
#pragma unroll
            for (int i = 0; i < 16; ++i) Us[4 * i + part][qpl] = real ? u[i] : 0.0;
        }
        lds_barrier();  // U12 is in Us (or nothing was written and the prologue below starts from the gather)
        TSTAMP(3);
      }
    }
    if (!fast_ok) {
    bool nz = NOPRO;
    if (NOPRO) {
#pragma unroll
        for (int pass = 0; pass < NB / 4; ++pass) Us[pass * 4 + wave][pl] = 1.0e-4 * (double)(1 + ((pass + lane) & 7));
        if (t == 0) { s_kmask[0] = 0xffffffffu; s_kmask[1] = 0xffffffffu; s_cmask[0] = ~0ull; s_cmask[1] = ~0ull; }
    } else if (MAXROWS > 1024) {
        // large n: one pivot row per lane, a column per load. Pivot rows that are neighbours in memory (a banded matrix that
        // pivots little or not at all) then share cache lines -- 4 lines per load instead of 64 sectors; most workgroups of
        // a banded matrix do nothing but this gather (their block turns out zero), and it bounds the launch.
        const int pr = prow[lane];
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const int cc = wave * 16 + i;
            const double g = (cc < ncols) ? A[(long)(cb0 + cc) * n + pr] : 0.0;
            nz = nz || (g != 0.0);
            Us[lane][4 * (cc & 15) + (cc >> 4)] = g;
        }
    } else {
#pragma unroll
        for (int pass = 0; pass < NB / 4; ++pass) {
            const int k = pass * 4 + wave;
            const int pr = ldc(prow + k);
            const double g = (lane < ncols) ? A[(long)(cb0 + lane) * n + pr] : 0.0;
            nz = nz || (g != 0.0);
            Us[k][pl] = g;
        }
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
    if (MAXROWS <= 1024 && !NOPRO) stage_l11(0);
    lds_barrier();
    if ((s_nz[0] | s_nz[1] | s_nz[2] | s_nz[3]) == 0) {
        // the 64 pivot rows are zero across this whole column block (banded matrices, off the band): the triangular solve
        // leaves them as they are (a_kj == 0: column untouched, dense.rs:148) and nothing is subtracted from the rows below
        if (split != 0) return;
        double* __restrict__ O = w.out + (long)b * w.ostride;
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const int cc = wave * 16 + i;
            if (cc < ncols) O[(long)(cb0 + cc) * n + k0 + lane] = Us[lane][4 * (cc & 15) + (cc >> 4)];
        }
        return;
    }
    if (MAXROWS > 1024) {  // large n: most column blocks of a banded matrix have left by now, without having read these
        for (int i = t; i < mrem; i += 256) s_live[i] = (unsigned short)live[i];
        stage_l11(0);
        lds_barrier();
    }
    // one triangular stage by wave 0: rows [R0, R0+32) of U12 against the diagonal block of L11 they share
    auto trsm32 = [&](const int R0) {
        double u[KC];
#pragma unroll
        for (int k = 0; k < KC; ++k) u[k] = Us[R0 + k][pl];
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
                    if (k > kk) u[k] = upd(u[k], ukk, Ls[kk][R0 + k]);  // a(i,j) -= a_kj * a_ik, ascending kk
            } else if (MAXROWS > 1024 && __ballot(real && !z) == 0ull) {
                // pivot row kk is zero across the whole block (banded matrices): every column is left as it is
            } else {
#pragma unroll
                for (int k = 0; k < KC; ++k)
                    if (k > kk) {
                        const double tn = upd(u[k], ukk, Ls[kk][R0 + k]);
                        u[k] = z ? u[k] : tn;  // dense.rs:148: a_kj == 0 -> column untouched
                    }
            }
        }
#pragma unroll
        for (int k = 0; k < KC; ++k) Us[R0 + k][pl] = u[k];
        // dense.rs:148 skips the whole row update when a_kj == 0: a pivot row that is zero across this column block
        // contributes nothing to it -- banded Jacobians (heat equation) have almost only such rows. (The mask is only read
        // on the select path, i.e. when some entry of the block is an exact zero.)
        unsigned km = 0xffffffffu;
        unsigned long long cm = ~0ull;
        if (__ballot(anyz) != 0ull || s_anyzero != 0) {
            km = 0u;
            bool cnz = false;
#pragma unroll
            for (int k = 0; k < KC; ++k) {
                const bool e = real && u[k] != 0.0;
                cnz = cnz || e;
                km |= (__ballot(e) != 0ull) ? (1u << k) : 0u;
            }
            cm = __ballot(cnz);
        }
        if (lane == 0) {
            s_kmask[R0 / KC] = km;
            s_cmask[R0 / KC] = cm;
            if (__ballot(anyz) != 0ull) s_anyzero = 1;
        }
    };

    if (wave == 0 && !NOPRO) trsm32(0);
    else if (s0 < nstrips) {  // live rows only (untouched by the U12 stores), in flight behind the solves
        load_C(s0);
        load_L(s0, 0);
    }
    lds_barrier();
    if (!NOPRO) {   // rows 32..63 receive the updates of pivot rows 0..31: 8 rows per wave, one column per lane
        const bool zpath = s_anyzero != 0;
        double v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = Us[KC + wave * 8 + i][pl];
        if (!zpath) {
#pragma unroll
            for (int kk = 0; kk < KC; ++kk) {
                const double ut = Us[kk][pl];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = upd(v[i], ut, Ls[kk][KC + wave * 8 + i]);
            }
        } else {
#pragma unroll 4
            for (int kk = 0; kk < KC; ++kk) {
                const double ut = Us[kk][pl];
                if (MAXROWS > 1024 && __ballot(ut != 0.0) == 0ull) continue;  // zero pivot row: nothing to subtract
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const double tn = upd(v[i], ut, Ls[kk][KC + wave * 8 + i]);
                    v[i] = (ut != 0.0) ? tn : v[i];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) Us[KC + wave * 8 + i][pl] = v[i];
    }
    if (!NOPRO) {
    lds_barrier();
    stage_l11(KC);
    lds_barrier();
    if (wave == 0) {
        trsm32(KC);
        if (s0 < nstrips) {
            load_C(s0);
            load_L(s0, 0);
        }
    }
    }
    lds_barrier();  // last workgroup barrier: from here on a wave touches only Us (read-only) and its own strip of Ls
    }  // !fast_ok
    const bool slow = s_anyzero != 0;
    // The solved pivot rows are final and nothing reads them in the work matrix again: they go straight to their place in
    // the factors -- pivot k of this super-panel is row k0 + k of the reference layout, so a column's 64 entries are one
    // contiguous 512-byte store (one row per lane) instead of 64 eight-byte stores into 64 different sectors of the work
    // matrix; lu_finalize_kernel skips this region.
    if (split == 0) {
        double* __restrict__ O = w.out + (long)b * w.ostride;
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const int cc = wave * 16 + i;  // column of the block; its LDS slot is 4 * (cc & 15) + (cc >> 4)
            if (cc < ncols) O[(long)(cb0 + cc) * n + k0 + lane] = Us[lane][4 * (cc & 15) + (cc >> 4)];
        }
    }
    if constexpr (tb::NOUPD) return;  // timing build: prologue only
    TSTAMP(4);
    const unsigned kmask0 = (unsigned)__builtin_amdgcn_readfirstlane((int)s_kmask[0]);
    const unsigned kmask1 = (unsigned)__builtin_amdgcn_readfirstlane((int)s_kmask[1]);
    if (slow && (kmask0 | kmask1) == 0u) return;  // (uniform over the workgroup) U12 of this block is all zeros: nothing to subtract
    if (MAXROWS > 1024 && slow) {
        // Large n, U12 nearly empty (a banded matrix: a few columns next to the panel, a few pivot rows): one live row per
        // thread, and only the columns and pivot rows that have a non-zero entry are touched -- the strips below would read
        // and write back all 64 columns of every live row to change one of them. Same arithmetic per entry: ascending k,
        // a_kj == 0 skipped (dense.rs:148-151).
        const unsigned long long cmask = s_cmask[0] | s_cmask[1];
        if (__popcll(cmask) <= LU_SPARSE_COLS) {
            for (int ri = split * 256 + t; ri < mrem; ri += 256 * nsp) {
                const int row = s_live[ri];
                for (unsigned long long cmm = cmask; cmm != 0ull; cmm &= cmm - 1ull) {
                    const int c = __builtin_ctzll(cmm);
                    const int cs = 4 * (c & 15) + (c >> 4);
                    double v = A[(long)(cb0 + c) * n + row];
                    for (unsigned mk = kmask0; mk != 0u; mk &= mk - 1u) {
                        const int kk = __builtin_ctz(mk);
                        const double u = Us[kk][cs];
                        if (u != 0.0) v = upd(v, u, A[(long)(k0 + kk) * n + row]);
                    }
                    for (unsigned mk = kmask1; mk != 0u; mk &= mk - 1u) {
                        const int kk = KC + __builtin_ctz(mk);
                        const double u = Us[kk][cs];
                        if (u != 0.0) v = upd(v, u, A[(long)(k0 + kk) * n + row]);
                    }
                    A[(long)(cb0 + c) * n + row] = v;
                }
            }
            return;
        }
    }
    double (*__restrict__ Lw)[16] = reinterpret_cast<double (*)[16]>(&Ls[0][0] + wave * (KC * 16));

    auto chunk = [&](double (&c)[4][4], const int kbase) {
        if (!slow) {
            // software-pipelined by hand: the operands of step k + 1 are requested from LDS before the arithmetic of step k
            // is issued (the compiler places each step's four ds_read_b128 right in front of their first use, so a wave stalls
            // for the LDS latency at every step; 16 more VGPRs buy that back)
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
                    for (int i = 0; i < 4; ++i) c[i][j] = upd(c[i][j], uv[j], lv[i]);  // dense.rs:151
            };
            if constexpr (tb::TRAIL_PIPE) {
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
#pragma unroll
                for (int k = 0; k < KC; ++k) {
                    rd(k, lvA, uvA);
                    mac(lvA, uvA);
                }
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
                        const double tn = upd(c[i][j], uv[j], lv[i]);
                        c[i][j] = (uv[j] != 0.0) ? tn : c[i][j];  // dense.rs:148
                    }
            }
        }
    };

#pragma unroll 1
    for (int s = s0; s < nstrips; s += sstep) {
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
        if constexpr (tb::STAMPS >= 0) {
            if (s == s0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); TSTAMP(5); }
        }
        chunk(c, 0);
        if constexpr (tb::STAMPS >= 0) {
            if (s == s0) TSTAMP(6);
        }
#pragma unroll
        for (int i = 0; i < LPT; ++i) Lw[4 * i + kq][lslot] = lreg[i];  // same wave, program order: chunk 0's reads are done
        if (s + sstep < nstrips) {  // next strip in flight behind the second chunk
            load_C(s + sstep);
            load_L(s + sstep, 0);
        }
        chunk(c, KC);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (cok[j] && sok[i]) A[coff[j] + srow[i]] = c[i][j];
    }
    TSTAMP(7);
#undef TSTAMP
}

// ------------------------------------------------------------------------------------------------ finalize
// out(pos[r], j) = work(r, j); perm[pos[r]] = r. One workgroup per (matrix, column group).
// sp = panel width of the pipeline (wp_rows: see below): its trailing kernels already wrote the U rows right of their panel into `out`; those
// entries are skipped here -- the work matrix no longer holds them (sp_lead: see below).
__global__ __launch_bounds__(256) void lu_finalize_kernel(LuWs w, double* __restrict__ out, long ostride, int* __restrict__ perm,
                                                          int cols_per_block, int sp, int sp_lead, int wp_rows) {
    if (w.cnt && (int)blockIdx.x >= *w.cnt) return;
    const int b = w.idx[blockIdx.x];
    if (w.info[b] != 0) return;
    const int n = w.n;
    double* __restrict__ A = w.mats + (long)b * w.mstride;  // (read; with jwzero also cleared)
    double* __restrict__ O = out + (long)b * ostride;
    const int* __restrict__ pos = w.pos + (long)b * n;
    const int jbeg = blockIdx.y * cols_per_block;
    const bool clearw = w.jwzero != nullptr && w.dirty != nullptr;
    if (clearw && blockIdx.y == 0 && threadIdx.x == 0) w.jwzero[b] = 1;  // (read by kernels launched after this one)
    const int jend = (jbeg + cols_per_block < n) ? jbeg + cols_per_block : n;
    for (int r = threadIdx.x; r < n; r += blockDim.x) {
        const int p = pos[r];
        if (blockIdx.y == 0 && perm) perm[(long)b * n + p] = r;
        int je = jend;
        if (sp > 0) {
            // large n: the leading 64-column super-panels (more than LU_MAX_N live rows) are built from sp_lead-column panels
            // wp_rows > 0: super-panels with at most wp_rows live rows were factored 64 columns at a time (lu_wavepanel_kernel)
            const int liv = n - (p & ~63);  // live rows when the super-panel of pivot position p was factored
            // (sp_lead == 64: lu_superpanel_kernel factored those super-panels whole and their U rows inside the super-panel are in the work matrix)
            const int w_ = (wp_rows > 0 && liv <= wp_rows) ? 64 : (sp_lead > 0 && liv > LU_MAX_N) ? (sp_lead >= 64 ? 64 : liv > LU_WIDE_ROWS ? sp_lead : 2 * sp_lead) : sp;
            const int spend = (p / w_ + 1) * w_;  // first column right of the panel that made row p a pivot row
            je = je < spend ? je : spend;
        }
        if (w.zmap && w.dirty) {
            // The same copy, noting which 64 x 64 blocks of the factors receive a non-zero (or a NaN) -- and without writing +0.0 over
            // +0.0: the ctx's factors start as zeros, `dirty` marks the blocks that have ever received a value with non-zero bits
            // (a -0.0 multiplier counts), and an all-zero-bits value bound for a block that never has is there already. For a banded
            // matrix in dense storage that is nearly every entry of L: the scatter reads the lower triangle and writes the band.
            unsigned char* __restrict__ zm = w.zmap + (long)b * 4096;
            unsigned char* __restrict__ dm = w.dirty + (long)b * 4096;
            // (a workgroup's columns lie in one 64-column block -- cols_per_block divides 64 --, so a row meets one block: one look
            // at the maps per row, not per entry)
            const int blk = (jbeg >> 6) * 64 + (p >> 6);
            const bool was_dirty = dm[blk] != 0;
            bool anybits = false, anynz = false;
            unsigned cmask = 0u;  // columns of this chunk (at most 32) in which the row had content
            for (int j0 = jbeg; j0 < je; j0 += 8) {  // eight columns' loads in flight per thread (the trip count differs from row to row: the compiler does not unroll it)
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = (j0 + u < je) ? A[(long)(j0 + u) * n + r] : 0.0;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const bool bits = __double_as_longlong(v[u]) != 0ll;  // (a column beyond je: +0.0, no bits, no store when the block is clean; and never stored: see the test on j)
                    anybits = anybits || bits;
                    anynz = anynz || (v[u] != 0.0);
                    if (j0 + u < je && (bits || was_dirty)) O[(long)(j0 + u) * n + p] = v[u];
                    if (bits && j0 + u < je) cmask |= 1u << (j0 + u - jbeg);
                }
            }
            if (anybits && !was_dirty) dm[blk] = 1;
            if (anynz) zm[blk] = 1;
            // jwzero: what was read has been read for the last time -- cleared after the loop (stores into the work matrix inside it would
            // order every group of loads behind the stores of the group before)
            if (clearw)
                for (unsigned m = cmask; m != 0u; m &= m - 1u) A[(long)(jbeg + __builtin_ctz(m)) * n + r] = 0.0;
            // The rest of the work matrix's content: a pivot row's entries right of its super-panel (what U12 was solved from). They lie
            // in blocks that lu_u12_zero_kernel marked in `dirty` from these very values; cleared blind, a row of a marked block at a time.
            if (clearw && je <= jbeg && dm[blk] != 0)
                for (int j = jbeg; j < jend; ++j) A[(long)j * n + r] = 0.0;
        } else if (w.zmap) {  // the same copy, noting which 64 x 64 blocks of the factors receive a non-zero (or a NaN)
            unsigned char* __restrict__ zm = w.zmap + (long)b * 4096;
            for (int j = jbeg; j < je; ++j) {
                const double v = A[(long)j * n + r];
                O[(long)j * n + p] = v;
                if (v != 0.0) zm[(j >> 6) * 64 + (p >> 6)] = 1;
            }
        } else {
            for (int j = jbeg; j < je; ++j) O[(long)j * n + p] = A[(long)j * n + r];
        }
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

}  // namespace idahip
