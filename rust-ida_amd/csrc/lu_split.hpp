// The rank-64 trailing update of the batched getrf as TWO kernels (n <= 1024, LuWs::u12 != null): dense.rs:139-155.
//
// lu_trail64w_kernel does the U12 solve and the update in one workgroup: 168 VGPRs and 52 KB of LDS, three workgroups per
// CU, all of one launch the same length -- they march in step, and the ~45 us a workgroup spends outside its update loop
// (dispatch, dependent loads, the solve, the drain) are hidden by nothing. Here the two parts are
// separate launches with the resources each one needs:
//
//   lu_u12_kernel      per (matrix, 64 columns): the FAST prologue of lu_trail64w_kernel alone -- pivot rows to registers in
//                      the quad layout, L11 staged once in LDS, 64 steps on the DPP crossbar -- 34 KB of LDS. U12 goes to
//                      the factors (as before) and, k-major, to LuWs::u12: row k of the block is 64 consecutive doubles.
//                      A block with an exact zero among its pivot-row entries (dense.rs:148) is flagged and left to
//                      lu_trail64w_kernel's per-entry path (launched for the flagged blocks only; none in a dense batch).
//   lu_update16_kernel per (matrix, 64 columns), a wave per 16 columns, NO LDS and no barrier: a lane owns one live row of
//                      a 64-row strip and keeps its 16 entries in registers; step k is one coalesced load of the strip's
//                      multipliers (column k0 + k of the work matrix, 8 in flight) and the 16 U entries of row k as
//                      SCALAR operands (s_load from LuWs::u12: they are wave-uniform), 16 unfused multiply-subtracts in
//                      ascending k. About 90 VGPRs: five waves per SIMD, and whatever else is on the device fits beside them.
//
// Same operands, same order, same instructions per element as lu_trail64w_kernel: the factors are bit-identical
// (tests/test_gpu_lsolver.py runs both pipelines against the oracle).
#pragma once
#include "lu_kernels.hpp"

namespace idahip {

constexpr int U12_BLOCKS = 16;          // column blocks per matrix that LuWs::u12 / u12f hold (n <= 1024: at most 15)
constexpr int U12_STRIDE = 64 * 64;     // doubles per block in LuWs::u12
typedef int v16i __attribute__((ext_vector_type(16)));  // 16 SGPRs: eight doubles of a row of U

__global__ __launch_bounds__(256, 4) void lu_u12_kernel(LuWs w, int k0, int nsys, int ncb) {
    constexpr int NB = 64;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int cbi = slot % ncb, mi = (slot / ncb) * 8 + xcd;
    if (mi >= (w.cnt ? ldc(w.cnt) : nsys)) return;
    const int b = w.idx[mi];
    if (w.info[b] != 0) return;
    const int n = w.n;
    double* __restrict__ A = w.mats + (long)b * w.mstride;
    const int* __restrict__ prow = w.prow + (long)b * n + k0;
    const double* __restrict__ l11 = w.l11 + (long)b * L11_STRIDE;  // l11ld == 64
    const int cb0 = k0 + NB + cbi * 64;
    const int ncols = (n - cb0) < 64 ? (n - cb0) : 64;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);

    __shared__ __align__(16) double Us[NB][66];  // first L11 (layout below), then U12 with its columns permuted by pl
    __shared__ int s_fz[4];

    // wave w owns columns 16w .. 16w+15, four lanes per column: lane `part` of a quad holds rows 4i + part (lu_trail64w_kernel)
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
            u[i] = colp[pr];
        }
    }
    double* __restrict__ Lq = &Us[0][0];  // [64][66]: row kk, entry of pivot row k at (k & 3) * 16 + (k >> 2) + 2 * ((k & 3) >> 1)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int e = i * 256 + t, kk = e >> 6, k = e & 63;
        Lq[kk * 66 + (k & 3) * 16 + (k >> 2) + 2 * ((k & 3) >> 1)] = l11r[i];
    }
    lds_barrier();
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
    const unsigned long long zb = __ballot(anyz && real);
    if (lane == 0) s_fz[wave] = zb != 0ull ? 1 : 0;
    lds_barrier();  // every wave has finished reading L11 from the memory of Us
    int* __restrict__ flag = w.u12f + (long)b * U12_BLOCKS + cbi;
    if ((s_fz[0] | s_fz[1] | s_fz[2] | s_fz[3]) != 0) {  // dense.rs:148 applies to some entry: nothing stored, the per-entry path takes the block
        if (t == 0) *flag = 2;
        return;
    }
    {
        const int qpl = 4 * (qc & 15) + (qc >> 4);
#pragma unroll
        for (int i = 0; i < 16; ++i) Us[4 * i + part][qpl] = real ? u[i] : 0.0;
    }
    lds_barrier();
    // the solved pivot rows: to their place in the factors (a column's 64 entries are one 512-byte store) ...
    double* __restrict__ O = w.out + (long)b * w.ostride;
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int cc = wave * 16 + i;
        if (cc < ncols) O[(long)(cb0 + cc) * n + k0 + lane] = Us[lane][4 * (cc & 15) + (cc >> 4)];
    }
    // ... and k-major for the update kernel's scalar loads (a row's 64 entries are one 512-byte store)
    double* __restrict__ S = w.u12 + ((long)b * U12_BLOCKS + cbi) * U12_STRIDE;
    const int pl = 4 * (lane & 15) + (lane >> 4);
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int k = wave * 16 + i;
        S[k * 64 + lane] = Us[k][pl];
    }
    if (t == 0) *flag = 1;
}

template <int DEPTH>
__global__ __launch_bounds__(256, 5) void lu_update16_kernel(LuWs w, int k0, int nsys, int ncb, int nstr) {
    // a workgroup per (matrix, 64 columns, 64 live rows): short workgroups of equal length -- whole column blocks per workgroup
    // (seven strips at n = 512) left a launch's last round of workgroups, a fifth of its time, on a nearly empty device
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int sidx = slot % nstr, cbi = (slot / nstr) % ncb, mi = (slot / (nstr * ncb)) * 8 + xcd;
    if (mi >= (w.cnt ? ldc(w.cnt) : nsys)) return;
    const int b = ldc(w.idx + mi);
    if (ldc(w.info + b) != 0) return;
    if (ldc(w.u12f + (long)b * U12_BLOCKS + cbi) != 1) return;  // flagged for (or already done by) lu_trail64w_kernel
    const int n = w.n;
    const int mrem = n - k0 - 64;  // live rows after this panel (> 0)
    const int cb0 = k0 + 64 + cbi * 64;
    const int ncols = (n - cb0) < 64 ? (n - cb0) : 64;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c0 = wave * 16;
    if (c0 >= ncols) return;  // (no barrier in this kernel)
    const int nc = (ncols - c0) < 16 ? (ncols - c0) : 16;
    double* __restrict__ A = w.mats + (long)b * w.mstride;
    static_assert((long)LU_MAX_N * LU_MAX_N * 8 < (1l << 31), "32-bit buffer offsets");
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(A, 0, n * n * 8, 0x00020000);
    const int* __restrict__ live = w.live + (long)b * n;
    const double* __restrict__ U = static_cast<const double*>(__builtin_assume_aligned(w.u12 + ((long)b * U12_BLOCKS + cbi) * U12_STRIDE + c0, 128));  // U[k * 64 + j]: row k, this wave's column j (hipMalloc'ed: 256-byte aligned)
    const int n8 = n * 8;
    const int lbase = k0 * n8;            // byte offset of the multipliers' first column
    const int cbase = (cb0 + c0) * n8;    // byte offset of this wave's first column

    // Scalar byte offsets advance by n8 along opaque chains (`+s` asm): as loop invariants of the strip loop the compiler would
    // keep all 64 + 16 of them, and the 16 column predicates, in SGPRs across the loop and spill them into vector lanes.
    // A lane without a row (rok false), or a column beyond the matrix (j >= nc), loads +0.0 and stores nothing: its per-lane
    // offset lies beyond the descriptor's range (the range check is on the vector offset).
    constexpr unsigned OOB = 0xfffffff0u;
    {
        const int s0 = 64 * sidx;  // (< mrem: nstr = ceil(mrem / 64))
        const int ri = s0 + lane;
        const bool rok = ri < mrem;
        const unsigned voff = 8u * (unsigned)live[rok ? ri : mrem - 1];
        int ncl = nc, so = cbase, slp = lbase;
        asm volatile("" : "+s"(ncl), "+s"(so), "+s"(slp));
        double c[16], l[DEPTH];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            c[j] = buf_load_f64(rsrc, j < ncl ? voff : OOB, so);
            so += n8;
            asm volatile("" : "+s"(so));
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            l[d] = buf_load_f64(rsrc, voff, slp);
            slp += n8;
            asm volatile("" : "+s"(slp));
        }
        // the 16 U entries of a row live in SGPRs, two rows at a time; the scheduler may not move anything across the fences
        // (left alone it hoists every scalar load of the strip to its top and spills the SGPRs into vector lanes)
        double ua[16], ub[16];
        auto rdu = [&](const int k, double (&uk)[16]) {
#pragma unroll
            for (int j = 0; j < 16; ++j) uk[j] = ldc(U + k * 64 + j);
        };
        rdu(0, ua);
        // a step: the first multiply-subtract of row k (the compiler's s_waitcnt lgkmcnt(0) in front of it then waits for row k
        // alone, requested a whole step ago -- scalar loads return out of order, so there is no counted wait for them), then the
        // requests for row k + 1 and for the multipliers of step k + DEPTH, then the other 15
        auto step = [&](auto kc, double (&uk)[16], double (&un)[16]) {
            constexpr int k = decltype(kc)::value;
            const double lk = l[k % DEPTH];
            __builtin_amdgcn_sched_barrier(0);
            c[0] = upd(c[0], uk[0], lk);  // dense.rs:151, ascending k
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (k + 1 < 64) rdu(k + 1, un);
            if constexpr (k + DEPTH < 64) {
                l[k % DEPTH] = buf_load_f64(rsrc, voff, slp);
                slp += n8;
                asm volatile("" : "+s"(slp));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 1; j < 16; ++j) c[j] = upd(c[j], uk[j], lk);
            __builtin_amdgcn_sched_barrier(0);
        };
        static_for<0, 32>([&](auto kt) {
            constexpr int k = 2 * decltype(kt)::value;
            step(std::integral_constant<int, k>{}, ua, ub);
            step(std::integral_constant<int, k + 1>{}, ub, ua);
        });
        const unsigned vst = rok ? voff : OOB;
        so = cbase;
        int ncs = nc;  // (a second opaque copy: the column predicates are formed again here instead of living in SGPRs across the strip)
        asm volatile("" : "+s"(so), "+s"(ncs));
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            buf_store_f64(rsrc, j < ncs ? vst : OOB, so, c[j]);
            so += n8;
            asm volatile("" : "+s"(so));
        }
    }
}

}  // namespace idahip
