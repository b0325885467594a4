// Host driver of the batched LU: picks the kernel pipeline (ctx->lu_variant) and launches it for a list of systems.
#pragma once
#include "lu_kernels.hpp"
#include "lu_wavepanel.hpp"
#include "lu_superpanel.hpp"

namespace idahip {

// ------------------------------------------------------------------------------------------------ host driver
// Factor the matrices `work[b]` (column-major, physical row order, destroyed) of the listed systems into out[b]
// (reference layout: rows at their pivoted positions). 64-column super-panels; per super-panel the panel factorisation
// (by live rows: > 1024 lu_panelr, > 512 or variant 3 lu_panel2 + narrow update, else lu_wavepanel) and one launch of the
// rank-64 trailing kernel; a final row scatter.
// d_cnt (optional): the list's length lives on the device and nsys is only its upper bound (variant 4): every kernel's
// surplus workgroups leave on reading it.
inline int lu_factor_batched(idahip_ctx* c, double* work, long wstride, double* out, long ostride, long long* piv, long pstride,
                             int* perm, const int* d_idx, int nsys, const int* d_cnt = nullptr) {
    const int n = c->n;
    if (nsys == 0) return 0;
    if (n <= TINY_N) {
        // tiny path factors in place in `work` (one thread per system, reference loops verbatim)
        hipLaunchKernelGGL(tiny_getrf_kernel, dim3((nsys + 63) / 64), dim3(64), 0, c->stream, work, wstride, d_idx, nsys, n, piv,
                           pstride, perm, c->lu_info);
        return 0;
    }
    if (n > LU_BIG_MAX_N) return fail(c, -3, "blocked LU supports n <= %d in this build (n = %d)", LU_BIG_MAX_N, n);
    if (d_cnt && c->lu_variant < 4) return fail(c, -3, "a device-side list length needs LU variant 4");
    LuWs w;
    w.mats = work; w.mstride = wstride; w.idx = d_idx; w.cnt = d_cnt; w.n = n;
    w.pos = c->lu_pos; w.live = c->lu_live; w.prow = c->lu_prow; w.piv = piv; w.pstride = pstride; w.info = c->lu_info; w.redo = c->lu_redo; w.nzb = c->lu_nzb; w.bz = c->lu_bz;
    w.zmap = (out == c->lu) ? c->lu_zmap : nullptr;  // the map describes the ctx's own factors (the Newton iteration's solves)
    w.dirty = (out == c->lu) ? c->lu_dirty : nullptr;  // (nothing else ever writes the ctx's own factors for n >= 2048)
    // the work matrix left all +0.0 for the next Jacobian (heat_jac_kernel): only where every super-panel is 64 columns wide -- the
    // scatter's regions are then whole blocks of `dirty` -- and the matrices factored are the ctx's own work matrices
    w.jwzero = (w.dirty && work == c->jw && c->lu_superpanel && n > LU_MAX_N) ? c->lu_jwzero : nullptr;
    if (!w.jwzero && c->lu_jwzero && (work == c->jw || out == c->jw))  // the work matrix is about to hold something this flag does not describe
        (void)hipMemsetAsync(c->lu_jwzero, 0, (size_t)c->batch * sizeof(int), c->stream);
    w.l11 = c->lu_l11; w.stamps = c->dbg_stamps; w.out = out; w.ostride = ostride; w.l11ld = 64;
    hipLaunchKernelGGL(lu_init_kernel, dim3(nsys), dim3(256), 0, c->stream, w);
    const int nsys8 = ((nsys + 7) / 8) * 8;
    constexpr int NB = LU_NB;
    auto panel2 = [&](int k0, int lbase) {  // two live rows per lane
        const int threads = (((n - k0 + 1) / 2 + 63) / 64) * 64;
        hipLaunchKernelGGL((lu_panel2_kernel<NB, 512, 2>), dim3(nsys), dim3(threads), 0, c->stream, w, k0, lbase);
    };
    for (int k0 = 0; k0 < n; k0 += 64) {
        if (n - k0 > LU_MAX_N) {
            // more than 1024 live rows: 8-column panels with eight rows per lane (16-column panels with four rows per lane
            // once at most LU_WIDE_ROWS rows are live: half the launches), each followed by its narrow update of the rest of
            // the super-panel
            KTimer kt(c, IDAHIP_K_LU_PANEL, nsys);
            const int cend = (k0 + 64 < n) ? k0 + 64 : n;
            if (c->lu_superpanel) {
                // the whole super-panel in one launch (lu_superpanel.hpp), eight rows per lane above LU_WIDE_ROWS live rows, four below (up to 512 threads)
                if (n - k0 > LU_WIDE_ROWS) {
                    const int threads = (((n - k0 + 7) / 8 + 63) / 64) * 64;
                    hipLaunchKernelGGL((lu_superpanel_kernel<8, 512, 2>), dim3(nsys), dim3(threads), 0, c->stream, w, k0);
                } else {
                    const int threads = (((n - k0 + 3) / 4 + 63) / 64) * 64;
                    hipLaunchKernelGGL((lu_superpanel_kernel<4, 512, 2>), dim3(nsys), dim3(threads), 0, c->stream, w, k0);
                }
            } else {
            auto narrow_split = [&](int kk, int nbs) {
                // one column block per matrix: the row tiles are dealt to several workgroups (a launch of one workgroup per
                // matrix leaves most of the chip idle at the batch sizes of large n)
                const int ntiles = (n - kk - nbs + 63) / 64;
                return ntiles >= 32 ? 8 : ntiles >= 16 ? 4 : ntiles >= 8 ? 2 : 1;
            };
            if (n - k0 > LU_WIDE_ROWS) {
                constexpr int NBS = 8;
                const int threads = (((n - k0 + 7) / 8 + 63) / 64) * 64;
                for (int lb = 0; lb < 64 && k0 + lb < n; lb += NBS) {
                    hipLaunchKernelGGL((lu_panelr_kernel<NBS, 8, 512, 2>), dim3(nsys), dim3(threads), 0, c->stream, w, k0 + lb, lb);
                    if (k0 + lb + NBS < cend) {
                        const int nsplit = narrow_split(k0 + lb, NBS);
                        hipLaunchKernelGGL((lu_trail_kernel<NBS, LU_BIG_MAX_N, 4>), dim3(nsys8 * nsplit), dim3(256), 0, c->stream, w, k0 + lb, nsys, 1,
                                           cend, lb * 65, nsplit);
                    }
                }
            } else {
                constexpr int NBS = 16;
                const int threads = (((n - k0 + 3) / 4 + 63) / 64) * 64;
                for (int lb = 0; lb < 64 && k0 + lb < n; lb += NBS) {
                    hipLaunchKernelGGL((lu_panelr_kernel<NBS, 4, 512, 2>), dim3(nsys), dim3(threads), 0, c->stream, w, k0 + lb, lb);
                    if (k0 + lb + NBS < cend) {
                        const int nsplit = narrow_split(k0 + lb, NBS);
                        hipLaunchKernelGGL((lu_trail_kernel<NBS, LU_BIG_MAX_N, 4>), dim3(nsys8 * nsplit), dim3(256), 0, c->stream, w, k0 + lb, nsys, 1,
                                           cend, lb * 65, nsplit);
                    }
                }
            }
            }  // !lu_superpanel
        } else if (c->lu_superpanel && c->lu_variant >= 4 && n > LU_MAX_N && n - k0 >= 64) {
            // large n (small batches): the remaining super-panels by the workgroup-per-matrix kernel too, two rows per lane -- a wave
            // per matrix (lu_wavepanel) is made for thousands of matrices per call, and on banded matrices its FAST mode hands
            // every super-panel to the one-wave SLOW launch (280 us each at ~100 matrices)
            KTimer kt(c, IDAHIP_K_LU_PANEL, nsys);
            const int threads = (((n - k0 + 1) / 2 + 63) / 64) * 64;
            hipLaunchKernelGGL((lu_superpanel_kernel<2, 512, 2>), dim3(nsys), dim3(threads), 0, c->stream, w, k0);
        } else if (c->lu_variant >= 4 && n - k0 <= WP_MAX_ROWS) {
            KTimer kt(c, IDAHIP_K_LU_PANEL, nsys);
            // one wave per matrix factors the whole 64-column super-panel (lu_wavepanel.hpp); the second launch finishes
            // the few systems whose matrices have exact zeros or special values (it returns at once for the others)
            const int ns = (n - k0 + 63) / 64;
#define IDAHIP_WP_LAUNCH(NSV) \
    hipLaunchKernelGGL((lu_wavepanel_kernel<false, NSV>), dim3(nsys), dim3(64), 0, c->stream, w, k0)
#define IDAHIP_WP_SWITCH()                                                                                    \
    switch (ns) {                                                                                              \
        case 1: IDAHIP_WP_LAUNCH(1); break;                                                                 \
        case 2: IDAHIP_WP_LAUNCH(2); break;                                                                 \
        case 3: IDAHIP_WP_LAUNCH(3); break;                                                                 \
        case 4: IDAHIP_WP_LAUNCH(4); break;                                                                 \
        case 5: IDAHIP_WP_LAUNCH(5); break;                                                                 \
        case 6: IDAHIP_WP_LAUNCH(6); break;                                                                 \
        case 7: IDAHIP_WP_LAUNCH(7); break;                                                                 \
        default: IDAHIP_WP_LAUNCH(8); break;                                                                \
    }
            IDAHIP_WP_SWITCH()
            hipLaunchKernelGGL((lu_wavepanel_kernel<true, 0>), dim3(nsys), dim3(64), 0, c->stream, w, k0);
#undef IDAHIP_WP_SWITCH
#undef IDAHIP_WP_LAUNCH
        } else {
            // two 32-column panels; the first one's update reaches the second through a narrow (32-column) launch of the
            // trailing kernel
            KTimer kt(c, IDAHIP_K_LU_PANEL, nsys);
            panel2(k0, 0);
            if (n - k0 > NB) {
                const int cend = (k0 + 64 < n) ? k0 + 64 : n;
                hipLaunchKernelGGL((lu_trail_kernel<NB, LU_MAX_N, 2>), dim3(nsys8), dim3(256), 0, c->stream, w, k0, nsys, 1, cend, 0, 1);
                panel2(k0 + NB, NB);
            }
        }
        const int ntrail = n - k0 - 64;
        if (ntrail > 0) {
            KTimer kt(c, IDAHIP_K_LU_TRAIL, nsys);
            const int ncb = (ntrail + 63) / 64;
            if (n > LU_MAX_N) {
                // Large n, small batches: a banded matrix in dense storage (the heat equation's Jacobian) leaves one column block
                // per matrix with work and that workgroup then walks all the live rows alone. The launch holds nsplit workgroups
                // for each of the first column blocks; they share the rows once the first super-panel has shown (on the device, LuWs::nzb) that few
                // blocks have work, and leave at once otherwise.
                const int nstrips = (ntrail + 15) / 16;
                int nsplit = nstrips >= 64 ? 8 : nstrips >= 32 ? 4 : nstrips >= 16 ? 2 : 1;
                // two workgroups per CU: the helpers must not queue behind each other. (With the list's length on the device the
                // kernel makes the same reduction from the count it reads; the launch then carries the helpers of the largest split.)
                while (!d_cnt && nsplit > 1 && nsys * nsplit > 512) nsplit >>= 1;
                const int nbs = nsplit > 1 ? std::min(ncb, LU_SPLIT_BLOCKS) : 0;
                // first the column blocks whose 64 pivot rows are zero (off the band): found, and their U12 written, by a light
                // kernel at full occupancy; the update kernel's workgroups for them leave at once
                hipLaunchKernelGGL(lu_u12_zero_kernel, dim3(nsys8 * ncb), dim3(256), 0, c->stream, w, k0, nsys, ncb);
                hipLaunchKernelGGL(lu_trail64w_kernel<LU_BIG_MAX_N>, dim3(nsys8 * (ncb + (nsplit - 1) * nbs)), dim3(256), 0, c->stream, w, k0, nsys, ncb, nsplit);
            } else
                hipLaunchKernelGGL(lu_trail64w_kernel<LU_MAX_N>, dim3(nsys8 * ncb), dim3(256), 0, c->stream, w, k0, nsys, ncb, 1);
        }
    }
    {
        KTimer kt(c, IDAHIP_K_LU_FINALIZE, nsys);
        hipLaunchKernelGGL(lu_finalize_kernel, dim3(nsys, (n + 31) / 32), dim3(256), 0, c->stream, w, out, ostride, perm, 32, NB,
                           n > LU_MAX_N ? (c->lu_superpanel ? 64 : 8) : 0,  // (8: 16 where <= LU_WIDE_ROWS rows were live)
                           c->lu_variant >= 4 ? ((c->lu_superpanel && n > LU_MAX_N) ? LU_MAX_N : WP_MAX_ROWS) : 0);  // super-panels factored 64 columns at a time
    }
    return 0;
}

}  // namespace idahip
