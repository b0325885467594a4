// Device-resident BDF stepper for small systems (n <= TINY_N): the WHOLE of Ida::solve -- step-size and order controller
// included -- runs on the device, one thread per IVP, for as many step attempts as the caller allows. No host round trip per
// Newton iteration, per step attempt or per output time: one launch per idaens_solve / _schedule / _stream call.
// (before this, a lock-step round of config 2 was ~12 launches and ~5 host synchronisations of pure latency.)
//
// The scalar decisions are the SAME source as the host stepper's (host/ida_controller.hpp: set_coeffs, begin_attempt,
// conv_test, test_error, handle_n_flag, complete_step_scalars, get_solution_coeffs), compiled for the device with
// pow = glibc_pow::pow (glibc_pow.hpp: glibc's __pow_fma restated instruction for instruction), so h, order and every counter
// carry the reference's bits. The vector parts repeat the arithmetic of the batched kernels (vector_kernels.hpp,
// problem_kernels.hpp, solve_kernels.hpp, lu_kernels.hpp's tiny_getrf) element for element; with one thread per system every
// sum is sequential by construction. The control flow of Ida::solve is ida_flow.hpp's (shared with the workgroup-per-system
// stepper of round_ida.hpp); this file supplies the one-thread vector backend and the in-thread Newton solve:
//   Ida::solve            /root/reference/src/impl_solve.rs:69-376     (first-call block, loop-top checks, stop tests)
//   Ida::step             /root/reference/src/lib.rs:613-711
//   nonlinear_solve       /root/reference/src/lib.rs:787-890
//   Newton::solve         /root/reference/crates/nonlinear/src/newton.rs:51-167 (Q3: break on ConvergenceRecover with a current J)
//   IdaNLProblem          /root/reference/src/ida_nls.rs:118-266
//   complete_step         /root/reference/src/impl_complete_step.rs:22-177
// Root finding for the family g_i = y[c_i] - thr_i runs here too (ida_flow.hpp). Not handled here (the host stepper keeps those
// cases): user root functions, IDA_ONE_STEP, host-callback problems, per-step traces.
#pragma once
#include "ida_flow.hpp"
#include "lu_kernels.hpp"
#include "problem_kernels.hpp"
#include "solve_kernels.hpp"
#include "vector_kernels.hpp"

namespace idahip {

struct TinyIdaArgs {
    FlowArgs f;
    idactl::SysCore* sys;  // [batch] the controller state, uploaded before and downloaded after the launch
    VecState v;            // phi, yy, yp, yypredict, yppredict, ewt, ee, delta
    double* savres;
    double* lu;            // [batch][n*n] Jacobian, factored in place
    long long* piv;        // [batch][n]
    const double* params;
    int nparam;
    const double *ic_y, *ic_yp;  // initial conditions (idaens_stream's restarts)
    long max_rounds;             // step attempts per system in this launch (0: until done)
    long long round_base;        // rounds executed before this launch
    double *yout, *ypout;        // [ntout][batch][n] or null: y, y' at every tout reached
    long long* rounds_done;      // [batch] rounds this system took part in during this launch
    idahip_root_state* roots;    // [batch] or null (f.nrt == 0)
};

// vector backend of IdaFlow: one thread owns system b (the arithmetic of vector_kernels.hpp, element for element)
struct TinyVec {
    const TinyIdaArgs& a;
    const int b, n;
    const long vb;   // offset of the system's vectors in a.v's arrays (0 when they are this thread's block of LDS)
    const long gvb;  // offset of the system in the global arrays (initial conditions)

    __device__ double& phi(int j, int i) const { return a.v.phi[j * a.v.phistride + vb + i]; }

    // initial_setup's ewt_set(phi[0]) and the two norms of the first call (init_first_kernel)
    __device__ void init_first(double* ypnorm, double* p0nrm) const {
        double s1 = 0.0, s0 = 0.0;
        for (int i = 0; i < n; ++i) {
            const double y = phi(0, i);
            const double e = ewt_of(a.v, y, i);
            a.v.ewt[vb + i] = e;
            const double p = phi(1, i) * e;
            s1 = s1 + p * p;
            const double q = y * e;
            s0 = s0 + q * q;
        }
        *ypnorm = sqrt(s1 / (double)n);
        *p0nrm = sqrt(s0 / (double)n);
    }
    __device__ void scale_phi1(double f) const {
        for (int i = 0; i < n; ++i) phi(1, i) *= f;
    }
    // predict_kernel
    __device__ void predict(const idactl::SysCore& s) const {
        for (int i = 0; i < n; ++i) {
            double yyp = 0.0, ypp = 0.0;
            for (int j = 0; j <= s.kk; ++j) {
                double p = phi(j, i);
                if (j >= s.ns) {
                    p *= s.beta[j];
                    phi(j, i) = p;
                }
                yyp = yyp + p;
                if (j >= 1) ypp = ypp + s.gamma[j] * p;
            }
            a.v.yypredict[vb + i] = yyp;
            a.v.yppredict[vb + i] = ypp;
        }
    }
    // final yy/yp and the four error-test norms (post_newton_kernel)
    __device__ void post_newton(const idactl::SysCore& s, double* norms) const {
        const int kk = s.kk;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        for (int i = 0; i < n; ++i) {
            const double e = a.v.ee[vb + i];
            const double w = a.v.ewt[vb + i];
            a.v.yy[vb + i] = a.v.yypredict[vb + i] + e;
            a.v.yp[vb + i] = a.v.yppredict[vb + i] + s.cj * e;
            double p = e * w;
            s0 = s0 + p * p;
            double d = 0.0;
            if (kk > 1) {
                d = phi(kk, i) + e;
                p = d * w;
                s1 = s1 + p * p;
            }
            if (kk > 2) {
                d = d + phi(kk - 1, i);
                p = d * w;
                s2 = s2 + p * p;
            }
            if (kk + 1 < MXORDP1) {
                const double tmp = e - phi(kk + 1, i);
                p = tmp * w;
                s3 = s3 + p * p;
            }
        }
        norms[0] = sqrt(s0 / (double)n);
        norms[1] = sqrt(s1 / (double)n);
        norms[2] = sqrt(s2 / (double)n);
        norms[3] = sqrt(s3 / (double)n);
    }
    // restore_kernel with the kk / ns of the failed attempt (cvals from restore_scalars)
    __device__ void restore_vec(const idactl::SysCore& s, int kk_att, int ns_att) const {
        if (ns_att > kk_att) return;
        for (int i = 0; i < n; ++i)
            for (int j = ns_att; j <= kk_att; ++j) phi(j, i) *= s.cvals[j - ns_att];
    }
    // complete_step_kernel: the phi recurrence, ee *= ck, the new ewt and ||phi[0]||
    __device__ void complete_step_vec(idactl::SysCore& s, int kused, double ck, int maxord) const {
        bool bad = false;
        double acc = 0.0;
        for (int i = 0; i < n; ++i) {
            const double e = a.v.ee[vb + i];
            if (kused < maxord) phi(kused + 1, i) = e;
            double tmp = e;
            for (int j = kused; j >= 0; --j) {
                tmp = tmp + phi(j, i);
                phi(j, i) = tmp;
            }
            a.v.ee[vb + i] = e * ck;
            const double w = ewt_of(a.v, tmp, i);
            a.v.ewt[vb + i] = w;
            if (!(w > 0.0)) bad = true;
            const double p = tmp * w;
            acc = acc + p * p;
        }
        s.phi0nrm = sqrt(acc / (double)n);
        s.ewt_bad = bad;
    }
    // get_solution_kernel
    __device__ void get_solution_vec(const idactl::SysCore& s, int kord) const {
        for (int i = 0; i < n; ++i) {
            double y = 0.0, yp = 0.0;
            for (int j = 0; j <= kord; ++j) {
                const double p = phi(j, i);
                y = y + s.cvals[j] * p;
                if (j >= 1) yp = yp + s.dvals[j - 1] * p;
            }
            a.v.yy[vb + i] = y;
            a.v.yp[vb + i] = yp;
        }
    }
    // root functions (ida_flow.hpp): one thread owns the system, nothing to synchronise
    __device__ void sync() const {}
    __device__ double yy_at(int i) const { return a.v.yy[vb + i]; }
    __device__ double phi_at(int j, int i) const { return phi(j, i); }
    __device__ void yy_from_phi01(double f) const {
        for (int i = 0; i < n; ++i) a.v.yy[vb + i] = phi(0, i) + f * phi(1, i);
    }
    __device__ void yy_add_phi1(double f) const {
        for (int i = 0; i < n; ++i) a.v.yy[vb + i] = a.v.yy[vb + i] + f * phi(1, i);
    }
    __device__ void emit_output(int slot) const {  // the output of the tout just reached (idaens_solve_schedule's hYout / hYPout)
        if (a.yout)
            for (int i = 0; i < n; ++i) a.yout[((long)slot * a.f.batch + b) * n + i] = a.v.yy[vb + i];
        if (a.ypout)
            for (int i = 0; i < n; ++i) a.ypout[((long)slot * a.f.batch + b) * n + i] = a.v.yp[vb + i];
    }
    __device__ void restore_initial() const {  // Ida::new again: restore_initial_kernel
        for (int i = 0; i < n; ++i) {
            const double y = a.ic_y[gvb + i], yp = a.ic_yp[gvb + i];
            phi(0, i) = y;
            phi(1, i) = yp;
            a.v.yy[vb + i] = y;
            a.v.yp[vb + i] = yp;
        }
    }
};

// Newton::solve for one attempt of one small system, in the owning thread (newton.rs:51-167 as ensemble_ida.cpp's
// newton_solve_batched runs it for one system; tiny_sys_kernel / tiny_jac_kernel / tiny_getrf / tiny_newton_iter_kernel)
template <int KIND>
struct TinyNewton {
    const TinyIdaArgs& a;
    idactl::SysCore& s;
    const int b, n;
    const long vb;   // as in TinyVec
    const long lub;  // offset of the system's matrix in a.lu

    // idaNlsResidual
    __device__ void nls_sys(bool reset_ee) const {
        double yy[TINY_N], yp[TINY_N], r[TINY_N];
        for (int i = 0; i < n; ++i) {
            double yc = a.v.ee[vb + i];
            if (reset_ee) {
                yc = 0.0;
                a.v.ee[vb + i] = 0.0;
            }
            yy[i] = a.v.yypredict[vb + i] + yc;
            yp[i] = a.v.yppredict[vb + i] + s.cj * yc;
            a.v.yy[vb + i] = yy[i];
            a.v.yp[vb + i] = yp[i];
        }
        if (KIND == IDAHIP_ROBERTS) roberts_res(yy, yp, r);
        else lorenz_res(a.params + (long)b * a.nparam, yy, yp, r);
        for (int i = 0; i < n; ++i) {
            a.v.delta[vb + i] = r[i];
            a.savres[vb + i] = r[i];
        }
        s.nre += 1;
    }
    // idaNlsLSetup: jac at the current yy + dense_get_rf in place; returns info
    __device__ int lsetup() const {
        double y[TINY_N], J[TINY_N * TINY_N];
        for (int i = 0; i < n; ++i) y[i] = a.v.yy[vb + i];
        if (KIND == IDAHIP_ROBERTS) roberts_jac(s.cj, y, J);
        else lorenz_jac(a.params + (long)b * a.nparam, s.cj, y, J);
        double* M = a.lu + lub;
        for (int e = 0; e < n * n; ++e) M[e] = J[e];
        int lperm[TINY_N];
        return tiny_getrf(M, n, a.piv + vb, lperm);
    }
    // one Newton iteration body; returns delnrm
    __device__ double newton_iter() const {
        double vv[TINY_N];
        for (int i = 0; i < n; ++i) vv[i] = -a.v.delta[vb + i];
        tiny_getrs(a.lu + lub, n, a.piv + vb, vv);
        const double sc = idactl::after_lsolve(s, IDAHIP_LS_DIRECT /* = idahip_ls_type(): the only LSolver of this library */, 0, false) ? 2.0 / (1.0 + s.cjratio) : 1.0;  // ida_ls.rs:387-418
        double acc = 0.0;
        for (int i = 0; i < n; ++i) {
            const double d = vv[i] * sc;
            a.v.delta[vb + i] = d;
            a.v.ee[vb + i] = a.v.ee[vb + i] + d;
            const double p = d * a.v.ewt[vb + i];
            acc = acc + p * p;
        }
        return sqrt(acc / (double)n);
    }
    __device__ void solve() const {
        bool restart = true;
        for (;;) {
            if (restart) {
                nls_sys(true);  // sys(y0), y <- y0 = 0 (newton.rs:73)
                if (s.call_lsetup) {
                    idactl::after_lsetup(s, lsetup());
                    if (s.nls_ret == idactl::NLS_LSETUP_RECVR) {
                        s.nconvfails += 1;  // jcur is true: no retry (newton.rs:146-153 with Q3)
                        return;
                    }
                }
                s.curiter = 0;
                restart = false;
            }
            const double delnrm = newton_iter();
            s.niters += 1;
            bool converged = false;
            int ret = idactl::conv_test(s, delnrm, &converged);
            if (ret == idactl::NLS_SUCCESS && converged) {
                s.jcur = false;
                s.nls_ret = idactl::NLS_SUCCESS;
                return;
            }
            if (ret == idactl::NLS_SUCCESS) {
                s.curiter += 1;
                if (s.curiter >= idactl::MAXNLSIT) ret = idactl::NLS_CONV_RECVR;
            }
            if (ret == idactl::NLS_SUCCESS) {
                nls_sys(false);  // sys(y), then iterate again
                continue;
            }
            s.nconvfails += 1;  // ConvergenceRecover
            if (!s.jcur) {
                s.call_lsetup = true;
                restart = true;
                continue;
            }
            s.nls_ret = idactl::NLS_CONV_RECVR;
            return;
        }
    }
};

// doubles of LDS one system's vectors take: phi[6], yy, yp, yypredict, yppredict, ewt, ee, delta, savres (14 n), the Jacobian /
// its factors (n^2) and the pivots (n)
__host__ __device__ constexpr int tiny_lds_doubles(int n) { return 14 * n + n * n + n; }

// Everything a system owns sits in LDS for the length of the launch -- the controller record (736 B) and, when the launch was
// given room for them (lds_vec), its vectors and its Jacobian: one lane walks dependent fp64 chains, and what it waits for is
// the latency of its own state (scratch and global memory: ~500 cycles a touch; LDS: ~60). The vector code is the same either way: the
// per-thread copy of the arguments points each field at this thread's block and the offsets vb / lub are zero. Config 2 (Lorenz63, 1024 systems): 26.0 -> 31.0 M iterations/s with the controller
// record alone.
template <int KIND, bool ROOTS>
__global__ __launch_bounds__(64) void tiny_ida_kernel(TinyIdaArgs ga, int lds_vec) {
    extern __shared__ __align__(16) unsigned char tiny_sm[];
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= ga.f.batch) return;
    // (blockDim.x = systems per wavefront, 64 or fewer: idahip_tiny_solve picks it by the batch size)
    idactl::SysCore& s = *reinterpret_cast<idactl::SysCore*>(tiny_sm + threadIdx.x * sizeof(idactl::SysCore));
    s = ga.sys[b];
    TinyIdaArgs a = ga;
    constexpr int n = 3;  // Roberts and Lorenz63 (idahip_create insists): a constant lets every vector loop unroll into registers
    const long gvb = (long)b * n;
    double* blk = reinterpret_cast<double*>(tiny_sm + blockDim.x * sizeof(idactl::SysCore)) + (long)threadIdx.x * tiny_lds_doubles(n);
    if (lds_vec) {
        for (int j = 0; j < MXORDP1; ++j)
            for (int i = 0; i < n; ++i) blk[j * n + i] = ga.v.phi[j * ga.v.phistride + gvb + i];
        double* const src[8] = {ga.v.yy, ga.v.yp, ga.v.yypredict, ga.v.yppredict, ga.v.ewt, ga.v.ee, ga.v.delta, ga.savres};
        for (int f = 0; f < 8; ++f)
            for (int i = 0; i < n; ++i) blk[(6 + f) * n + i] = src[f][gvb + i];
        for (int e = 0; e < n * n; ++e) blk[14 * n + e] = ga.lu[(long)b * n * n + e];
        long long* pv = reinterpret_cast<long long*>(blk + 14 * n + n * n);
        for (int i = 0; i < n; ++i) pv[i] = ga.piv[gvb + i];
        a.v.phi = blk;  // every field is indexed from 0 (vb = 0 below)
        a.v.phistride = n;
        a.v.yy = blk + 6 * n;
        a.v.yp = blk + 7 * n;
        a.v.yypredict = blk + 8 * n;
        a.v.yppredict = blk + 9 * n;
        a.v.ewt = blk + 10 * n;
        a.v.ee = blk + 11 * n;
        a.v.delta = blk + 12 * n;
        a.savres = blk + 13 * n;
        a.lu = blk + 14 * n;
        a.piv = pv;
    }
    const long vb = lds_vec ? 0 : gvb, lub = lds_vec ? 0 : (long)b * n * n;
    TinyVec v{a, b, n, vb, gvb};
    idahip_root_state rs;
    if (ROOTS) rs = ga.roots[b];
    const IdaFlow<TinyVec, ROOTS> F{a.f, s, v, ROOTS ? &rs : nullptr};
    const TinyNewton<KIND> N{a, s, b, n, vb, lub};
    long long ground = a.round_base;  // global round counter (idaens_stream: every system takes part in every round)
    long long done = 0;
    bool stepping = F.enter(ground, b);
    for (;;) {
        if (a.max_rounds > 0 && done >= a.max_rounds) break;
        if (!stepping && !a.f.recycle) break;
        if (stepping) {
            if (s.ph == idactl::PH_LOOP_TOP && !F.loop_top()) {
                stepping = false;
                if (!a.f.recycle) break;  // (the call returned at the top of a round the system takes no part in)
            } else {
                F.attempt_begin();
                N.solve();
                stepping = F.attempt_end();
            }
        }
        done += 1;
        ground += 1;
        if (a.f.recycle) {
            stepping = F.after_round_stream(stepping, ground, b, true);
            if (!stepping && s.status < 0) break;  // failed while streaming: the host reports it
        }
    }
    if (lds_vec) {
        for (int j = 0; j < MXORDP1; ++j)
            for (int i = 0; i < n; ++i) ga.v.phi[j * ga.v.phistride + gvb + i] = blk[j * n + i];
        double* const dst[8] = {ga.v.yy, ga.v.yp, ga.v.yypredict, ga.v.yppredict, ga.v.ewt, ga.v.ee, ga.v.delta, ga.savres};
        for (int f = 0; f < 8; ++f)
            for (int i = 0; i < n; ++i) dst[f][gvb + i] = blk[(6 + f) * n + i];
        for (int e = 0; e < n * n; ++e) ga.lu[(long)b * n * n + e] = blk[14 * n + e];
        const long long* pv = reinterpret_cast<const long long*>(blk + 14 * n + n * n);
        for (int i = 0; i < n; ++i) ga.piv[gvb + i] = pv[i];
    }
    ga.sys[b] = s;
    if (ROOTS) ga.roots[b] = rs;
    ga.rounds_done[b] = done;
}

}  // namespace idahip
