// Device-resident LOCK-STEP stepper for larger systems (8 < n <= 1024 linear dense, n <= 4096 heat; device residual): a round = one step attempt of every
// system that is stepping, as in host/ensemble_ida.cpp -- but the scalar controller of every system lives on the device, the
// index lists are built there, and the host only enqueues the round's fixed sequence of launches. No host round trip inside a
// round; in throughput mode (idaens_stream, a fixed number of rounds) none at all between rounds.
//
// One round:
//   round_begin_kernel   workgroup / system: loop-top checks, begin_attempt (set_coeffs, lsetup decision), prediction
//   round_lists_kernel   one workgroup: the list of systems whose Newton solve starts with a linear setup (+ its length)
//   sys / sys+jac        the residual kernels of problem_kernels.hpp over the whole batch, each system skipped by one of them
//   lu_factor_batched    on the device-built list (launches sized for the worst case; surplus workgroups leave at once)
//   4 x { newton_iter_kernel ; round_newton_ctl_kernel (idaNlsConvTest and Newton's bookkeeping, thread / system) ; sys }
//   round_end_kernel     workgroup / system: final yy/yp + error-test norms, test_error / handle_n_flag / complete_step,
//                        stop tests, interpolation to tout, the next tout of the schedule, Ida::new again when streaming
// The control flow is ida_flow.hpp's (shared with the one-thread-per-system stepper), the decisions ida_controller.hpp's
// (shared with the host stepper). A Newton solve that ends in ConvergenceRecover with a stale Jacobian (newton.rs:146-152:
// set up again, iterate again) continues in the NEXT round's setup and iteration passes instead of extending this round: a
// system's own sequence of operations -- hence its results -- is unchanged, only the round in which they happen is.
#pragma once
#include "ida_flow.hpp"
#include "solve_kernels.hpp"
#include "vector_kernels.hpp"

namespace idahip {

struct RoundArgs {
    FlowArgs f;
    idactl::SysCore* sys;   // [batch]
    VecState v;
    const double *ic_y, *ic_yp;
    double *yout, *ypout;   // [ntout][batch][n] or null
    long long round_base;   // rounds completed before this launch sequence started
    long long round;        // index of this round within the call
    int first_round;        // systems enter the call in this round
    int fused_jac;          // the problem has a fused residual + Jacobian kernel (linear dense); otherwise two launches serve a setup
    int lu_period;          // idahip_set_lu_period: > 1 = a round may postpone its linear setups (round_lists_kernel), at most lu_period - 1 rounds in a row
    // per-system round state (device arrays of length batch)
    int* stepping;          // the system takes step attempts
    int* in_newton;         // the system's Newton solve is under way in this round
    int* skipP;             // residual kernel without setup: nonzero = not this system
    int* skipL;             // residual + Jacobian kernel and LU list
    int* skipI;             // newton_iter_kernel
    int* skipS;             // residual kernel inside the iteration passes
    int* ident;             // 0, 1, 2, ...: the kernels' index list
    int* lu_list;           // systems to factor
    int* lu_cnt;            // [1]
    int* lu_wait;           // [1] rounds in a row that have postponed their setups
    const int* lu_info;     // [batch] zero-pivot flags of the last factorisation
    double* tn;             // [batch] arguments of the residual kernels
    double* cj;
    double* scale;          // newton_iter_kernel's 2 / (1 + cjratio)
    double* nrm_out;        // newton_iter_kernel's sum of squares
    long long* rounds_done; // [batch]
    unsigned long long* stats;  // [IDAHIP_K_COUNT] systems served per kernel class in this call (the event timers' bookkeeping)
    int* summary;           // [2]: systems stepping after this round (zeroed before every round), systems that failed (this call)
    idahip_root_state* roots;  // [batch] or null (f.nrt == 0)
};

// vector backend of IdaFlow: a workgroup of WG_NT threads (one wavefront: every lane runs the scalar logic, in lock-step, on the
// workgroup's one copy of the controller record in LDS) owns system b; sums are accumulated left to right by one lane
// (seq_sum_lds) and broadcast through LDS, exactly as the batched kernels of vector_kernels.hpp do
constexpr int WG_NT = 64;

struct WgVec {
    const RoundArgs& a;
    const int b, n;
    const long vb;
    double* sm;  // LDS: 4 * n doubles of summands + 8 doubles of results

    __device__ double& phi(int j, int i) const { return a.v.phi[j * a.v.phistride + vb + i]; }
    __device__ double* res() const { return sm + 4 * n; }

    __device__ void init_first(double* ypnorm, double* p0nrm) const {
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += WG_NT) {
            const double y = phi(0, i);
            const double e = ewt_of(a.v, y, i);
            a.v.ewt[vb + i] = e;
            const double p = phi(1, i) * e;
            sm[i] = p * p;
            const double q = y * e;
            sm[n + i] = q * q;
        }
        __syncthreads();
        if (threadIdx.x < 2) res()[threadIdx.x] = seq_sum_lds(sm + threadIdx.x * n, n);
        __syncthreads();
        *ypnorm = sqrt(res()[0] / (double)n);
        *p0nrm = sqrt(res()[1] / (double)n);
    }
    __device__ void scale_phi1(double f) const {
        for (int i = threadIdx.x; i < n; i += WG_NT) phi(1, i) *= f;
    }
    __device__ void predict(const idactl::SysCore& s) const {
        for (int i = threadIdx.x; i < n; i += WG_NT) {
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
    __device__ void post_newton(const idactl::SysCore& s, double* norms) const {
        const int kk = s.kk;
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += WG_NT) {
            const double e = a.v.ee[vb + i];
            const double w = a.v.ewt[vb + i];
            a.v.yy[vb + i] = a.v.yypredict[vb + i] + e;
            a.v.yp[vb + i] = a.v.yppredict[vb + i] + s.cj * e;
            double p = e * w;
            sm[i] = p * p;
            double d = 0.0;
            if (kk > 1) {
                d = phi(kk, i) + e;
                p = d * w;
                sm[n + i] = p * p;
            } else {
                sm[n + i] = 0.0;
            }
            if (kk > 2) {
                d = d + phi(kk - 1, i);
                p = d * w;
                sm[2 * n + i] = p * p;
            } else {
                sm[2 * n + i] = 0.0;
            }
            if (kk + 1 < MXORDP1) {
                const double tmp = e - phi(kk + 1, i);
                p = tmp * w;
                sm[3 * n + i] = p * p;
            } else {
                sm[3 * n + i] = 0.0;
            }
        }
        __syncthreads();
        if (threadIdx.x < 4) res()[threadIdx.x] = seq_sum_lds(sm + threadIdx.x * n, n);
        __syncthreads();
        for (int k = 0; k < 4; ++k) norms[k] = sqrt(res()[k] / (double)n);
    }
    __device__ void restore_vec(const idactl::SysCore& s, int kk_att, int ns_att) const {
        if (ns_att > kk_att) return;
        for (int i = threadIdx.x; i < n; i += WG_NT)
            for (int j = ns_att; j <= kk_att; ++j) phi(j, i) *= s.cvals[j - ns_att];
    }
    __device__ void complete_step_vec(idactl::SysCore& s, int kused, double ck, int maxord) const {
        __syncthreads();
        if (threadIdx.x == 0) res()[1] = 0.0;
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += WG_NT) {
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
            if (!(w > 0.0)) res()[1] = 1.0;
            const double p = tmp * w;
            sm[i] = p * p;
        }
        __syncthreads();
        if (threadIdx.x == 0) res()[0] = seq_sum_lds(sm, n);
        __syncthreads();
        s.phi0nrm = sqrt(res()[0] / (double)n);
        s.ewt_bad = res()[1] != 0.0;
    }
    __device__ void get_solution_vec(const idactl::SysCore& s, int kord) const {
        for (int i = threadIdx.x; i < n; i += WG_NT) {
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
    // root functions (ida_flow.hpp): an element of yy may have been written by another thread of the workgroup
    __device__ void sync() const { __syncthreads(); }
    __device__ double yy_at(int i) const { return a.v.yy[vb + i]; }
    __device__ double phi_at(int j, int i) const { return a.v.phi[j * a.v.phistride + vb + i]; }
    __device__ void yy_from_phi01(double f) const {
        for (int i = threadIdx.x; i < n; i += WG_NT) a.v.yy[vb + i] = phi(0, i) + f * phi(1, i);
    }
    __device__ void yy_add_phi1(double f) const {
        for (int i = threadIdx.x; i < n; i += WG_NT) a.v.yy[vb + i] = a.v.yy[vb + i] + f * phi(1, i);
    }
    __device__ void emit_output(int slot) const {
        // (yy / yp were written by this thread's own get_solution_vec just before, same index mapping: no barrier needed)
        if (a.yout)
            for (int i = threadIdx.x; i < n; i += WG_NT) a.yout[((long)slot * a.f.batch + b) * n + i] = a.v.yy[vb + i];
        if (a.ypout)
            for (int i = threadIdx.x; i < n; i += WG_NT) a.ypout[((long)slot * a.f.batch + b) * n + i] = a.v.yp[vb + i];
    }
    __device__ void restore_initial() const {
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += WG_NT) {
            const double y = a.ic_y[vb + i], yp = a.ic_yp[vb + i];
            phi(0, i) = y;
            phi(1, i) = yp;
            a.v.yy[vb + i] = y;
            a.v.yp[vb + i] = yp;
        }
        __syncthreads();
    }
};

// ---- begin of a round: who steps, begin_attempt + prediction, which residual kernel serves the system
// The workgroup (one wavefront) keeps ONE copy of its system's controller record, in LDS: every lane runs the scalar logic
// on it in lock-step (same loads, same stores of the same values). A private copy per lane meant 64 reads and a spilled
// 736-byte struct per system and kernel.
static_assert(sizeof(idactl::SysCore) % 8 == 0 && WG_NT == 64, "word copies of the record; one wave per workgroup");
__device__ __forceinline__ void wg_copy_words(void* dst, const void* src) {
    for (int i = threadIdx.x; i < (int)(sizeof(idactl::SysCore) / 8); i += WG_NT)
        static_cast<unsigned long long*>(dst)[i] = static_cast<const unsigned long long*>(src)[i];
}

static_assert(sizeof(idahip_root_state) % 8 == 0 && sizeof(idahip_root_state) / 8 <= WG_NT, "word copies of the root state by one wave");
__device__ __forceinline__ void wg_copy_root(void* dst, const void* src) {
    if (threadIdx.x < (int)(sizeof(idahip_root_state) / 8)) static_cast<unsigned long long*>(dst)[threadIdx.x] = static_cast<const unsigned long long*>(src)[threadIdx.x];
}

template <bool ROOTS>
__global__ __launch_bounds__(WG_NT) void round_begin_kernel(RoundArgs a) {
    extern __shared__ __align__(16) double sm[];
    const int b = blockIdx.x;
    __shared__ __align__(16) unsigned char s_raw[sizeof(idactl::SysCore)];
    __shared__ __align__(16) idahip_root_state s_rt;
    wg_copy_words(s_raw, a.sys + b);
    if (ROOTS) wg_copy_root(&s_rt, a.roots + b);
    __syncthreads();
    idactl::SysCore& s = *reinterpret_cast<idactl::SysCore*>(s_raw);
    WgVec v{a, b, a.v.n, (long)b * a.v.n, sm};
    const IdaFlow<WgVec, ROOTS> F{a.f, s, v, ROOTS ? &s_rt : nullptr};
    const long long ground = a.round_base + a.round;
    bool stepping = a.first_round ? F.enter(ground, b) : (a.stepping[b] != 0);
    int kind = 0;  // 1: residual only, 2: residual + Jacobian + LU
    if (stepping) {
        if (s.newton_retry) {
            kind = 2;  // the Newton solve of the running attempt starts over with a linear setup (call_lsetup is set)
            s.newton_retry = false;
        } else if (s.ph == idactl::PH_LOOP_TOP && !F.loop_top()) {
            stepping = false;
        } else {
            F.attempt_begin();
            kind = s.call_lsetup ? 2 : 1;
        }
    }
    __syncthreads();
    wg_copy_words(a.sys + b, s_raw);
    if (ROOTS) wg_copy_root(a.roots + b, &s_rt);
    if (threadIdx.x == 0) {
        a.stepping[b] = stepping ? 1 : 0;
        a.in_newton[b] = kind != 0;
        a.skipP[b] = kind != 1;
        a.skipL[b] = kind != 2;
        a.skipI[b] = 1;
        a.skipS[b] = 1;
        a.tn[b] = s.tn;
        a.cj[b] = s.cj;
        if (kind != 0) a.rounds_done[b] += 1;
        if (kind == 1) atomicAdd(&a.stats[IDAHIP_K_SYS], 1ull);  // (kind 2: counted by round_lists_kernel, once the round's setups are decided)
    }
}

// ---- the list of systems to factor (ascending system id) and its length; one workgroup of 1024 threads
// Setups batched over rounds (idahip_set_lu_period(k), k > 1): a batched factorisation of matrices with thousands of rows costs about
// the same for 50 matrices as for 250 (a workgroup per matrix, a long chain of launches), and in a stream of integrations a third
// of the systems ask for one in any round. This kernel counts the systems whose attempt calls for a setup and the systems that
// step at all; unless the former are at least (k - 1) / k of the latter, or k - 1 rounds in a row have waited already, the round
// sets nothing up: those systems take no part in it -- their attempts begun, their predictions made, in the state of a system
// whose Newton solve starts over with a setup (newton_retry; call_lsetup is set either way: round_begin_kernel) -- and ask again
// next round, while the others go on stepping. Which round a system's attempt runs in is the scheduler's business: its
// arithmetic, its counters and its results do not change (tests/test_gpu_device_controller.py).
__global__ __launch_bounds__(1024) void round_lists_kernel(RoundArgs a) {
    __shared__ int s_wave[16], s_wave2[16];
    __shared__ int s_base, s_hold;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) { s_base = 0; s_hold = 0; }
    __syncthreads();
    if (a.lu_period > 1) {
        int want = 0, step = 0;
        for (int b = t; b < a.f.batch; b += 1024) {
            want += a.skipL[b] == 0 ? 1 : 0;
            step += a.stepping[b] != 0 ? 1 : 0;
        }
        for (int o = 32; o > 0; o >>= 1) {
            want += __shfl_xor(want, o);
            step += __shfl_xor(step, o);
        }
        if (lane == 0) { s_wave[wave] = want; s_wave2[wave] = step; }
        __syncthreads();
        if (t == 0) {
            int w = 0, st = 0;
            for (int q = 0; q < 16; ++q) { w += s_wave[q]; st += s_wave2[q]; }
            const int waited = a.lu_wait[0];
            const bool hold = w > 0 && (long)w * a.lu_period < (long)st * (a.lu_period - 1) && waited < a.lu_period - 1;
            s_hold = hold ? 1 : 0;
            a.lu_wait[0] = hold ? waited + 1 : 0;
        }
        __syncthreads();
        if (s_hold) {
            for (int b = t; b < a.f.batch; b += 1024)
                if (a.skipL[b] == 0) {
                    a.sys[b].newton_retry = true;
                    a.in_newton[b] = 0;
                    a.skipL[b] = 1;
                    a.rounds_done[b] -= 1;
                }
            if (t == 0) a.lu_cnt[0] = 0;
            return;
        }
        __syncthreads();
    }
    for (int b0 = 0; b0 < a.f.batch; b0 += 1024) {
        const int b = b0 + t;
        const bool in = b < a.f.batch && a.skipL[b] == 0;
        const unsigned long long bal = __ballot(in);
        if (lane == 0) s_wave[wave] = __popcll(bal);
        __syncthreads();
        int off = s_base;
        for (int q = 0; q < wave; ++q) off += s_wave[q];
        if (in) a.lu_list[off + __popcll(bal & ((1ull << lane) - 1ull))] = b;
        __syncthreads();
        if (t == 0) {
            int tot = 0;
            for (int q = 0; q < 16; ++q) tot += s_wave[q];
            s_base += tot;
        }
        __syncthreads();
    }
    if (t == 0) {
        a.lu_cnt[0] = s_base;
        const unsigned long long cnt = (unsigned long long)s_base;  // the per-class statistics of the systems that set up in this round
        if (cnt) {
            if (a.fused_jac) {
                atomicAdd(&a.stats[IDAHIP_K_SYS_JAC], cnt);
            } else {
                atomicAdd(&a.stats[IDAHIP_K_SYS], cnt);
                atomicAdd(&a.stats[IDAHIP_K_JAC], cnt);
            }
            atomicAdd(&a.stats[IDAHIP_K_LU], cnt);
        }
    }
}

// ---- Newton's bookkeeping between the batched kernels (newton.rs:73-153, ida_nls.rs:168-179, 218-266); one thread per system.
// phase 0: after the residual (and setup) kernels; phase 1..4: after the m-th newton_iter_kernel of the round
__global__ void round_newton_ctl_kernel(RoundArgs a, int phase) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    const bool mine = b < a.f.batch && a.in_newton[b] && (phase == 0 || a.skipI[b] == 0);
    // counters of the per-class statistics: one atomic per wavefront, not one per system
    const unsigned long long m_iter = __ballot(mine && phase != 0);
    if (m_iter != 0ull && (threadIdx.x & 63) == 0) atomicAdd(&a.stats[IDAHIP_K_NEWTON_ITER], (unsigned long long)__popcll(m_iter));
    bool again = false;
    if (mine) {
        idactl::SysCore& s = a.sys[b];  // in place: the few fields this step touches, not the 736-byte record both ways
        const int n = a.v.n;
        if (phase == 0) {
            s.nre += 1;  // sys(y0)
            bool go = true;
            if (a.skipL[b] == 0) {
                idactl::after_lsetup(s, a.lu_info[b]);
                if (s.nls_ret == idactl::NLS_LSETUP_RECVR) {
                    s.nconvfails += 1;  // jcur is true: no retry (newton.rs:146-153 with Q3)
                    go = false;
                }
            }
            if (go) {
                s.curiter = 0;
                a.skipI[b] = 0;
                a.scale[b] = idactl::after_lsolve(s, IDAHIP_LS_DIRECT /* = idahip_ls_type(): the only LSolver of this library */, 0, false) ? 2.0 / (1.0 + s.cjratio) : 1.0;  // ida_ls.rs:387-418
            }
        } else {
            a.skipS[b] = 1;
            const double delnrm = sqrt(a.nrm_out[b] / (double)n);
            s.niters += 1;
            bool converged = false;
            int ret = idactl::conv_test(s, delnrm, &converged);
            if (ret == idactl::NLS_SUCCESS && converged) {
                s.jcur = false;
                s.nls_ret = idactl::NLS_SUCCESS;
                a.skipI[b] = 1;
            } else {
                if (ret == idactl::NLS_SUCCESS) {
                    s.curiter += 1;
                    if (s.curiter >= idactl::MAXNLSIT) ret = idactl::NLS_CONV_RECVR;
                }
                if (ret == idactl::NLS_SUCCESS) {
                    a.skipS[b] = 0;  // sys(y), then iterate again
                    s.nre += 1;
                    again = true;
                } else {
                    s.nconvfails += 1;  // ConvergenceRecover
                    a.skipI[b] = 1;
                    if (!s.jcur) {
                        s.call_lsetup = true;
                        s.newton_retry = true;  // sys(y0) + setup + iterations again: in the next round
                    } else {
                        s.nls_ret = idactl::NLS_CONV_RECVR;
                    }
                }
            }
        }
    }
    const unsigned long long m_sys = __ballot(again);
    if (m_sys != 0ull && (threadIdx.x & 63) == 0) atomicAdd(&a.stats[IDAHIP_K_SYS], (unsigned long long)__popcll(m_sys));
}

// ---- end of a round: the rest of the attempt, the schedule, Ida::new again when streaming, the round's summary
template <bool ROOTS>
__global__ __launch_bounds__(WG_NT) void round_end_kernel(RoundArgs a) {
    extern __shared__ __align__(16) double sm[];
    const int b = blockIdx.x;
    __shared__ __align__(16) unsigned char s_raw[sizeof(idactl::SysCore)];
    __shared__ __align__(16) idahip_root_state s_rt;
    wg_copy_words(s_raw, a.sys + b);
    if (ROOTS) wg_copy_root(&s_rt, a.roots + b);
    __syncthreads();
    idactl::SysCore& s = *reinterpret_cast<idactl::SysCore*>(s_raw);
    WgVec v{a, b, a.v.n, (long)b * a.v.n, sm};
    const IdaFlow<WgVec, ROOTS> F{a.f, s, v, ROOTS ? &s_rt : nullptr};
    bool stepping = a.stepping[b] != 0;
    if (stepping && a.in_newton[b] && !s.newton_retry) stepping = F.attempt_end();
    const long long ground = a.round_base + a.round + 1;
    if (a.f.recycle) stepping = F.after_round_stream(stepping, ground, b, threadIdx.x == 0);
    __syncthreads();
    wg_copy_words(a.sys + b, s_raw);
    if (ROOTS) wg_copy_root(a.roots + b, &s_rt);
    if (threadIdx.x == 0) {
        a.stepping[b] = stepping ? 1 : 0;
        if (stepping) atomicAdd(&a.summary[0], 1);
        else if (s.status < 0 && a.in_newton[b]) atomicAdd(&a.summary[1], 1);
    }
}

__global__ void round_init_kernel(RoundArgs a) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b == 0) a.lu_wait[0] = 0;
    if (b >= a.f.batch) return;
    a.ident[b] = b;
    a.stepping[b] = 0;
    a.rounds_done[b] = 0;
}

}  // namespace idahip
