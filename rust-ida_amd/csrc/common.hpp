// libidahip -- shared definitions: the ensemble context, per-call argument staging, error plumbing.
// gfx950 only; fp64; compiled with -ffp-contract=off so device arithmetic is the reference's (no FMA).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdlib>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/ida_hip.h"
#include "exp_switches.hpp"

namespace idahip {

constexpr int MXORDP1 = 6;      // src/constants.rs:6
constexpr int TINY_N = 8;       // n <= TINY_N: one thread per system (whole Newton body in registers/L1)
constexpr int LU_NB = 32;       // panel width of the blocked LU
constexpr int LU_MAX_N = 1024;  // blocked LU, fast pipelines: at most two panel rows per lane of a 512-thread workgroup
constexpr int LU_WIDE_ROWS = 2048;  // n > 1024: at most this many live rows -> 16-column panels with four rows per lane (above: 8 columns, eight rows)
constexpr int LU_ZMAP_MIN_N = 2048;  // from this size on the factorisation leaves a map of the factors' zero blocks for the triangular solves
constexpr int LU_BIG_MAX_N = 4096;  // blocked LU with eight panel rows per lane for the leading super-panels
constexpr int NSLOT = 8;

struct Slot {
    char* h = nullptr;  // pinned host
    char* d = nullptr;  // device mirror
    hipEvent_t done = nullptr;
    bool pending = false;
};

}  // namespace idahip

struct idahip_ctx {
    int device = 0;
    int n = 0;
    int batch = 0;
    int npad16 = 0;  // n rounded up to a multiple of 16 (Ubuf row stride)
    int simds = 0;   // SIMDs of the device (4 per CU): idahip_tiny_solve sizes its wavefronts by it
    idahip_problem kind = IDAHIP_ROBERTS;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;

    // tolerances
    double rtol = 0.0;
    double atol_s = 0.0;
    double* d_atol_v = nullptr;  // [n] or null

    // state vectors [batch][n]
    double *yy = nullptr, *yp = nullptr, *yypredict = nullptr, *yppredict = nullptr, *ewt = nullptr, *ee = nullptr,
           *delta = nullptr, *savres = nullptr;
    double* phi = nullptr;  // [6][batch][n]

    // linear solver state
    double* lu = nullptr;    // [batch][n*n] factored Jacobian, reference layout (rows in pivoted order)
    double* jw = nullptr;    // [batch][n*n] work matrix of the factorisation (physical row order)
    int64_t* piv = nullptr;  // [batch][n]  reference pivots (dense.rs:118)
    int32_t* perm = nullptr; // [batch][n]  composed row permutation: b_perm[i] = b[perm[i]]
    // blocked-LU workspace
    int32_t *lu_pos = nullptr, *lu_live = nullptr, *lu_prow = nullptr, *lu_info = nullptr, *lu_redo = nullptr, *lu_nzb = nullptr, *lu_bz = nullptr;
    int32_t* lu_jwzero = nullptr; // [batch], n >= 2048: 1 = the last factorisation has left the system's work matrix (jw) all +0.0 (LuWs::jwzero)
    uint8_t* lu_dirty = nullptr; // [batch][64][64], n >= 2048: [K][I] = 1 once the 64 x 64 block (rows I, columns K) of `lu` has received a value with non-zero bits (never reset: `lu` starts as zeros and lu_finalize_kernel does not rewrite zeros into blocks that have only ever held zeros)
    uint8_t* lu_zmap = nullptr;  // [batch][64][64], n >= 2048: [K][I] = 1 when the 64 x 64 block (rows I, columns K) of the factors in `lu` may hold a non-zero
    double* lu_l11 = nullptr;
    double *ic_y = nullptr, *ic_yp = nullptr;  // [batch][n] initial conditions kept for idahip_restore_initial (lazy)
    double* dky = nullptr;                     // [batch][n] result buffer of idahip_get_dky (lazy)
    int lu_variant = 4;  // 4: one wave per matrix factors each 64-column super-panel (lu_wavepanel.hpp, default)
    // n > 1024: 1 = a 64-column super-panel is ONE launch of lu_superpanel_kernel (left-looking; made for banded matrices in dense
    // storage, where it is 26 % faster: config 4), 0 = round 4's eight 8-column panel launches with a narrow update after each,
    // which spread a dense matrix's update over up to 8 workgroups and stay faster there (dense n = 1536 / 2048 / 4096: 6 / 11 / 48 %).
    // Default by problem: on for IDAHIP_HEAT1D (tridiagonal content by construction), off otherwise; idahip_set_lu_superpanel
    // or IDAHIP_LU_SUPERPANEL=0/1 (read at idahip_create) override. Bit-identical factors either way (tests run both).
    int lu_superpanel = 0;
    // device lock-step stepper: linear setups batched over rounds (round_lists_kernel): with k > 1 a round postpones its setups unless
    // (k - 1) / k of the stepping systems ask for one, at most k - 1 rounds in a row. Pays where a factorisation's cost hardly depends
    // on the number of matrices (n > 1024: one workgroup per matrix, a chain of launches): idahip_set_lu_period / IDAHIP_LU_PERIOD.
    int lu_period = 1;
                         // 3: panel kernels with two rows per lane + narrow update (lu_kernels.hpp): cross-check, and n > 512

    // device-resident stepper for small systems (tiny_ida.hpp): controller states and per-call buffers (lazy)
    void* tiny_sys = nullptr;
    double *tiny_touts = nullptr, *tiny_yout = nullptr, *tiny_ypout = nullptr;
    int64_t *tiny_start = nullptr, *tiny_rounds = nullptr;
    uint64_t* tiny_acc = nullptr;
    void* tiny_roots = nullptr;  // [batch] idahip_root_state (lazy: root finding on the device steppers)
    int tiny_ntout_cap = 0, tiny_yout_cap = 0;

    // device-resident lock-step stepper (round_ida.hpp): per-system round state, the LU list, the round summary (lazy)
    int32_t* rnd_i = nullptr;   // 8 int arrays of length batch + lu_cnt[1] + summary[2]
    double* rnd_d = nullptr;    // 4 double arrays of length batch
    int32_t* rnd_host = nullptr;  // pinned: the round summary

    unsigned long long* dbg_stamps = nullptr;  // timing builds only (idahip_debug_stamps)

    // problem data
    double* params = nullptr;  // [batch][nparam]
    int nparam = 0;
    double *A = nullptr, *B = nullptr, *C = nullptr;  // LINEAR_DENSE
    idahip_res_fn cb_res = nullptr;                   // HOST_CALLBACK
    idahip_jac_fn cb_jac = nullptr;
    void* cb_user = nullptr;
    double* cb_stage = nullptr;                       // [batch][3][n] device staging of yy, yp, res of the listed systems
    std::vector<double> cb_host;                      // host mirror of cb_stage
    double *cb_jpin = nullptr, *cb_jdev = nullptr;    // pinned / device staging of a chunk of user Jacobians (lazy)
    size_t cb_jcap = 0;                               // systems the two staging buffers hold

    // staging ring
    idahip::Slot slots[idahip::NSLOT];
    size_t slot_cap = 0;
    int next_slot = 0;

    // timing
    int timing = 0;  // 0 off, 1 per kernel class, 2 also per kernel of the LU
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
    hipEvent_t ev_cnt = nullptr;  // idahip_round_solve, n > 1024: the LU list's length has reached the host
    double k_ms[IDAHIP_K_COUNT] = {0};
    int64_t k_launches[IDAHIP_K_COUNT] = {0};
    int64_t k_systems[IDAHIP_K_COUNT] = {0};
};

namespace idahip {

inline int fail(idahip_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    return code;
}

#define IDAHIP_HIP(c, call)                                                                              \
    do {                                                                                                 \
        hipError_t e__ = (call);                                                                         \
        if (e__ != hipSuccess) return idahip::fail((c), -100, "%s failed: %s", #call, hipGetErrorString(e__)); \
    } while (0)

// Every entry point of the C ABI runs on the ctx's own device whatever device is current in the calling thread (one host
// thread may drive several GPUs); the caller's current device is put back on return.
struct DevGuard {
    int prev = -1;
    bool switched = false;
    explicit DevGuard(int dev) {
        if (dev >= 0 && hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
    }
    explicit DevGuard(const idahip_ctx* c) : DevGuard(c ? c->device : -1) {}
    ~DevGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
    DevGuard(const DevGuard&) = delete;
    DevGuard& operator=(const DevGuard&) = delete;
};

// One call's host->device arguments and device->host results, carved from a ring slot.
struct ArgPack {
    idahip_ctx* c;
    Slot* s;
    size_t in_bytes = 0;   // bytes to upload (prefix of the slot)
    size_t off = 0;
    size_t out_begin = 0, out_end = 0;
    bool overflow = false;

    int begin(idahip_ctx* ctx) {
        c = ctx;
        s = &ctx->slots[ctx->next_slot];
        ctx->next_slot = (ctx->next_slot + 1) % NSLOT;
        if (s->pending) {
            IDAHIP_HIP(c, hipEventSynchronize(s->done));
            s->pending = false;
        }
        off = 0;
        return 0;
    }
    static size_t align(size_t x) { return (x + 63) & ~(size_t)63; }
    // copy `bytes` from host into the slot; returns the device address
    // (a request that does not fit the slot is not staged: `overflow` makes upload() / fetch() fail before anything of this
    // call is launched or read back -- the largest call today, predict, stages about 110 bytes per system of slot_cap's 256)
    template <class T>
    const T* in(const T* src, size_t count) {
        off = align(off);
        if (off + count * sizeof(T) > c->slot_cap) {
            overflow = true;
            return (const T*)s->d;
        }
        memcpy(s->h + off, src, count * sizeof(T));
        const T* d = (const T*)(s->d + off);
        off += count * sizeof(T);
        in_bytes = off;
        return d;
    }
    int upload() {
        if (overflow) return fail(c, -5, "argument staging slot too small (%zu bytes)", c->slot_cap);
        if (in_bytes) IDAHIP_HIP(c, hipMemcpyAsync(s->d, s->h, in_bytes, hipMemcpyHostToDevice, c->stream));
        return 0;
    }
    // reserve an output region (call after all in()); returns the device address
    template <class T>
    T* out(size_t count) {
        off = align(off);
        if (off + count * sizeof(T) > c->slot_cap) {
            overflow = true;
            return nullptr;  // checked by the callers through ok() before the launch
        }
        if (out_begin == 0 && out_end == 0) out_begin = off;
        T* d = (T*)(s->d + off);
        off += count * sizeof(T);
        out_end = off;
        return d;
    }
    template <class T>
    const T* host_of(const T* dptr) const { return (const T*)(s->h + ((const char*)dptr - s->d)); }
    int ok() { return overflow ? fail(c, -5, "argument staging slot too small (%zu bytes)", c->slot_cap) : 0; }
    // download the output region and wait for it
    int fetch() {
        if (overflow) return ok();
        if (out_end > out_begin)
            IDAHIP_HIP(c, hipMemcpyAsync(s->h + out_begin, s->d + out_begin, out_end - out_begin, hipMemcpyDeviceToHost, c->stream));
        IDAHIP_HIP(c, hipStreamSynchronize(c->stream));
        return 0;
    }
    // mark the slot busy until the stream reaches this point (calls without results)
    int finish_async() {
        IDAHIP_HIP(c, hipEventRecord(s->done, c->stream));
        s->pending = true;
        return 0;
    }
};

struct KTimer {
    idahip_ctx* c;
    idahip_kclass k;
    bool on;
    hipEvent_t e0, e1;
    KTimer(idahip_ctx* ctx, idahip_kclass kc, int nsys) : c(ctx), k(kc) {
        const bool sub = kc >= IDAHIP_K_LU_PANEL;  // a single kernel inside a class: its own event pair, level 2 only
        on = sub ? c->timing >= 2 : c->timing >= 1;
        e0 = sub ? c->ev2 : c->ev0;
        e1 = sub ? c->ev3 : c->ev1;
        c->k_launches[k] += 1;
        c->k_systems[k] += nsys;
        if (on) (void)hipEventRecord(e0, c->stream);
    }
    ~KTimer() {
        if (on) {
            (void)hipEventRecord(e1, c->stream);
            (void)hipEventSynchronize(e1);
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            c->k_ms[k] += (double)ms;
        }
    }
};

// Load through the constant address space: for wave-uniform addresses of memory that no thread of the running kernel
// writes, hipcc then emits scalar loads (s_load_dword*) and the value lives in SGPRs -- usable directly as a VALU
// operand, with no LDS or vector-memory traffic. (A plain load from a struct-member pointer is a per-lane
// global_load even when the address is uniform.)
template <class T>
__device__ __forceinline__ T ldc(const T* p) {
    typedef const T __attribute__((address_space(4))) * cptr;
    return *reinterpret_cast<cptr>(reinterpret_cast<uintptr_t>(p));
}

__device__ __forceinline__ double readlane_f64(double v, int lane /* wave-uniform */) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// ---- wave64 reductions on the DPP crossbar (no LDS round trips): row_shr 1/2/4/8 fold each 16-lane row into its last
// lane, row_bcast:15 / row_bcast:31 fold the four rows into lane 63; the result is read back with v_readlane.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_mov_i32(int x) {
    return __builtin_amdgcn_update_dpp(x, x, CTRL, ROW_MASK, 0xf, false);  // lanes without a source keep their own value
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_mov_f64(double x) {
    const int lo = dpp_mov_i32<CTRL, ROW_MASK>(__double2loint(x));
    const int hi = dpp_mov_i32<CTRL, ROW_MASK>(__double2hiint(x));
    return __hiloint2double(hi, lo);
}
// max over the wave of a NaN-free double; every lane returns the maximum
__device__ __forceinline__ double wave_max_f64(double v) {
    double o;
    o = dpp_mov_f64<0x111, 0xf>(v); v = o > v ? o : v;  // row_shr:1
    o = dpp_mov_f64<0x112, 0xf>(v); v = o > v ? o : v;  // row_shr:2
    o = dpp_mov_f64<0x114, 0xf>(v); v = o > v ? o : v;  // row_shr:4
    o = dpp_mov_f64<0x118, 0xf>(v); v = o > v ? o : v;  // row_shr:8
    o = dpp_mov_f64<0x142, 0xa>(v); v = o > v ? o : v;  // row_bcast:15 -> rows 1, 3
    o = dpp_mov_f64<0x143, 0xc>(v); v = o > v ? o : v;  // row_bcast:31 -> rows 2, 3
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}
// min over the wave of an int; every lane returns the minimum
__device__ __forceinline__ int wave_min_i32(int v) {
    int o;
    o = dpp_mov_i32<0x111, 0xf>(v); v = o < v ? o : v;
    o = dpp_mov_i32<0x112, 0xf>(v); v = o < v ? o : v;
    o = dpp_mov_i32<0x114, 0xf>(v); v = o < v ? o : v;
    o = dpp_mov_i32<0x118, 0xf>(v); v = o < v ? o : v;
    o = dpp_mov_i32<0x142, 0xa>(v); v = o < v ? o : v;
    o = dpp_mov_i32<0x143, 0xc>(v); v = o < v ? o : v;
    return __builtin_amdgcn_readlane(v, 63);
}

// ---- one-instruction DPP folds: the neutral element is the `old` operand (and bound_ctrl for the unsigned max), so the
// compiler fuses the cross-lane move into v_max_u32_dpp / v_min_i32_dpp -- one VALU op per stage instead of six for a
// double compare-and-select. ROWS16 = true stops after the row folds: every 16-lane row's result sits in its lane 15.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_umax(unsigned x) {
    const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROW_MASK, 0xf, true);
    return o > x ? o : x;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_imin(int x) {
    const int o = __builtin_amdgcn_update_dpp(0x7fffffff, x, CTRL, ROW_MASK, 0xf, false);
    return o < x ? o : x;
}
template <bool ROWS16>
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    v = dpp_umax<0x111, 0xf>(v);
    v = dpp_umax<0x112, 0xf>(v);
    v = dpp_umax<0x114, 0xf>(v);
    v = dpp_umax<0x118, 0xf>(v);
    if (ROWS16) return (unsigned)__builtin_amdgcn_readlane((int)v, 15);
    v = dpp_umax<0x142, 0xa>(v);
    v = dpp_umax<0x143, 0xc>(v);
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
template <bool ROWS16>
__device__ __forceinline__ int wave_min_i32f(int v) {
    v = dpp_imin<0x111, 0xf>(v);
    v = dpp_imin<0x112, 0xf>(v);
    v = dpp_imin<0x114, 0xf>(v);
    v = dpp_imin<0x118, 0xf>(v);
    if (ROWS16) return __builtin_amdgcn_readlane(v, 15);
    v = dpp_imin<0x142, 0xa>(v);
    v = dpp_imin<0x143, 0xc>(v);
    return __builtin_amdgcn_readlane(v, 63);
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// One update of the factorisation, a(i,j) -= a_kj * a_ik (dense.rs:151): the file is compiled with -ffp-contract=off, so this is
// a multiply then a subtract -- the reference's arithmetic. (Round 2 also carried an FMA-contracted `fast` variant of the LU: it
// was 13 % faster -- the trailing update is bound by LDS operand delivery, not by the arithmetic --, changed the step sequence of
// one system in nine, and was removed in round 3.)
__device__ __forceinline__ double upd(double a, double u, double l) { return a - u * l; }

// Raw buffer loads / stores: a per-lane 32-bit byte offset plus a scalar byte offset, no 64-bit vector address arithmetic; a
// per-lane offset at or beyond the descriptor's size reads +0.0 without touching memory (range check on the vector offset).
__device__ __forceinline__ double buf_load_f64(__amdgpu_buffer_rsrc_t rsrc, unsigned voffset, int soffset) {
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    const v2u r = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)voffset, soffset, 0);
    return __hiloint2double((int)r.y, (int)r.x);
}

__device__ __forceinline__ void buf_store_f64(__amdgpu_buffer_rsrc_t rsrc, unsigned voffset, int soffset, double v) {
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    v2u d;
    d.x = (unsigned)__double2loint(v);
    d.y = (unsigned)__double2hiint(v);
    __builtin_amdgcn_raw_buffer_store_b64(d, rsrc, (int)voffset, soffset, 0);
}

__device__ __forceinline__ double shfl_xor_f64(double v, int mask) {
    const int lo = __shfl_xor(__double2loint(v), mask);
    const int hi = __shfl_xor(__double2hiint(v), mask);
    return __hiloint2double(hi, lo);
}

}  // namespace idahip
