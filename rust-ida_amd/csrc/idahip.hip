// libidahip.so -- implementation of include/ida_hip.h (single translation unit; gfx950; -ffp-contract=off).
#include "common.hpp"
#include "lu_kernels.hpp"
#include "lu_driver.hpp"
#include "problem_kernels.hpp"
#include "solve_kernels.hpp"
#include "vector_kernels.hpp"
#include "tiny_ida.hpp"
#include "round_ida.hpp"

using namespace idahip;

namespace {

template <class T>
int dalloc(idahip_ctx* c, T** p, size_t count) {
    *p = nullptr;
    if (count == 0) return 0;
    IDAHIP_HIP(c, hipMalloc((void**)p, count * sizeof(T)));
    return 0;
}

int check_list(idahip_ctx* c, const int32_t* hIdx, int nsys) {
    if (!c) return -1;
    if (nsys < 0 || nsys > c->batch) return fail(c, -2, "nsys = %d out of range (batch = %d)", nsys, c->batch);
    if (nsys > 0 && !hIdx) return fail(c, -2, "null system list");
    for (int s = 0; s < nsys; ++s)
        if (hIdx[s] < 0 || hIdx[s] >= c->batch) return fail(c, -2, "system id %d out of range at list position %d", hIdx[s], s);
    return 0;
}

double* field_ptr(idahip_ctx* c, idahip_field f) {
    switch (f) {
        case IDAHIP_F_YY: return c->yy;
        case IDAHIP_F_YP: return c->yp;
        case IDAHIP_F_YYPREDICT: return c->yypredict;
        case IDAHIP_F_YPPREDICT: return c->yppredict;
        case IDAHIP_F_EWT: return c->ewt;
        case IDAHIP_F_EE: return c->ee;
        case IDAHIP_F_DELTA: return c->delta;
        case IDAHIP_F_SAVRES: return c->savres;
        default: break;
    }
    if (f >= IDAHIP_F_PHI0 && f <= IDAHIP_F_PHI5) return c->phi + (size_t)(f - IDAHIP_F_PHI0) * c->batch * c->n;
    return nullptr;
}

VecState vec_state(idahip_ctx* c) {
    VecState s;
    s.phi = c->phi;
    s.phistride = (long)c->batch * c->n;
    s.yy = c->yy; s.yp = c->yp; s.yypredict = c->yypredict; s.yppredict = c->yppredict;
    s.ewt = c->ewt; s.ee = c->ee; s.delta = c->delta;
    s.n = c->n;
    s.rtol = c->rtol; s.atol_s = c->atol_s; s.atol_v = c->d_atol_v;
    return s;
}

int post_launch(idahip_ctx* c, const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(c, -101, "launch of %s failed: %s", what, hipGetErrorString(e));
    return 0;
}

}  // namespace

extern "C" {

int idahip_create(idahip_ctx** out, int device, int n, int batch, idahip_problem kind, void* hip_stream) {
    if (!out) return -1;
    *out = nullptr;
    if (n < 1 || batch < 1) return -2;
    if ((int)kind < 0 || (int)kind > (int)IDAHIP_HOST_CALLBACK) return -2;
    if ((kind == IDAHIP_ROBERTS || kind == IDAHIP_LORENZ63) && n != 3) return -2;
    idahip_ctx* c = new idahip_ctx();
    c->device = device; c->n = n; c->batch = batch; c->kind = kind;
    c->npad16 = (n + 15) & ~15;
    c->lu_superpanel = kind == IDAHIP_HEAT1D ? 1 : 0;
    if (const char* sp = std::getenv("IDAHIP_LU_SUPERPANEL")) c->lu_superpanel = std::strtol(sp, nullptr, 10) != 0 ? 1 : 0;
    if (const char* sp = std::getenv("IDAHIP_LU_PERIOD")) {
        const int v = (int)std::strtol(sp, nullptr, 10);
        if (v >= 1 && v <= 64) c->lu_period = v;
    }
    int ndev = 0;
    if (device < 0 || hipGetDeviceCount(&ndev) != hipSuccess || device >= ndev) { delete c; return -100; }
    DevGuard dev_guard__(device);  // allocations, stream and events below belong to `device`; the caller's device is restored
    if (hipSetDevice(device) != hipSuccess) { delete c; return -100; }
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) c->simds = 4 * cus;
    }
    if (hip_stream) {
        c->stream = (hipStream_t)hip_stream;
    } else {
        if (hipStreamCreate(&c->stream) != hipSuccess) { delete c; return -100; }
        c->own_stream = true;
    }
    const size_t bn = (size_t)batch * n, bnn = bn * n;
    int rc = 0;
    rc |= dalloc(c, &c->yy, bn); rc |= dalloc(c, &c->yp, bn); rc |= dalloc(c, &c->yypredict, bn); rc |= dalloc(c, &c->yppredict, bn);
    rc |= dalloc(c, &c->ewt, bn); rc |= dalloc(c, &c->ee, bn); rc |= dalloc(c, &c->delta, bn); rc |= dalloc(c, &c->savres, bn);
    rc |= dalloc(c, &c->phi, (size_t)MXORDP1 * bn);
    rc |= dalloc(c, &c->lu, bnn); rc |= dalloc(c, &c->piv, bn); rc |= dalloc(c, &c->perm, bn);
    rc |= dalloc(c, &c->lu_info, (size_t)batch);
    rc |= dalloc(c, &c->lu_redo, (size_t)batch);
    rc |= dalloc(c, &c->lu_nzb, (size_t)batch);
    if (n > LU_MAX_N) rc |= dalloc(c, &c->lu_bz, (size_t)batch * 64);
    if (n >= LU_ZMAP_MIN_N) rc |= dalloc(c, &c->lu_zmap, (size_t)batch * 4096);
    if (n >= LU_ZMAP_MIN_N) rc |= dalloc(c, &c->lu_dirty, (size_t)batch * 4096);
    if (n >= LU_ZMAP_MIN_N) rc |= dalloc(c, &c->lu_jwzero, (size_t)batch);
    if (n > TINY_N) {
        rc |= dalloc(c, &c->jw, bnn);
        rc |= dalloc(c, &c->lu_pos, bn); rc |= dalloc(c, &c->lu_live, bn); rc |= dalloc(c, &c->lu_prow, bn);
        rc |= dalloc(c, &c->lu_l11, (size_t)batch * L11_STRIDE);
    }
    if (kind == IDAHIP_LINEAR_DENSE) {
        rc |= dalloc(c, &c->A, bnn); rc |= dalloc(c, &c->B, bnn); rc |= dalloc(c, &c->C, bn);
    }
    if (kind == IDAHIP_LORENZ63) { c->nparam = 3; rc |= dalloc(c, &c->params, (size_t)batch * 3); }
    if (kind == IDAHIP_HEAT1D) { c->nparam = 1; rc |= dalloc(c, &c->params, (size_t)batch); }
    if (kind == IDAHIP_HOST_CALLBACK) rc |= dalloc(c, &c->cb_stage, 3 * bn);
    c->slot_cap = (size_t)batch * 256 + 4096;  // per call: <= ~110 B of scalars per system (predict) + list ids + results
    for (int i = 0; i < NSLOT && !rc; ++i) {
        if (hipHostMalloc((void**)&c->slots[i].h, c->slot_cap) != hipSuccess) rc = -100;
        else if (hipMalloc((void**)&c->slots[i].d, c->slot_cap) != hipSuccess) rc = -100;
        else if (hipEventCreateWithFlags(&c->slots[i].done, hipEventDisableTiming) != hipSuccess) rc = -100;
    }
    if (!rc && (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess || hipEventCreate(&c->ev2) != hipSuccess ||
                hipEventCreate(&c->ev3) != hipSuccess || hipEventCreateWithFlags(&c->ev_cnt, hipEventDisableTiming) != hipSuccess))
        rc = -100;

    if (rc) {
        idahip_destroy(c);
        return -100;
    }
    // deterministic initial contents
    (void)hipMemsetAsync(c->ee, 0, bn * sizeof(double), c->stream);
    (void)hipMemsetAsync(c->delta, 0, bn * sizeof(double), c->stream);
    (void)hipMemsetAsync(c->phi, 0, (size_t)MXORDP1 * bn * sizeof(double), c->stream);
    (void)hipMemsetAsync(c->lu_info, 0, (size_t)batch * sizeof(int), c->stream);
    (void)hipMemsetAsync(c->lu_redo, 0, (size_t)batch * sizeof(int), c->stream);
    if (c->lu_dirty) {  // the factors start as zeros, and so does the map of their blocks that have ever held anything else
        (void)hipMemsetAsync(c->lu, 0, bnn * sizeof(double), c->stream);
        (void)hipMemsetAsync(c->lu_dirty, 0, (size_t)batch * 4096, c->stream);
        (void)hipMemsetAsync(c->lu_jwzero, 0, (size_t)batch * sizeof(int), c->stream);
    }
    (void)hipStreamSynchronize(c->stream);
    *out = c;
    return 0;
}

int idahip_destroy(idahip_ctx* c) {
    DevGuard dev_guard__(c);
    if (!c) return 0;
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    void* ptrs[] = {c->yy, c->yp, c->yypredict, c->yppredict, c->ewt, c->ee, c->delta, c->savres, c->phi, c->lu, c->jw, c->piv, c->perm,
                    c->lu_pos, c->lu_live, c->lu_prow, c->lu_info, c->lu_redo, c->lu_nzb, c->lu_bz, c->lu_zmap, c->lu_dirty, c->lu_jwzero, c->lu_l11, c->params, c->A, c->B, c->C, c->d_atol_v, c->ic_y,
                    c->ic_yp, c->dky, c->cb_stage, c->tiny_sys, c->tiny_touts, c->tiny_yout, c->tiny_ypout, c->tiny_start, c->tiny_rounds,
                    c->tiny_acc, c->tiny_roots, c->rnd_i, c->rnd_d};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (c->rnd_host) (void)hipHostFree(c->rnd_host);
    if (c->cb_jpin) (void)hipHostFree(c->cb_jpin);
    if (c->cb_jdev) (void)hipFree(c->cb_jdev);
    for (int i = 0; i < NSLOT; ++i) {
        if (c->slots[i].h) (void)hipHostFree(c->slots[i].h);
        if (c->slots[i].d) (void)hipFree(c->slots[i].d);
        if (c->slots[i].done) (void)hipEventDestroy(c->slots[i].done);
    }
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->ev2) (void)hipEventDestroy(c->ev2);
    if (c->ev3) (void)hipEventDestroy(c->ev3);
    if (c->ev_cnt) (void)hipEventDestroy(c->ev_cnt);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

const char* idahip_last_error(const idahip_ctx* c) { return c ? c->err.c_str() : "null ctx"; }
int idahip_n(const idahip_ctx* c) { return c ? c->n : -1; }
int idahip_batch(const idahip_ctx* c) { return c ? c->batch : -1; }

int idahip_sync(idahip_ctx* c) {
    DevGuard dev_guard__(c);
    if (!c) return -1;
    IDAHIP_HIP(c, hipStreamSynchronize(c->stream));
    return 0;
}

// ------------------------------------------------------------------------------------------------ streams that really run side by side
// A HIP stream is not a hardware queue: the runtime multiplexes its streams onto a few of them (four by default) and which
// stream lands on which is its business -- the kernel trace of two builds of the same program showed three of four streams on
// ONE queue, their launches strictly one after the other (tools/group_trace.sh). Ensembles that are meant
// to fill each other's idle stretches (idaens_stream_group) need streams the device runs CONCURRENTLY, so this entry point
// finds them by experiment: it creates candidate streams and lets a probe kernel (one wavefront that waits 300 us on the
// constant 100 MHz clock and notes when it started and ended) run on the candidate and on every stream chosen so far at once;
// the candidate is kept if its interval overlaps all of theirs. Rejected candidates stay alive until the end (a destroyed
// stream gives its queue back and the next one would take the same) and are destroyed then.
__global__ void stream_probe_kernel(unsigned long long* out, long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while ((long long)(wall_clock64() - t0) < ticks) __builtin_amdgcn_s_sleep(16);  // (ends by itself: the clock runs on)
    if (threadIdx.x == 0) {
        out[0] = t0;
        out[1] = wall_clock64();
    }
}

int idahip_concurrent_streams(int device, int count, void** streams_out, int* nconcurrent) {
    if (!streams_out || count < 1 || count > 64) return -2;
    int ndev = 0;
    if (device < 0 || hipGetDeviceCount(&ndev) != hipSuccess || device >= ndev) return -100;
    DevGuard dev_guard__(device);
    if (hipSetDevice(device) != hipSuccess) return -100;
    unsigned long long* stamps = nullptr;
    if (hipHostMalloc((void**)&stamps, sizeof(unsigned long long) * 2 * (size_t)(count + 1)) != hipSuccess) return -100;
    std::vector<hipStream_t> sel, rejected;
    int rc = 0;
    const long long ticks = 30000;  // 300 us at 100 MHz
    for (int attempt = 0; attempt < 6 * count + 8 && (int)sel.size() < count && !rc; ++attempt) {
        hipStream_t s = nullptr;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { rc = -100; break; }
        if (sel.empty()) {
            sel.push_back(s);
            continue;
        }
        const int k = (int)sel.size();
        for (int i = 0; i <= k; ++i) stamps[2 * i] = stamps[2 * i + 1] = 0ull;
        for (int i = 0; i < k; ++i) hipLaunchKernelGGL(stream_probe_kernel, dim3(1), dim3(64), 0, sel[i], stamps + 2 * i, ticks);
        hipLaunchKernelGGL(stream_probe_kernel, dim3(1), dim3(64), 0, s, stamps + 2 * k, ticks);
        bool ok = hipGetLastError() == hipSuccess;
        for (int i = 0; i < k; ++i) ok = (hipStreamSynchronize(sel[i]) == hipSuccess) && ok;
        ok = (hipStreamSynchronize(s) == hipSuccess) && ok;
        if (!ok) { rc = -100; (void)hipStreamDestroy(s); break; }
        bool overlaps_all = stamps[2 * k + 1] > stamps[2 * k];
        for (int i = 0; i < k; ++i)
            overlaps_all = overlaps_all && stamps[2 * k] < stamps[2 * i + 1] && stamps[2 * i] < stamps[2 * k + 1];
        (overlaps_all ? sel : rejected).push_back(s);
    }
    const int found = (int)sel.size();
    // fewer hardware queues than asked for: the remaining streams are ordinary ones (they share a queue with somebody)
    while (!rc && (int)sel.size() < count) {
        if (!rejected.empty()) {
            sel.push_back(rejected.back());
            rejected.pop_back();
        } else {
            hipStream_t s = nullptr;
            if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) rc = -100;
            else sel.push_back(s);
        }
    }
    for (hipStream_t s : rejected) (void)hipStreamDestroy(s);
    (void)hipHostFree(stamps);
    if (rc) {
        for (hipStream_t s : sel) (void)hipStreamDestroy(s);
        return rc;
    }
    for (int i = 0; i < count; ++i) streams_out[i] = (void*)sel[i];
    if (nconcurrent) *nconcurrent = found;
    return 0;
}

// Diagnostic (tools/half_streams.py): how evenly the device shares itself between two streams whose kernels each want all of it.
// A grid of 2048 workgroups that hold 64 KB of LDS each (two fit a CU: 512 resident) and wait 20 us is launched on A and, right
// behind it, on B; every workgroup notes its start and end. *frac = the part of the two kernels' joint span in which both had
// workgroups running: ~1 when the dispatcher interleaves the two grids, ~0.5 when B's workgroups only start once A's are all out.
__global__ void stream_share_kernel(unsigned long long* out, long long ticks) {
    extern __shared__ unsigned char pad_lds[];
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) pad_lds[0] = 1;  // (keeps the allocation)
    while ((long long)(wall_clock64() - t0) < ticks) __builtin_amdgcn_s_sleep(16);
    if (threadIdx.x == 0) {
        atomicMin(out, t0);
        atomicMax(out + 1, wall_clock64());
    }
}

int idahip_stream_pair_share(int device, void* streamA, void* streamB, double* frac) {
    if (!streamA || !streamB || !frac) return -2;
    DevGuard dev_guard__(device);
    unsigned long long* st = nullptr;
    if (hipHostMalloc((void**)&st, 4 * sizeof(unsigned long long)) != hipSuccess) return -100;
    st[0] = st[2] = ~0ull;
    st[1] = st[3] = 0ull;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)stream_share_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        attr_set = true;
    }
    hipLaunchKernelGGL(stream_share_kernel, dim3(2048), dim3(64), 65536, (hipStream_t)streamA, st, 2000ll);
    hipLaunchKernelGGL(stream_share_kernel, dim3(2048), dim3(64), 65536, (hipStream_t)streamB, st + 2, 2000ll);
    int rc = hipGetLastError() == hipSuccess ? 0 : -100;
    if (hipStreamSynchronize((hipStream_t)streamA) != hipSuccess) rc = -100;
    if (hipStreamSynchronize((hipStream_t)streamB) != hipSuccess) rc = -100;
    if (!rc) {
        const double lo = (double)(st[0] < st[2] ? st[0] : st[2]), hi = (double)(st[1] > st[3] ? st[1] : st[3]);
        const double olo = (double)(st[0] > st[2] ? st[0] : st[2]), ohi = (double)(st[1] < st[3] ? st[1] : st[3]);
        *frac = hi > lo ? (ohi > olo ? (ohi - olo) / (hi - lo) : 0.0) : 0.0;
    }
    (void)hipHostFree(st);
    return rc;
}

int idahip_release_streams(int device, int count, void** streams) {
    if (!streams || count < 0) return -2;
    DevGuard dev_guard__(device);
    int rc = 0;
    for (int i = 0; i < count; ++i)
        if (streams[i] && hipStreamDestroy((hipStream_t)streams[i]) != hipSuccess) rc = -100;
    return rc;
}

int idahip_set_tolerances(idahip_ctx* c, double rtol, const double* hAtol, int natol) {
    DevGuard dev_guard__(c);
    if (!c || !hAtol) return -1;
    if (natol != 1 && natol != c->n) return fail(c, -2, "natol must be 1 or n");
    c->rtol = rtol;
    if (natol == 1 && c->n != 1) {
        c->atol_s = hAtol[0];
        if (c->d_atol_v) { (void)hipFree(c->d_atol_v); c->d_atol_v = nullptr; }
    } else {
        c->atol_s = hAtol[0];
        if (!c->d_atol_v) { int rc = dalloc(c, &c->d_atol_v, (size_t)c->n); if (rc) return rc; }
        IDAHIP_HIP(c, hipMemcpy(c->d_atol_v, hAtol, sizeof(double) * c->n, hipMemcpyHostToDevice));
    }
    return 0;
}

int idahip_set_problem_params(idahip_ctx* c, int first, int count, const double* hParams, int nparam) {
    DevGuard dev_guard__(c);
    if (!c || !hParams) return -1;
    if (!c->params || nparam != c->nparam) return fail(c, -2, "problem kind takes %d parameters per system", c->nparam);
    if (first < 0 || count < 0 || first + count > c->batch) return fail(c, -2, "system range out of bounds");
    IDAHIP_HIP(c, hipMemcpy(c->params + (size_t)first * nparam, hParams, sizeof(double) * count * nparam, hipMemcpyHostToDevice));
    return 0;
}

int idahip_set_linear_dense(idahip_ctx* c, int first, int count, const double* hA, const double* hB, const double* hC) {
    DevGuard dev_guard__(c);
    if (!c || !hA || !hB || !hC) return -1;
    if (c->kind != IDAHIP_LINEAR_DENSE) return fail(c, -2, "not a LINEAR_DENSE ctx");
    if (first < 0 || count < 0 || first + count > c->batch) return fail(c, -2, "system range out of bounds");
    const size_t nn = (size_t)c->n * c->n;
    IDAHIP_HIP(c, hipMemcpy(c->A + first * nn, hA, sizeof(double) * count * nn, hipMemcpyHostToDevice));
    IDAHIP_HIP(c, hipMemcpy(c->B + first * nn, hB, sizeof(double) * count * nn, hipMemcpyHostToDevice));
    IDAHIP_HIP(c, hipMemcpy(c->C + (size_t)first * c->n, hC, sizeof(double) * count * c->n, hipMemcpyHostToDevice));
    return 0;
}

int idahip_set_host_problem(idahip_ctx* c, idahip_res_fn res, idahip_jac_fn jac, void* user) {
    if (!c || !res || !jac) return -1;
    if (c->kind != IDAHIP_HOST_CALLBACK) return fail(c, -2, "not an IDAHIP_HOST_CALLBACK ctx");
    c->cb_res = res;
    c->cb_jac = jac;
    c->cb_user = user;
    return 0;
}

int idahip_upload(idahip_ctx* c, idahip_field f, int first, int count, const double* h) {
    DevGuard dev_guard__(c);
    if (!c || !h) return -1;
    double* d = field_ptr(c, f);
    if (!d) return fail(c, -2, "unknown field %d", (int)f);
    if (first < 0 || count < 0 || first + count > c->batch) return fail(c, -2, "system range out of bounds");
    IDAHIP_HIP(c, hipStreamSynchronize(c->stream));
    IDAHIP_HIP(c, hipMemcpy(d + (size_t)first * c->n, h, sizeof(double) * count * c->n, hipMemcpyHostToDevice));
    return 0;
}

int idahip_download(idahip_ctx* c, idahip_field f, int first, int count, double* h) {
    DevGuard dev_guard__(c);
    if (!c || !h) return -1;
    double* d = field_ptr(c, f);
    if (!d) return fail(c, -2, "unknown field %d", (int)f);
    if (first < 0 || count < 0 || first + count > c->batch) return fail(c, -2, "system range out of bounds");
    IDAHIP_HIP(c, hipStreamSynchronize(c->stream));
    IDAHIP_HIP(c, hipMemcpy(h, d + (size_t)first * c->n, sizeof(double) * count * c->n, hipMemcpyDeviceToHost));
    return 0;
}

int idahip_download_lu(idahip_ctx* c, int sys, double* hLU, int64_t* hPiv) {
    DevGuard dev_guard__(c);
    if (!c) return -1;
    if (sys < 0 || sys >= c->batch) return fail(c, -2, "system out of range");
    IDAHIP_HIP(c, hipStreamSynchronize(c->stream));
    const size_t nn = (size_t)c->n * c->n;
    if (hLU) IDAHIP_HIP(c, hipMemcpy(hLU, c->lu + sys * nn, sizeof(double) * nn, hipMemcpyDeviceToHost));
    if (hPiv) IDAHIP_HIP(c, hipMemcpy(hPiv, c->piv + (size_t)sys * c->n, sizeof(int64_t) * c->n, hipMemcpyDeviceToHost));
    return 0;
}

void* idahip_dev_alloc(idahip_ctx* c, size_t bytes) {
    DevGuard dev_guard__(c);
    void* p = nullptr;
    if (!c || hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    return p;
}
int idahip_dev_free(idahip_ctx* c, void* d) {
    DevGuard dev_guard__(c);
    if (!c) return -1;
    IDAHIP_HIP(c, hipStreamSynchronize(c->stream));
    IDAHIP_HIP(c, hipFree(d));
    return 0;
}
int idahip_memcpy_h2d(idahip_ctx* c, void* d, const void* h, size_t bytes) {
    DevGuard dev_guard__(c);
    if (!c) return -1;
    IDAHIP_HIP(c, hipStreamSynchronize(c->stream));
    IDAHIP_HIP(c, hipMemcpy(d, h, bytes, hipMemcpyHostToDevice));
    return 0;
}
int idahip_memcpy_d2h(idahip_ctx* c, void* h, const void* d, size_t bytes) {
    DevGuard dev_guard__(c);
    if (!c) return -1;
    IDAHIP_HIP(c, hipStreamSynchronize(c->stream));
    IDAHIP_HIP(c, hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost));
    return 0;
}

// ------------------------------------------------------------------------------------------------ LSolver
int idahip_ls_setup(idahip_ctx* c, double* dA, int64_t* dPiv, int32_t* hInfo, const int32_t* hIdx, int nsys) {
    DevGuard dev_guard__(c);
    int rc = check_list(c, hIdx, nsys);
    if (rc) return rc;
    if (!dA || !dPiv || !hInfo) return fail(c, -2, "null argument");
    if (nsys == 0) return 0;
    const int n = c->n;
    const long nn = (long)n * n;
    ArgPack ap;
    if ((rc = ap.begin(c))) return rc;
    const int* d_idx = ap.in(hIdx, nsys);
    if ((rc = ap.upload())) return rc;
    {
        KTimer kt(c, IDAHIP_K_LU, nsys);
        if (n <= TINY_N) {
            rc = lu_factor_batched(c, dA, nn, dA, nn, (long long*)dPiv, n, nullptr, d_idx, nsys);
        } else {
            // factor in place in dA (physical row order), scatter rows into the ctx work matrix, copy back
            rc = lu_factor_batched(c, dA, nn, c->jw, nn, (long long*)dPiv, n, nullptr, d_idx, nsys);
            if (!rc) {
                for (int s = 0; s < nsys; ++s)  // listed systems only; info != 0 systems keep their partial factors
                    (void)hipMemcpyAsync(dA + hIdx[s] * nn, c->jw + hIdx[s] * nn, sizeof(double) * nn, hipMemcpyDeviceToDevice, c->stream);
            }
        }
        if (rc) return rc;
        if ((rc = post_launch(c, "lu"))) return rc;
    }
    std::vector<int32_t> info(c->batch);
    IDAHIP_HIP(c, hipMemcpyAsync(info.data(), c->lu_info, sizeof(int32_t) * c->batch, hipMemcpyDeviceToHost, c->stream));
    IDAHIP_HIP(c, hipStreamSynchronize(c->stream));
    int any = 0;
    for (int s = 0; s < nsys; ++s) {
        hInfo[s] = info[hIdx[s]];
        any |= hInfo[s] != 0;
    }
    return any ? 1 : 0;
}

int idahip_ls_solve(idahip_ctx* c, const double* dLU, const int64_t* dPiv, double* dX, const double* dB, double /*tol*/,
                    const int32_t* hIdx, int nsys) {
    DevGuard dev_guard__(c);
    int rc = check_list(c, hIdx, nsys);
    if (rc) return rc;
    if (!dLU || !dPiv || !dX || !dB) return fail(c, -2, "null argument");
    if (nsys == 0) return 0;
    const int n = c->n;
    ArgPack ap;
    if ((rc = ap.begin(c))) return rc;
    const int* d_idx = ap.in(hIdx, nsys);
    if ((rc = ap.upload())) return rc;
    {
        KTimer kt(c, IDAHIP_K_SOLVE, nsys);
        if (n <= TINY_N) {
            hipLaunchKernelGGL(tiny_solve_kernel, dim3((nsys + 63) / 64), dim3(64), 0, c->stream, dLU, (long)n * n, (const long long*)dPiv,
                               (long)n, dX, dB, n, d_idx, nsys);
        } else {
            const size_t shm = sizeof(double) * n + sizeof(int) * n;
            if (n % 2 == 0)
                hipLaunchKernelGGL(ls_solve_kernel<2>, dim3(nsys), dim3(256), shm, c->stream, dLU, (long)n * n, (const long long*)dPiv,
                                   (long)n, (const int*)nullptr, dX, dB, n, d_idx);
            else
                hipLaunchKernelGGL(ls_solve_kernel<1>, dim3(nsys), dim3(256), shm, c->stream, dLU, (long)n * n, (const long long*)dPiv,
                                   (long)n, (const int*)nullptr, dX, dB, n, d_idx);
        }
        if ((rc = post_launch(c, "ls_solve"))) return rc;
    }
    return ap.finish_async();
}

int idahip_wrms(idahip_ctx* c, const double* dX, const double* dW, double* hOut, const int32_t* hIdx, int nsys) {
    DevGuard dev_guard__(c);
    int rc = check_list(c, hIdx, nsys);
    if (rc) return rc;
    if (!dX || !dW || !hOut) return fail(c, -2, "null argument");
    if (nsys == 0) return 0;
    const int n = c->n;
    ArgPack ap;
    if ((rc = ap.begin(c))) return rc;
    const int* d_idx = ap.in(hIdx, nsys);
    double* d_out = ap.out<double>(nsys);
    if ((rc = ap.ok())) return rc;
    if ((rc = ap.upload())) return rc;
    {
        KTimer kt(c, IDAHIP_K_VECTOR, nsys);
        if (n <= TINY_N)
            hipLaunchKernelGGL(tiny_wrms_kernel, dim3((nsys + 63) / 64), dim3(64), 0, c->stream, dX, dW, d_out, n, d_idx, nsys);
        else
            hipLaunchKernelGGL(wrms_kernel, dim3(nsys), dim3(256), sizeof(double) * n, c->stream, dX, dW, d_out, n, d_idx);
        if ((rc = post_launch(c, "wrms"))) return rc;
    }
    if ((rc = ap.fetch())) return rc;
    const double* h = ap.host_of(d_out);
    for (int s = 0; s < nsys; ++s) hOut[s] = sqrt(h[s] / (double)n);  // divide, then sqrt: host libm (norm_rms.rs:36-37)
    return 0;
}

// ------------------------------------------------------------------------------------------------ NLProblem
namespace {

// residual kernels of IdaNLProblem::sys; jac_out != nullptr (linear dense, column-major work matrix only) also forms J
// residual of a host-callback problem: device forms and packs yy, yp; host calls the user's res per listed system; device
// scatters the residuals (launch_sys with c->kind == IDAHIP_HOST_CALLBACK)
int callback_sys(idahip_ctx* c, const SysArgs& a, const double* hTn, const int32_t* hIdx, int nsys) {
    const int n = c->n;
    if (!c->cb_res || !c->cb_jac) return fail(c, -2, "IDAHIP_HOST_CALLBACK: idahip_set_host_problem has not been called");
    const size_t cnt = (size_t)nsys * 3 * n;
    hipLaunchKernelGGL(callback_pre_kernel, dim3(nsys), dim3(256), 0, c->stream, a, c->cb_stage);
    if (c->cb_host.size() < cnt) c->cb_host.resize(cnt);
    double* h = c->cb_host.data();
    IDAHIP_HIP(c, hipMemcpyAsync(h, c->cb_stage, sizeof(double) * cnt, hipMemcpyDeviceToHost, c->stream));
    IDAHIP_HIP(c, hipStreamSynchronize(c->stream));
    for (int s = 0; s < nsys; ++s) {
        double* hs = h + (size_t)s * 3 * n;
        if (c->cb_res(hIdx[s], hTn[s], hs, hs + n, hs + 2 * n, c->cb_user) != 0)
            return fail(c, -7, "the user's residual function failed for system %d", hIdx[s]);
    }
    IDAHIP_HIP(c, hipMemcpyAsync(c->cb_stage, h, sizeof(double) * cnt, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(callback_post_kernel, dim3(nsys), dim3(256), 0, c->stream, a, (const double*)c->cb_stage);
    IDAHIP_HIP(c, hipStreamSynchronize(c->stream));  // cb_host is reused by the next call
    return post_launch(c, "nls_sys (host callback)");
}

// Jacobian of a host-callback problem into the LU work matrices (column-major per system)
int callback_jac(idahip_ctx* c, double* work, const int* d_idx, const double* hTn, const double* hCj, const int32_t* hIdx, int nsys) {
    const int n = c->n;
    const size_t nn = (size_t)n * n;
    if (!c->cb_res || !c->cb_jac) return fail(c, -2, "IDAHIP_HOST_CALLBACK: idahip_set_host_problem has not been called");
    const size_t cnt = (size_t)nsys * 3 * n;
    hipLaunchKernelGGL(callback_pack_kernel, dim3(nsys), dim3(256), 0, c->stream, (const double*)c->yy, (const double*)c->yp,
                       (const double*)c->savres, d_idx, n, c->cb_stage);
    if (c->cb_host.size() < cnt) c->cb_host.resize(cnt);
    double* h = c->cb_host.data();
    // The Jacobians go up a chunk at a time: the user's function fills a pinned buffer system by system, ONE asynchronous copy
    // takes the chunk to a device staging buffer and one kernel scatters it into the listed systems' work matrices (a pageable
    // hipMemcpy per system on the null stream before: O(nsys) blocking copies per setup). A chunk is <= 64 MB.
    const size_t chunk = std::max<size_t>(1, std::min<size_t>((size_t)nsys, ((size_t)64 << 20) / (nn * sizeof(double))));
    if (c->cb_jcap < chunk) {
        if (c->cb_jpin) (void)hipHostFree(c->cb_jpin);
        if (c->cb_jdev) (void)hipFree(c->cb_jdev);
        c->cb_jpin = nullptr; c->cb_jdev = nullptr; c->cb_jcap = 0;
        if (hipHostMalloc((void**)&c->cb_jpin, chunk * nn * sizeof(double)) != hipSuccess) { c->cb_jpin = nullptr; return fail(c, -100, "pinned staging for %zu Jacobians", chunk); }
        if (hipMalloc((void**)&c->cb_jdev, chunk * nn * sizeof(double)) != hipSuccess) {
            (void)hipHostFree(c->cb_jpin);
            c->cb_jpin = nullptr; c->cb_jdev = nullptr;
            return fail(c, -100, "device staging for %zu Jacobians", chunk);
        }
        c->cb_jcap = chunk;
    }
    IDAHIP_HIP(c, hipMemcpyAsync(h, c->cb_stage, sizeof(double) * cnt, hipMemcpyDeviceToHost, c->stream));
    IDAHIP_HIP(c, hipStreamSynchronize(c->stream));
    for (size_t s0 = 0; s0 < (size_t)nsys; s0 += chunk) {
        const size_t m = std::min(chunk, (size_t)nsys - s0);
        for (size_t q = 0; q < m; ++q) {
            const size_t s = s0 + q;
            const double* hs = h + s * 3 * n;
            double* J = c->cb_jpin + q * nn;
            for (size_t e = 0; e < nn; ++e) J[e] = 0.0;  // J <- 0 (ida_ls.rs:255)
            if (c->cb_jac(hIdx[s], hTn[s], hCj[s], hs, hs + n, hs + 2 * n, J, c->cb_user) != 0)
                return fail(c, -7, "the user's Jacobian function failed for system %d", hIdx[s]);
        }
        IDAHIP_HIP(c, hipMemcpyAsync(c->cb_jdev, c->cb_jpin, sizeof(double) * m * nn, hipMemcpyHostToDevice, c->stream));
        const int gy = (int)std::min<size_t>(64, (nn + 255) / 256);
        hipLaunchKernelGGL(callback_scatter_jac_kernel, dim3((unsigned)m, gy), dim3(256), 0, c->stream, (const double*)c->cb_jdev, work, d_idx + s0, (long)nn);
        if (s0 + chunk < (size_t)nsys) IDAHIP_HIP(c, hipStreamSynchronize(c->stream));  // the pinned buffer is filled again
    }
    return 0;
}

int launch_sys(idahip_ctx* c, const SysArgs& a, int nsys, double* jac_out, const double* hTn = nullptr, const int32_t* hIdx = nullptr) {
    const int n = c->n;
    switch (c->kind) {
        case IDAHIP_HOST_CALLBACK:
            return callback_sys(c, a, hTn, hIdx, nsys);
        case IDAHIP_ROBERTS:
            hipLaunchKernelGGL(tiny_sys_kernel<IDAHIP_ROBERTS>, dim3((nsys + 63) / 64), dim3(64), 0, c->stream, a, (const double*)nullptr, 0, nsys);
            break;
        case IDAHIP_LORENZ63:
            hipLaunchKernelGGL(tiny_sys_kernel<IDAHIP_LORENZ63>, dim3((nsys + 63) / 64), dim3(64), 0, c->stream, a, (const double*)c->params, 3, nsys);
            break;
        case IDAHIP_LINEAR_DENSE: {
            const size_t shm = 2 * sizeof(double) * n;
            const double *A = c->A, *B = c->B, *C = c->C;
            if (jac_out) {
                if (n % 2 == 0) hipLaunchKernelGGL((linear_sys_kernel<2, true>), dim3(nsys), dim3(256), shm, c->stream, a, A, B, C, jac_out);
                else hipLaunchKernelGGL((linear_sys_kernel<1, true>), dim3(nsys), dim3(256), shm, c->stream, a, A, B, C, jac_out);
            } else {
                if (n % 2 == 0) hipLaunchKernelGGL((linear_sys_kernel<2, false>), dim3(nsys), dim3(256), shm, c->stream, a, A, B, C, (double*)nullptr);
                else hipLaunchKernelGGL((linear_sys_kernel<1, false>), dim3(nsys), dim3(256), shm, c->stream, a, A, B, C, (double*)nullptr);
            }
            break;
        }
        case IDAHIP_HEAT1D:
            hipLaunchKernelGGL(heat_sys_kernel, dim3(nsys), dim3(256), sizeof(double) * n, c->stream, a, (const double*)c->params);
            break;
    }
    return post_launch(c, "nls_sys");
}

// Jacobian kernels of IdaNLProblem::setup (jac at the current yy, yp, cj) into the LU work matrix
int launch_jac(idahip_ctx* c, double* work, const int* d_idx, const double* d_cj, int nsys, const double* hTn = nullptr,
               const double* hCj = nullptr, const int32_t* hIdx = nullptr, const int* d_skip = nullptr) {
    const int n = c->n;
    const long nn = (long)n * n;
    switch (c->kind) {
        case IDAHIP_HOST_CALLBACK:
            return callback_jac(c, work, d_idx, hTn, hCj, hIdx, nsys);
        case IDAHIP_ROBERTS:
            hipLaunchKernelGGL(tiny_jac_kernel<IDAHIP_ROBERTS>, dim3((nsys + 63) / 64), dim3(64), 0, c->stream, work, (const double*)c->yy,
                               (const double*)nullptr, 0, d_idx, d_cj, nsys);
            break;
        case IDAHIP_LORENZ63:
            hipLaunchKernelGGL(tiny_jac_kernel<IDAHIP_LORENZ63>, dim3((nsys + 63) / 64), dim3(64), 0, c->stream, work, (const double*)c->yy,
                               (const double*)c->params, 3, d_idx, d_cj, nsys);
            break;
        case IDAHIP_LINEAR_DENSE: {
            int chunks = 1;
            while ((long)nsys * chunks < 2048 && chunks < 64) chunks *= 2;
            hipLaunchKernelGGL(linear_jac_kernel, dim3(nsys, chunks), dim3(256), 0, c->stream, work, (const double*)c->A,
                               (const double*)c->B, nn, d_idx, d_cj, chunks);
            break;
        }
        case IDAHIP_HEAT1D: {
            int chunks = 1;
            while ((long)nsys * chunks < 2048 && chunks < n) chunks *= 2;
            hipLaunchKernelGGL(heat_jac_kernel, dim3(nsys, chunks), dim3(256), 0, c->stream, work, n, (const double*)c->params, d_idx, d_cj, chunks, d_skip,
                               (work == c->jw) ? (const int*)c->lu_jwzero : nullptr);
            break;
        }
    }
    return post_launch(c, "jac");
}

// batched getrf of the work matrices into ctx->lu / piv / perm, then the per-system info of the listed systems
int factor_and_report(idahip_ctx* c, double* work, const int* d_idx, const int32_t* hIdx, int nsys, int32_t* hInfo) {
    const int n = c->n;
    const long nn = (long)n * n;
    int rc;
    {
        KTimer kt(c, IDAHIP_K_LU, nsys);
        rc = lu_factor_batched(c, work, nn, c->lu, nn, (long long*)c->piv, n, c->perm, d_idx, nsys);
        if (rc) return rc;
        if ((rc = post_launch(c, "lu"))) return rc;
    }
    std::vector<int32_t> info(c->batch);
    IDAHIP_HIP(c, hipMemcpyAsync(info.data(), c->lu_info, sizeof(int32_t) * c->batch, hipMemcpyDeviceToHost, c->stream));
    IDAHIP_HIP(c, hipStreamSynchronize(c->stream));
    int any = 0;
    for (int s = 0; s < nsys; ++s) {
        hInfo[s] = info[hIdx[s]];
        any |= hInfo[s] != 0;
    }
    return any ? 1 : 0;
}

void fill_sys_args(idahip_ctx* c, SysArgs& a, int reset_ee) {
    a.yypredict = c->yypredict; a.yppredict = c->yppredict; a.yy = c->yy; a.yp = c->yp; a.ee = c->ee; a.delta = c->delta;
    a.savres = c->savres; a.n = c->n; a.reset_ee = reset_ee;
}

}  // namespace

int idahip_nls_sys(idahip_ctx* c, const double* hTn, const double* hCj, int reset_ee, const int32_t* hIdx, int nsys) {
    DevGuard dev_guard__(c);
    int rc = check_list(c, hIdx, nsys);
    if (rc) return rc;
    if (!hTn || !hCj) return fail(c, -2, "null argument");
    if (nsys == 0) return 0;
    ArgPack ap;
    if ((rc = ap.begin(c))) return rc;
    SysArgs a;
    a.idx = ap.in(hIdx, nsys);
    a.tn = ap.in(hTn, nsys);
    a.cj = ap.in(hCj, nsys);
    if ((rc = ap.upload())) return rc;
    fill_sys_args(c, a, reset_ee);
    {
        KTimer kt(c, IDAHIP_K_SYS, nsys);
        if ((rc = launch_sys(c, a, nsys, nullptr, hTn, hIdx))) return rc;
    }
    return ap.finish_async();
}

int idahip_nls_lsetup(idahip_ctx* c, const double* hTn, const double* hCj, int32_t* hInfo, const int32_t* hIdx, int nsys) {
    DevGuard dev_guard__(c);
    int rc = check_list(c, hIdx, nsys);
    if (rc) return rc;
    if (!hTn || !hCj || !hInfo) return fail(c, -2, "null argument");
    if (nsys == 0) return 0;
    const int n = c->n;
    ArgPack ap;
    if ((rc = ap.begin(c))) return rc;
    const int* d_idx = ap.in(hIdx, nsys);
    const double* d_cj = ap.in(hCj, nsys);
    if ((rc = ap.upload())) return rc;
    double* work = (n <= TINY_N) ? c->lu : c->jw;
    {
        KTimer kt(c, IDAHIP_K_JAC, nsys);
        if ((rc = launch_jac(c, work, d_idx, d_cj, nsys, hTn, hCj, hIdx))) return rc;
    }
    return factor_and_report(c, work, d_idx, hIdx, nsys, hInfo);
}

int idahip_nls_sys_setup(idahip_ctx* c, const double* hTn, const double* hCj, int reset_ee, int32_t* hInfo, const int32_t* hIdx,
                         int nsys) {
    DevGuard dev_guard__(c);
    int rc = check_list(c, hIdx, nsys);
    if (rc) return rc;
    if (!hTn || !hCj || !hInfo) return fail(c, -2, "null argument");
    if (nsys == 0) return 0;
    const int n = c->n;
    ArgPack ap;
    if ((rc = ap.begin(c))) return rc;
    SysArgs a;
    a.idx = ap.in(hIdx, nsys);
    a.tn = ap.in(hTn, nsys);
    a.cj = ap.in(hCj, nsys);
    if ((rc = ap.upload())) return rc;
    fill_sys_args(c, a, reset_ee);
    double* work = (n <= TINY_N) ? c->lu : c->jw;
    // the linear dense residual sweeps A and B anyway: J = B + cj*A falls out of the same pass
    const bool fused = c->kind == IDAHIP_LINEAR_DENSE && n > TINY_N;
    {
        KTimer kt(c, fused ? IDAHIP_K_SYS_JAC : IDAHIP_K_SYS, nsys);
        if ((rc = launch_sys(c, a, nsys, fused ? work : nullptr, hTn, hIdx))) return rc;
    }
    if (!fused) {
        KTimer kt(c, IDAHIP_K_JAC, nsys);
        if ((rc = launch_jac(c, work, a.idx, a.cj, nsys, hTn, hCj, hIdx))) return rc;
    }
    return factor_and_report(c, work, a.idx, hIdx, nsys, hInfo);
}

// One Newton iteration body for the listed systems (shared by idahip_newton_iter and idahip_newton_iter2)
static int launch_newton_iter(idahip_ctx* c, const int* d_idx, const double* d_scale, double* d_out, const int* d_skip, int nsys) {
    const int n = c->n;
    if (n <= TINY_N) {
        hipLaunchKernelGGL(tiny_newton_iter_kernel, dim3((nsys + 63) / 64), dim3(64), 0, c->stream, (const double*)c->lu,
                           (const long long*)c->piv, c->delta, c->ee, (const double*)c->ewt, n, d_idx, nsys, d_scale, d_out, d_skip);
    } else if (n % 2 == 0 && n >= 2048) {
        hipLaunchKernelGGL((newton_iter_kernel<2, 1024>), dim3(nsys), dim3(1024), sizeof(double) * (n + (n > 4096 ? n : 4096)), c->stream, (const double*)c->lu,
                           (const int*)c->perm, c->delta, c->ee, (const double*)c->ewt, n, d_idx, d_scale, d_out, d_skip, (const unsigned char*)c->lu_zmap);
    } else if (n % 2 == 0) {
        hipLaunchKernelGGL(newton_iter_kernel<2>, dim3(nsys), dim3(256), 2 * sizeof(double) * n, c->stream, (const double*)c->lu,
                           (const int*)c->perm, c->delta, c->ee, (const double*)c->ewt, n, d_idx, d_scale, d_out, d_skip, (const unsigned char*)nullptr);
    } else {
        hipLaunchKernelGGL(newton_iter_kernel<1>, dim3(nsys), dim3(256), 2 * sizeof(double) * n, c->stream, (const double*)c->lu,
                           (const int*)c->perm, c->delta, c->ee, (const double*)c->ewt, n, d_idx, d_scale, d_out, d_skip, (const unsigned char*)nullptr);
    }
    return post_launch(c, "newton_iter");
}

int idahip_newton_iter(idahip_ctx* c, const double* hScale, double* hDelnrm, const int32_t* hIdx, int nsys) {
    DevGuard dev_guard__(c);
    int rc = check_list(c, hIdx, nsys);
    if (rc) return rc;
    if (!hScale || !hDelnrm) return fail(c, -2, "null argument");
    if (nsys == 0) return 0;
    const int n = c->n;
    ArgPack ap;
    if ((rc = ap.begin(c))) return rc;
    const int* d_idx = ap.in(hIdx, nsys);
    const double* d_scale = ap.in(hScale, nsys);
    double* d_out = ap.out<double>(nsys);
    if ((rc = ap.ok())) return rc;
    if ((rc = ap.upload())) return rc;
    {
        KTimer kt(c, IDAHIP_K_NEWTON_ITER, nsys);
        if ((rc = launch_newton_iter(c, d_idx, d_scale, d_out, nullptr, nsys))) return rc;
    }
    if ((rc = ap.fetch())) return rc;
    const double* h = ap.host_of(d_out);
    for (int s = 0; s < nsys; ++s) hDelnrm[s] = sqrt(h[s] / (double)n);
    return 0;
}

int idahip_newton_iter2(idahip_ctx* c, const double* hScale, const double* hTn, const double* hCj, const double* hToldel, const double* hSs,
                        const double* hEpsNewt, double* hDelnrm, int32_t* hConv, const int32_t* hIdx, int nsys) {
    DevGuard dev_guard__(c);
    int rc = check_list(c, hIdx, nsys);
    if (rc) return rc;
    if (!hScale || !hTn || !hCj || !hToldel || !hSs || !hEpsNewt || !hDelnrm || !hConv) return fail(c, -2, "null argument");
    if (c->kind == IDAHIP_HOST_CALLBACK) return fail(c, -2, "idahip_newton_iter2 needs a device residual (not for IDAHIP_HOST_CALLBACK)");
    if (nsys == 0) return 0;
    const int n = c->n;
    ArgPack ap;
    if ((rc = ap.begin(c))) return rc;
    SysArgs a;
    a.idx = ap.in(hIdx, nsys);
    a.tn = ap.in(hTn, nsys);
    a.cj = ap.in(hCj, nsys);
    const double* d_scale = ap.in(hScale, nsys);
    const double* d_toldel = ap.in(hToldel, nsys);
    const double* d_ss = ap.in(hSs, nsys);
    const double* d_eps = ap.in(hEpsNewt, nsys);
    double* d_dn = ap.out<double>(2 * (size_t)nsys);
    int* d_conv = ap.out<int>(nsys);
    double* d_sum = ap.out<double>(nsys);
    if ((rc = ap.ok())) return rc;
    if ((rc = ap.upload())) return rc;
    fill_sys_args(c, a, 0);
    const dim3 tg((nsys + 255) / 256), tb(256);
    {   // iteration m = 0 and its convergence test
        KTimer kt(c, IDAHIP_K_NEWTON_ITER, nsys);
        if ((rc = launch_newton_iter(c, a.idx, d_scale, d_sum, nullptr, nsys))) return rc;
        hipLaunchKernelGGL(ctest_kernel, tg, tb, 0, c->stream, (const double*)d_sum, n, 0, d_toldel, d_ss, d_eps, d_dn, d_conv, nsys);
    }
    a.skip = d_conv;
    {   // NLProblem::sys at the corrected y for the systems that go on (conv == 0)
        KTimer kt(c, IDAHIP_K_SYS, 0);
        if ((rc = launch_sys(c, a, nsys, nullptr, hTn, hIdx))) return rc;
    }
    {   // iteration m = 1 and its convergence test
        KTimer kt(c, IDAHIP_K_NEWTON_ITER, 0);
        if ((rc = launch_newton_iter(c, a.idx, d_scale, d_sum, d_conv, nsys))) return rc;
        hipLaunchKernelGGL(ctest_kernel, tg, tb, 0, c->stream, (const double*)d_sum, n, 1, d_toldel, d_ss, d_eps, d_dn, d_conv, nsys);
        if ((rc = post_launch(c, "ctest"))) return rc;
    }
    if ((rc = ap.fetch())) return rc;
    const double* hd = ap.host_of((const double*)d_dn);
    const int* hc = ap.host_of((const int*)d_conv);
    for (int s = 0; s < nsys; ++s) {
        hDelnrm[2 * s] = hd[2 * s];
        hDelnrm[2 * s + 1] = hd[2 * s + 1];
        hConv[s] = hc[s];
    }
    return 0;
}

int idahip_kind(const idahip_ctx* c) { return c ? (int)c->kind : -1; }

// ------------------------------------------------------------------------------------------------ stepper vector ops
int idahip_init_first(idahip_ctx* c, double* hYpnorm, double* hPhi0Nrm, const int32_t* hIdx, int nsys) {
    DevGuard dev_guard__(c);
    int rc = check_list(c, hIdx, nsys);
    if (rc) return rc;
    if (!hYpnorm || !hPhi0Nrm) return fail(c, -2, "null argument");
    if (nsys == 0) return 0;
    ArgPack ap;
    if ((rc = ap.begin(c))) return rc;
    const int* d_idx = ap.in(hIdx, nsys);
    double* d_out = ap.out<double>(2 * (size_t)nsys);
    if ((rc = ap.ok())) return rc;
    if ((rc = ap.upload())) return rc;
    {
        KTimer kt(c, IDAHIP_K_VECTOR, nsys);
        hipLaunchKernelGGL(init_first_kernel, dim3(nsys), dim3(256), 2 * sizeof(double) * c->n, c->stream, vec_state(c), d_idx, d_out);
        if ((rc = post_launch(c, "init_first"))) return rc;
    }
    if ((rc = ap.fetch())) return rc;
    const double* h = ap.host_of(d_out);
    for (int s = 0; s < nsys; ++s) {
        hYpnorm[s] = sqrt(h[2 * s] / (double)c->n);
        hPhi0Nrm[s] = sqrt(h[2 * s + 1] / (double)c->n);
    }
    return 0;
}

int idahip_scale_phi1(idahip_ctx* c, const double* hFac, const int32_t* hIdx, int nsys) {
    DevGuard dev_guard__(c);
    int rc = check_list(c, hIdx, nsys);
    if (rc) return rc;
    if (!hFac) return fail(c, -2, "null argument");
    if (nsys == 0) return 0;
    ArgPack ap;
    if ((rc = ap.begin(c))) return rc;
    const int* d_idx = ap.in(hIdx, nsys);
    const double* d_fac = ap.in(hFac, nsys);
    if ((rc = ap.upload())) return rc;
    {
        KTimer kt(c, IDAHIP_K_VECTOR, nsys);
        hipLaunchKernelGGL(scale_phi1_kernel, dim3(nsys), dim3(256), 0, c->stream, vec_state(c), d_idx, d_fac);
        if ((rc = post_launch(c, "scale_phi1"))) return rc;
    }
    return ap.finish_async();
}

int idahip_predict(idahip_ctx* c, const int32_t* hKkNs, const double* hBeta, const double* hGamma, const int32_t* hIdx, int nsys) {
    DevGuard dev_guard__(c);
    int rc = check_list(c, hIdx, nsys);
    if (rc) return rc;
    if (!hKkNs || !hBeta || !hGamma) return fail(c, -2, "null argument");
    if (nsys == 0) return 0;
    for (int s = 0; s < nsys; ++s)
        if (hKkNs[2 * s] < 1 || hKkNs[2 * s] >= MXORDP1 || hKkNs[2 * s + 1] < 0) return fail(c, -2, "bad kk/ns at list position %d", s);
    ArgPack ap;
    if ((rc = ap.begin(c))) return rc;
    const int* d_idx = ap.in(hIdx, nsys);
    const int* d_kkns = ap.in(hKkNs, 2 * (size_t)nsys);
    const double* d_beta = ap.in(hBeta, (size_t)MXORDP1 * nsys);
    const double* d_gamma = ap.in(hGamma, (size_t)MXORDP1 * nsys);
    if ((rc = ap.upload())) return rc;
    {
        KTimer kt(c, IDAHIP_K_VECTOR, nsys);
        hipLaunchKernelGGL(predict_kernel, dim3(nsys), dim3(256), 0, c->stream, vec_state(c), d_idx, d_kkns, d_beta, d_gamma);
        if ((rc = post_launch(c, "predict"))) return rc;
    }
    return ap.finish_async();
}

int idahip_post_newton(idahip_ctx* c, const double* hCj, const int32_t* hKk, double* hNorms, const int32_t* hIdx, int nsys) {
    DevGuard dev_guard__(c);
    int rc = check_list(c, hIdx, nsys);
    if (rc) return rc;
    if (!hCj || !hKk || !hNorms) return fail(c, -2, "null argument");
    if (nsys == 0) return 0;
    for (int s = 0; s < nsys; ++s)
        if (hKk[s] < 1 || hKk[s] >= MXORDP1) return fail(c, -2, "bad kk at list position %d", s);
    ArgPack ap;
    if ((rc = ap.begin(c))) return rc;
    const int* d_idx = ap.in(hIdx, nsys);
    const double* d_cj = ap.in(hCj, nsys);
    const int* d_kk = ap.in(hKk, nsys);
    double* d_out = ap.out<double>(4 * (size_t)nsys);
    if ((rc = ap.ok())) return rc;
    if ((rc = ap.upload())) return rc;
    {
        KTimer kt(c, IDAHIP_K_VECTOR, nsys);
        hipLaunchKernelGGL(post_newton_kernel, dim3(nsys), dim3(256), 4 * sizeof(double) * c->n, c->stream, vec_state(c), d_idx, d_cj, d_kk, d_out);
        if ((rc = post_launch(c, "post_newton"))) return rc;
    }
    if ((rc = ap.fetch())) return rc;
    const double* h = ap.host_of(d_out);
    for (size_t e = 0; e < 4 * (size_t)nsys; ++e) hNorms[e] = sqrt(h[e] / (double)c->n);
    return 0;
}

int idahip_restore(idahip_ctx* c, const int32_t* hKkNs, const double* hCvals, const int32_t* hIdx, int nsys) {
    DevGuard dev_guard__(c);
    int rc = check_list(c, hIdx, nsys);
    if (rc) return rc;
    if (!hKkNs || !hCvals) return fail(c, -2, "null argument");
    if (nsys == 0) return 0;
    for (int s = 0; s < nsys; ++s)
        if (hKkNs[2 * s] < 1 || hKkNs[2 * s] >= MXORDP1 || hKkNs[2 * s + 1] < 0) return fail(c, -2, "bad kk/ns at list position %d", s);
    ArgPack ap;
    if ((rc = ap.begin(c))) return rc;
    const int* d_idx = ap.in(hIdx, nsys);
    const int* d_kkns = ap.in(hKkNs, 2 * (size_t)nsys);
    const double* d_cv = ap.in(hCvals, (size_t)MXORDP1 * nsys);
    if ((rc = ap.upload())) return rc;
    {
        KTimer kt(c, IDAHIP_K_VECTOR, nsys);
        hipLaunchKernelGGL(restore_kernel, dim3(nsys), dim3(256), 0, c->stream, vec_state(c), d_idx, d_kkns, d_cv);
        if ((rc = post_launch(c, "restore"))) return rc;
    }
    return ap.finish_async();
}

int idahip_complete_step(idahip_ctx* c, const int32_t* hKused, const double* hCk, int maxord, double* hPhi0Nrm, int32_t* hEwtBad,
                         const int32_t* hIdx, int nsys) {
    DevGuard dev_guard__(c);
    int rc = check_list(c, hIdx, nsys);
    if (rc) return rc;
    if (!hKused || !hCk || !hPhi0Nrm || !hEwtBad) return fail(c, -2, "null argument");
    if (maxord < 1 || maxord > 5) return fail(c, -2, "bad maxord");
    if (nsys == 0) return 0;
    for (int s = 0; s < nsys; ++s)
        if (hKused[s] < 1 || hKused[s] > maxord) return fail(c, -2, "bad kused at list position %d", s);
    ArgPack ap;
    if ((rc = ap.begin(c))) return rc;
    const int* d_idx = ap.in(hIdx, nsys);
    const int* d_ku = ap.in(hKused, nsys);
    const double* d_ck = ap.in(hCk, nsys);
    double* d_out = ap.out<double>(nsys);
    int* d_bad = ap.out<int>(nsys);
    if ((rc = ap.ok())) return rc;
    if ((rc = ap.upload())) return rc;
    {
        KTimer kt(c, IDAHIP_K_VECTOR, nsys);
        hipLaunchKernelGGL(complete_step_kernel, dim3(nsys), dim3(256), sizeof(double) * c->n, c->stream, vec_state(c), d_idx, d_ku, d_ck, maxord,
                           d_out, d_bad);
        if ((rc = post_launch(c, "complete_step"))) return rc;
    }
    if ((rc = ap.fetch())) return rc;
    const double* h = ap.host_of(d_out);
    const int* hb = ap.host_of(d_bad);
    for (int s = 0; s < nsys; ++s) {
        hPhi0Nrm[s] = sqrt(h[s] / (double)c->n);
        hEwtBad[s] = hb[s];
    }
    return 0;
}

int idahip_get_solution(idahip_ctx* c, const int32_t* hKord, const double* hCvals, const double* hDvals, const int32_t* hIdx, int nsys) {
    DevGuard dev_guard__(c);
    int rc = check_list(c, hIdx, nsys);
    if (rc) return rc;
    if (!hKord || !hCvals || !hDvals) return fail(c, -2, "null argument");
    if (nsys == 0) return 0;
    for (int s = 0; s < nsys; ++s)
        if (hKord[s] < 1 || hKord[s] > 5) return fail(c, -2, "bad kord at list position %d", s);
    ArgPack ap;
    if ((rc = ap.begin(c))) return rc;
    const int* d_idx = ap.in(hIdx, nsys);
    const int* d_ko = ap.in(hKord, nsys);
    const double* d_cv = ap.in(hCvals, (size_t)MXORDP1 * nsys);
    const double* d_dv = ap.in(hDvals, (size_t)5 * nsys);
    if ((rc = ap.upload())) return rc;
    {
        KTimer kt(c, IDAHIP_K_VECTOR, nsys);
        hipLaunchKernelGGL(get_solution_kernel, dim3(nsys), dim3(256), 0, c->stream, vec_state(c), d_idx, d_ko, d_cv, d_dv);
        if ((rc = post_launch(c, "get_solution"))) return rc;
    }
    return ap.finish_async();
}

int idahip_get_dky(idahip_ctx* c, const int32_t* hKfirst, const int32_t* hKlast, const double* hCjk, double* hOut, const int32_t* hIdx,
                   int nsys) {
    DevGuard dev_guard__(c);
    int rc = check_list(c, hIdx, nsys);
    if (rc) return rc;
    if (!hKfirst || !hKlast || !hCjk || !hOut) return fail(c, -2, "null argument");
    if (nsys == 0) return 0;
    for (int s = 0; s < nsys; ++s)
        if (hKfirst[s] < 0 || hKfirst[s] > hKlast[s] || hKlast[s] >= MXORDP1) return fail(c, -2, "bad derivative range at list position %d", s);
    const size_t bn = (size_t)c->batch * c->n;
    if (!c->dky && (rc = dalloc(c, &c->dky, bn))) return rc;
    ArgPack ap;
    if ((rc = ap.begin(c))) return rc;
    const int* d_idx = ap.in(hIdx, nsys);
    const int* d_k0 = ap.in(hKfirst, nsys);
    const int* d_k1 = ap.in(hKlast, nsys);
    const double* d_c = ap.in(hCjk, (size_t)MXORDP1 * nsys);
    if ((rc = ap.upload())) return rc;
    {
        KTimer kt(c, IDAHIP_K_VECTOR, nsys);
        hipLaunchKernelGGL(get_dky_kernel, dim3(nsys), dim3(256), 0, c->stream, vec_state(c), d_idx, d_k0, d_k1, d_c, c->dky);
        if ((rc = post_launch(c, "get_dky"))) return rc;
    }
    IDAHIP_HIP(c, hipMemcpyAsync(hOut, c->dky, sizeof(double) * (size_t)nsys * c->n, hipMemcpyDeviceToHost, c->stream));
    IDAHIP_HIP(c, hipStreamSynchronize(c->stream));
    return 0;
}

int idahip_snapshot_initial(idahip_ctx* c) {
    DevGuard dev_guard__(c);
    if (!c) return -1;
    const size_t bn = (size_t)c->batch * c->n;
    int rc = 0;
    if (!c->ic_y) {
        rc |= dalloc(c, &c->ic_y, bn);
        rc |= dalloc(c, &c->ic_yp, bn);
        if (rc) return fail(c, -4, "device allocation of the initial-condition copies failed");
    }
    IDAHIP_HIP(c, hipMemcpyAsync(c->ic_y, c->phi, bn * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    IDAHIP_HIP(c, hipMemcpyAsync(c->ic_yp, c->phi + bn, bn * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    return 0;
}

int idahip_restore_initial(idahip_ctx* c, const int32_t* hIdx, int nsys) {
    DevGuard dev_guard__(c);
    int rc = check_list(c, hIdx, nsys);
    if (rc) return rc;
    if (!c->ic_y) return fail(c, -2, "idahip_restore_initial before idahip_snapshot_initial");
    if (nsys == 0) return 0;
    ArgPack ap;
    if ((rc = ap.begin(c))) return rc;
    const int* d_idx = ap.in(hIdx, nsys);
    if ((rc = ap.upload())) return rc;
    {
        KTimer kt(c, IDAHIP_K_VECTOR, nsys);
        hipLaunchKernelGGL(restore_initial_kernel, dim3(nsys), dim3(256), 0, c->stream, vec_state(c), (const double*)c->ic_y,
                           (const double*)c->ic_yp, d_idx);
        if ((rc = post_launch(c, "restore_initial"))) return rc;
    }
    return ap.finish_async();
}

// ---------------------------------------------------------------------------------------------- device-resident stepper (n <= 8)
__global__ void pow_batch_kernel(const double* __restrict__ x, const double* __restrict__ y, double* __restrict__ out, size_t count) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = glibc_pow::pow(x[i], y[i]);
}

int idahip_pow_batch(idahip_ctx* c, const double* hX, const double* hY, double* hOut, size_t count) {
    DevGuard dev_guard__(c);
    if (!c) return -1;
    if (!hX || !hY || !hOut) return fail(c, -2, "null argument");
    if (count == 0) return 0;
    double *dx = nullptr, *dy = nullptr, *dout = nullptr;
    int rc = dalloc(c, &dx, count);
    if (!rc) rc = dalloc(c, &dy, count);
    if (!rc) rc = dalloc(c, &dout, count);
    if (!rc) {
        hipError_t e = hipMemcpyAsync(dx, hX, count * sizeof(double), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(dy, hY, count * sizeof(double), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(pow_batch_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, c->stream, (const double*)dx,
                               (const double*)dy, dout, count);
            e = hipMemcpyAsync(hOut, dout, count * sizeof(double), hipMemcpyDeviceToHost, c->stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) rc = fail(c, -100, "idahip_pow_batch: %s", hipGetErrorString(e));
    }
    if (dx) (void)hipFree(dx);
    if (dy) (void)hipFree(dy);
    if (dout) (void)hipFree(dout);
    return rc;
}

static int stepper_buffers(idahip_ctx* c, const idahip_tiny_call* call, bool outputs);

// the schedule and the limits of a stepper call as the device-resident steppers take them (touts and start rounds already on
// the device)
static FlowArgs flow_args(idahip_ctx* c, const idahip_tiny_call* call) {
    FlowArgs f;
    f.touts = c->tiny_touts; f.ntout = call->ntout; f.recycle = call->recycle; f.resume = call->resume;
    f.mxstep = call->mxstep; f.maxord = call->maxord; f.maxnef = call->maxnef; f.maxncf = call->maxncf;
    f.epcon = call->epcon; f.hmax_inv = call->hmax_inv; f.t0 = call->t0;
    f.start_round = call->start_round ? (const long long*)c->tiny_start : nullptr;
    f.acc = (unsigned long long*)c->tiny_acc;
    f.batch = c->batch;
    f.nrt = call->nroots;
    for (int i = 0; i < call->nroots && i < IDAHIP_MAX_ROOTS; ++i) {
        f.rt_comp[i] = call->root_comps[i];
        f.rt_thr[i] = call->root_thresholds[i];
    }
    return f;
}

int idahip_tiny_solve(idahip_ctx* c, void* hSys, size_t sys_bytes, const idahip_tiny_call* call, int64_t* hRoundsDone, uint64_t* hAcc,
                      double* hYout, double* hYPout) {
    DevGuard dev_guard__(c);
    if (!c) return -1;
    if (!hSys || !call || !hRoundsDone || !hAcc || !call->touts || call->ntout < 1) return fail(c, -2, "null argument");
    if (sys_bytes != sizeof(idactl::SysCore)) return fail(c, -2, "controller state of %zu bytes, this library expects %zu", sys_bytes, sizeof(idactl::SysCore));
    if (c->n != 3 || (c->kind != IDAHIP_ROBERTS && c->kind != IDAHIP_LORENZ63))
        return fail(c, -2, "the device-resident stepper takes the Roberts and Lorenz63 problems (n = 3)");
    if (call->recycle && (!c->ic_y || !c->ic_yp)) return fail(c, -2, "recycle needs idahip_snapshot_initial");
    if (call->recycle && call->max_rounds < 1) return fail(c, -2, "recycle needs a round limit");
    const int batch = c->batch, n = c->n;
    int rc = stepper_buffers(c, call, hYout || hYPout);
    if (rc) return rc;
    const size_t ysz = (size_t)call->ntout * batch * n;
    IDAHIP_HIP(c, hipMemcpyAsync(c->tiny_sys, hSys, (size_t)batch * sizeof(idactl::SysCore), hipMemcpyHostToDevice, c->stream));
    IDAHIP_HIP(c, hipMemcpyAsync(c->tiny_touts, call->touts, sizeof(double) * call->ntout, hipMemcpyHostToDevice, c->stream));
    if (call->start_round)
        IDAHIP_HIP(c, hipMemcpyAsync(c->tiny_start, call->start_round, sizeof(int64_t) * batch, hipMemcpyHostToDevice, c->stream));
    IDAHIP_HIP(c, hipMemsetAsync(c->tiny_acc, 0, 2 * sizeof(uint64_t), c->stream));
    TinyIdaArgs a;
    a.sys = (idactl::SysCore*)c->tiny_sys;
    a.v = vec_state(c);
    a.savres = c->savres; a.lu = c->lu; a.piv = (long long*)c->piv; a.params = c->params; a.nparam = c->nparam;
    a.ic_y = c->ic_y; a.ic_yp = c->ic_yp;
    a.f = flow_args(c, call);
    a.roots = (idahip_root_state*)c->tiny_roots;
    a.max_rounds = call->max_rounds;
    a.round_base = call->round_base;
    a.yout = hYout ? c->tiny_yout : nullptr;
    a.ypout = hYPout ? c->tiny_ypout : nullptr;
    a.rounds_done = (long long*)c->tiny_rounds;
    {
        KTimer kt(c, IDAHIP_K_VECTOR, batch);
        // Systems per wavefront. A lane walks one system's dependent fp64 chains, and the lanes of a wavefront, each in another phase
        // of its step (one, two or four Newton iterations, a failed error test, an order change, an output), execute each
        // other's branches. A batch that does not fill the device therefore runs on more, narrower wavefronts: the fewest
        // systems per wavefront (64, 32, ... 1) that keep the wavefront count at or below one per CU. Config 2 (1024 systems,
        // same box): 64 per wavefront 57.3 M Newton iterations/s in the stream (67.1 M whole pass), 16: 63.1 (70.5), 4: 70.9 (75.8),
        // 1: 65.6 (67.5). IDAHIP_TINY_SPW overrides (measurements). The per-system program is the same: results do not depend on it.
        int spw = 64;
        {
            const int simds = c->simds > 0 ? c->simds : 1024;
            while (spw > 1 && (long)(batch + spw / 2 - 1) / (spw / 2) <= simds / 4) spw >>= 1;
            if (const char* e = std::getenv("IDAHIP_TINY_SPW")) {
                const int v = (int)std::strtol(e, nullptr, 10);
                if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16 || v == 32 || v == 64) spw = v;
            }
        }
        // LDS: the controller records of a workgroup, and the systems' vectors too when the device grants that much
        const size_t lds_state = (size_t)spw * sizeof(idactl::SysCore), lds_all = lds_state + (size_t)spw * sizeof(double) * tiny_lds_doubles(n);
        // four instantiations: problem x (root finding compiled in | out -- the bracketing code costs the plain stepper registers)
        const bool roots = call->nroots > 0;
        auto kern = c->kind == IDAHIP_ROBERTS ? (roots ? tiny_ida_kernel<IDAHIP_ROBERTS, true> : tiny_ida_kernel<IDAHIP_ROBERTS, false>)
                                              : (roots ? tiny_ida_kernel<IDAHIP_LORENZ63, true> : tiny_ida_kernel<IDAHIP_LORENZ63, false>);
        const int lds_vec = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(64 * (sizeof(idactl::SysCore) + sizeof(double) * tiny_lds_doubles(n)))) == hipSuccess ? 1 : 0;
        if (!lds_vec) (void)hipGetLastError();
        const size_t shm = lds_vec ? lds_all : lds_state;
        hipLaunchKernelGGL(kern, dim3((batch + spw - 1) / spw), dim3(spw), shm, c->stream, a, lds_vec);
        if ((rc = post_launch(c, "tiny_ida"))) return rc;
    }
    IDAHIP_HIP(c, hipMemcpyAsync(hSys, c->tiny_sys, (size_t)batch * sizeof(idactl::SysCore), hipMemcpyDeviceToHost, c->stream));
    if (call->nroots > 0)
        IDAHIP_HIP(c, hipMemcpyAsync(call->root_states, c->tiny_roots, (size_t)batch * sizeof(idahip_root_state), hipMemcpyDeviceToHost, c->stream));
    IDAHIP_HIP(c, hipMemcpyAsync(hRoundsDone, c->tiny_rounds, sizeof(int64_t) * batch, hipMemcpyDeviceToHost, c->stream));
    IDAHIP_HIP(c, hipMemcpyAsync(hAcc, c->tiny_acc, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    if (hYout) IDAHIP_HIP(c, hipMemcpyAsync(hYout, c->tiny_yout, sizeof(double) * ysz, hipMemcpyDeviceToHost, c->stream));
    if (hYPout) IDAHIP_HIP(c, hipMemcpyAsync(hYPout, c->tiny_ypout, sizeof(double) * ysz, hipMemcpyDeviceToHost, c->stream));
    IDAHIP_HIP(c, hipStreamSynchronize(c->stream));
    return 0;
}

// buffers every device-resident stepper call needs: controller states, the schedule, the stagger, counters, output slots
static int stepper_buffers(idahip_ctx* c, const idahip_tiny_call* call, bool outputs) {
    const int batch = c->batch, n = c->n;
    int rc = 0;
    // (each buffer is guarded by itself: a failed allocation must not leave a later call with one non-null pointer that
    // stands for the whole group)
    if (!c->tiny_sys) IDAHIP_HIP(c, hipMalloc(&c->tiny_sys, (size_t)batch * sizeof(idactl::SysCore)));
    if (!c->tiny_start) rc |= dalloc(c, &c->tiny_start, (size_t)batch);
    if (!c->tiny_rounds) rc |= dalloc(c, &c->tiny_rounds, (size_t)batch);
    if (!c->tiny_acc) rc |= dalloc(c, &c->tiny_acc, (size_t)2);
    if (rc) return rc;
    if (call->nroots < 0 || call->nroots > IDAHIP_MAX_ROOTS) return fail(c, -2, "the device steppers take at most %d root functions", IDAHIP_MAX_ROOTS);
    if (call->nroots > 0) {
        if (!call->root_comps || !call->root_thresholds || !call->root_states) return fail(c, -2, "root functions without their arrays");
        for (int i = 0; i < call->nroots; ++i)
            if (call->root_comps[i] < 0 || call->root_comps[i] >= n) return fail(c, -2, "root function %d watches component %d of %d", i, call->root_comps[i], n);
        if (call->recycle) return fail(c, -2, "root finding and idaens_stream's restarts do not combine");
        if (!c->tiny_roots) IDAHIP_HIP(c, hipMalloc(&c->tiny_roots, (size_t)batch * sizeof(idahip_root_state)));
        IDAHIP_HIP(c, hipMemcpyAsync(c->tiny_roots, call->root_states, (size_t)batch * sizeof(idahip_root_state), hipMemcpyHostToDevice, c->stream));
    }
    if (call->ntout > c->tiny_ntout_cap) {
        if (c->tiny_touts) (void)hipFree(c->tiny_touts);
        c->tiny_touts = nullptr;
        if ((rc = dalloc(c, &c->tiny_touts, (size_t)call->ntout))) return rc;
        c->tiny_ntout_cap = call->ntout;
    }
    if (outputs && call->ntout > c->tiny_yout_cap) {
        const size_t ysz = (size_t)call->ntout * batch * n;
        if (c->tiny_yout) (void)hipFree(c->tiny_yout);
        if (c->tiny_ypout) (void)hipFree(c->tiny_ypout);
        c->tiny_yout = c->tiny_ypout = nullptr;
        if ((rc = dalloc(c, &c->tiny_yout, ysz))) return rc;
        if ((rc = dalloc(c, &c->tiny_ypout, ysz))) return rc;
        c->tiny_yout_cap = call->ntout;
    }
    return 0;
}

int idahip_round_solve(idahip_ctx* c, void* hSys, size_t sys_bytes, const idahip_tiny_call* call, int64_t* hRoundsDone, uint64_t* hAcc,
                       double* hYout, double* hYPout, int64_t* rounds_run) {
    DevGuard dev_guard__(c);
    if (!c) return -1;
    if (!hSys || !call || !hRoundsDone || !hAcc || !rounds_run || !call->touts || call->ntout < 1) return fail(c, -2, "null argument");
    if (sys_bytes != sizeof(idactl::SysCore)) return fail(c, -2, "controller state of %zu bytes, this library expects %zu", sys_bytes, sizeof(idactl::SysCore));
    const bool lin = c->kind == IDAHIP_LINEAR_DENSE && c->n <= LU_BIG_MAX_N;
    const bool heat = c->kind == IDAHIP_HEAT1D && c->n <= LU_BIG_MAX_N;
    if (c->n <= TINY_N || !(lin || heat) || c->lu_variant < 4)
        return fail(c, -2, "the device-resident lock-step stepper takes linear dense and heat problems with %d < n <= %d (LU variant 4)", TINY_N, LU_BIG_MAX_N);
    if (call->recycle && (!c->ic_y || !c->ic_yp)) return fail(c, -2, "recycle needs idahip_snapshot_initial");
    if (call->recycle && call->max_rounds < 1) return fail(c, -2, "recycle needs a round limit");
    const int batch = c->batch, n = c->n;
    const long nn = (long)n * n;
    int rc = stepper_buffers(c, call, hYout || hYPout);
    if (rc) return rc;
    if (!c->rnd_i) rc |= dalloc(c, &c->rnd_i, (size_t)8 * batch + 8 + 2 * IDAHIP_K_COUNT);
    if (!c->rnd_d) rc |= dalloc(c, &c->rnd_d, (size_t)4 * batch);
    if (rc) return rc;
    if (!c->rnd_host) IDAHIP_HIP(c, hipHostMalloc((void**)&c->rnd_host, 4 * sizeof(int32_t)));
    const size_t ysz = (size_t)call->ntout * batch * n;
    IDAHIP_HIP(c, hipMemcpyAsync(c->tiny_sys, hSys, (size_t)batch * sizeof(idactl::SysCore), hipMemcpyHostToDevice, c->stream));
    IDAHIP_HIP(c, hipMemcpyAsync(c->tiny_touts, call->touts, sizeof(double) * call->ntout, hipMemcpyHostToDevice, c->stream));
    if (call->start_round)
        IDAHIP_HIP(c, hipMemcpyAsync(c->tiny_start, call->start_round, sizeof(int64_t) * batch, hipMemcpyHostToDevice, c->stream));
    IDAHIP_HIP(c, hipMemsetAsync(c->tiny_acc, 0, 2 * sizeof(uint64_t), c->stream));
    RoundArgs a;
    a.f = flow_args(c, call);
    a.sys = (idactl::SysCore*)c->tiny_sys;
    a.v = vec_state(c);
    a.ic_y = c->ic_y; a.ic_yp = c->ic_yp;
    a.yout = hYout ? c->tiny_yout : nullptr;
    a.ypout = hYPout ? c->tiny_ypout : nullptr;
    a.round_base = call->round_base;
    a.fused_jac = c->kind == IDAHIP_LINEAR_DENSE ? 1 : 0;
    a.lu_period = c->lu_period;
    a.roots = (idahip_root_state*)c->tiny_roots;
    int* ib = c->rnd_i;
    a.stepping = ib; a.in_newton = ib + batch; a.skipP = ib + 2 * batch; a.skipL = ib + 3 * batch; a.skipI = ib + 4 * batch;
    a.skipS = ib + 5 * batch; a.ident = ib + 6 * batch; a.lu_list = ib + 7 * batch; a.lu_cnt = ib + 8 * batch; a.summary = ib + 8 * batch + 1; a.lu_wait = ib + 8 * batch + 3;
    a.stats = (unsigned long long*)(ib + 8 * batch + 4);  // (8 * batch + 4 ints: 8-byte aligned for even batch; checked below)
    if (((uintptr_t)a.stats & 7) != 0) a.stats = (unsigned long long*)(ib + 8 * batch + 5);
    IDAHIP_HIP(c, hipMemsetAsync(a.stats, 0, IDAHIP_K_COUNT * sizeof(unsigned long long), c->stream));
    a.lu_info = c->lu_info;
    a.tn = c->rnd_d; a.cj = c->rnd_d + batch; a.scale = c->rnd_d + 2 * batch; a.nrm_out = c->rnd_d + 3 * batch;
    a.rounds_done = (long long*)c->tiny_rounds;
    hipLaunchKernelGGL(round_init_kernel, dim3((batch + 255) / 256), dim3(256), 0, c->stream, a);
    IDAHIP_HIP(c, hipMemsetAsync(a.summary, 0, 2 * sizeof(int), c->stream));
    const size_t shm_wg = sizeof(double) * (4 * (size_t)n + 8);
    const size_t shm_it = 2 * sizeof(double) * n;
    SysArgs sa;
    fill_sys_args(c, sa, 1);
    sa.idx = a.ident; sa.tn = a.tn; sa.cj = a.cj;
    int64_t r = 0;
    for (;; ++r) {
        if (call->max_rounds > 0 && r >= call->max_rounds) break;
        a.round = r;
        a.first_round = r == 0;
        {
            KTimer kt(c, IDAHIP_K_VECTOR, 0);
            if (call->nroots > 0) hipLaunchKernelGGL(round_begin_kernel<true>, dim3(batch), dim3(WG_NT), shm_wg, c->stream, a);
            else hipLaunchKernelGGL(round_begin_kernel<false>, dim3(batch), dim3(WG_NT), shm_wg, c->stream, a);
            hipLaunchKernelGGL(round_lists_kernel, dim3(1), dim3(1024), 0, c->stream, a);
        }
        // n > 1024 (config 4): a factorisation is ~640 launches, most of them one workgroup per matrix, and a third of a small
        // batch needs one in any round -- launches sized for the whole batch cost more than they do at n = 512 (11.4 against
        // 12.3 k iters/s in round 4). The list's length comes back to the host (4 bytes, behind the residual kernels enqueued
        // below: the copy's round trip is hidden) and the LU's launches are sized by it, as the host stepper's are.
        const bool lu_cnt_on_host = n > LU_MAX_N;
        if (lu_cnt_on_host) {
            IDAHIP_HIP(c, hipMemcpyAsync(c->rnd_host + 2, a.lu_cnt, sizeof(int), hipMemcpyDeviceToHost, c->stream));
            IDAHIP_HIP(c, hipEventRecord(c->ev_cnt, c->stream));
        }
        {   // sys(y0), y <- y0 = 0 (newton.rs:73): without and with the Jacobian (J = B + cj A falls out of the same sweep)
            KTimer kt(c, IDAHIP_K_SYS, 0);
            sa.reset_ee = 1; sa.skip = a.skipP;
            if ((rc = launch_sys(c, sa, batch, nullptr))) return rc;
        }
        if (c->kind == IDAHIP_LINEAR_DENSE) {
            KTimer kt(c, IDAHIP_K_SYS_JAC, 0);
            sa.skip = a.skipL;
            if ((rc = launch_sys(c, sa, batch, c->jw))) return rc;
        } else {  // no fused residual + Jacobian kernel for this problem: the residual, then the Jacobian of the same systems
            {
                KTimer kt(c, IDAHIP_K_SYS, 0);
                sa.skip = a.skipL;
                if ((rc = launch_sys(c, sa, batch, nullptr))) return rc;
            }
            KTimer kt(c, IDAHIP_K_JAC, 0);
            if ((rc = launch_jac(c, c->jw, a.ident, a.cj, batch, nullptr, nullptr, nullptr, a.skipL))) return rc;
        }
        {
            int nlu = batch;
            const int* d_cnt = a.lu_cnt;
            if (lu_cnt_on_host) {
                IDAHIP_HIP(c, hipEventSynchronize(c->ev_cnt));
                nlu = c->rnd_host[2];
                d_cnt = nullptr;
                if (nlu < 0 || nlu > batch) return fail(c, -4, "list length %d outside 0..%d", nlu, batch);
            }
            KTimer kt(c, IDAHIP_K_LU, 0);
            if ((rc = lu_factor_batched(c, c->jw, nn, c->lu, nn, (long long*)c->piv, n, c->perm, a.lu_list, nlu, d_cnt))) return rc;
            if ((rc = post_launch(c, "lu"))) return rc;
        }
        hipLaunchKernelGGL(round_newton_ctl_kernel, dim3((batch + 255) / 256), dim3(256), 0, c->stream, a, 0);
        for (int m = 1; m <= idactl::MAXNLSIT; ++m) {
            {
                KTimer kt(c, IDAHIP_K_NEWTON_ITER, 0);
                if ((rc = launch_newton_iter(c, a.ident, a.scale, a.nrm_out, a.skipI, batch))) return rc;
            }
            hipLaunchKernelGGL(round_newton_ctl_kernel, dim3((batch + 255) / 256), dim3(256), 0, c->stream, a, m);
            if (m < idactl::MAXNLSIT) {
                KTimer kt(c, IDAHIP_K_SYS, 0);
                sa.reset_ee = 0; sa.skip = a.skipS;
                if ((rc = launch_sys(c, sa, batch, nullptr))) return rc;
            }
        }
        IDAHIP_HIP(c, hipMemsetAsync(a.summary, 0, sizeof(int), c->stream));
        {
            KTimer kt(c, IDAHIP_K_VECTOR, 0);
            if (call->nroots > 0) hipLaunchKernelGGL(round_end_kernel<true>, dim3(batch), dim3(WG_NT), shm_wg, c->stream, a);
            else hipLaunchKernelGGL(round_end_kernel<false>, dim3(batch), dim3(WG_NT), shm_wg, c->stream, a);
            if ((rc = post_launch(c, "round_end"))) return rc;
        }
        if (!call->recycle) {  // one synchronisation per round: does any system still step?
            IDAHIP_HIP(c, hipMemcpyAsync(c->rnd_host, a.summary, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
            IDAHIP_HIP(c, hipStreamSynchronize(c->stream));
            if (c->rnd_host[0] == 0) {
                ++r;
                break;
            }
        }
    }
    (void)shm_it;
    *rounds_run = r;
    IDAHIP_HIP(c, hipMemcpyAsync(hSys, c->tiny_sys, (size_t)batch * sizeof(idactl::SysCore), hipMemcpyDeviceToHost, c->stream));
    if (call->nroots > 0)
        IDAHIP_HIP(c, hipMemcpyAsync(call->root_states, c->tiny_roots, (size_t)batch * sizeof(idahip_root_state), hipMemcpyDeviceToHost, c->stream));
    IDAHIP_HIP(c, hipMemcpyAsync(hRoundsDone, c->tiny_rounds, sizeof(int64_t) * batch, hipMemcpyDeviceToHost, c->stream));
    IDAHIP_HIP(c, hipMemcpyAsync(hAcc, c->tiny_acc, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    if (hYout) IDAHIP_HIP(c, hipMemcpyAsync(hYout, c->tiny_yout, sizeof(double) * ysz, hipMemcpyDeviceToHost, c->stream));
    if (hYPout) IDAHIP_HIP(c, hipMemcpyAsync(hYPout, c->tiny_ypout, sizeof(double) * ysz, hipMemcpyDeviceToHost, c->stream));
    unsigned long long hstats[IDAHIP_K_COUNT];
    IDAHIP_HIP(c, hipMemcpyAsync(hstats, a.stats, sizeof hstats, hipMemcpyDeviceToHost, c->stream));
    IDAHIP_HIP(c, hipStreamSynchronize(c->stream));
    for (int k = 0; k < IDAHIP_K_COUNT; ++k) c->k_systems[k] += (int64_t)hstats[k];  // systems actually served, per kernel class
    c->k_systems[IDAHIP_K_VECTOR] += 2 * r * (int64_t)batch;
    return 0;
}

int idahip_lu_variant(const idahip_ctx* c) { return c ? c->lu_variant : -1; }
int idahip_timing_build(void) { return tb::TIMING_BUILD ? 1 : 0; }
int idahip_set_lu_superpanel(idahip_ctx* c, int on) {
    if (!c) return -1;
    c->lu_superpanel = on != 0 ? 1 : 0;
    return 0;
}
int idahip_lu_superpanel(const idahip_ctx* c) { return c ? c->lu_superpanel : -1; }
int idahip_set_lu_period(idahip_ctx* c, int rounds) {
    if (!c || rounds < 1 || rounds > 64) return -1;
    c->lu_period = rounds;
    return 0;
}
int idahip_lu_period(const idahip_ctx* c) { return c ? c->lu_period : -1; }

// LSolver::get_type / num_iters / res_norm of the dense direct solver (crates/linear/src/dense.rs:30-36, traits.rs:82-90)
int idahip_ls_type(const idahip_ctx* c) { return c ? IDAHIP_LS_DIRECT : -1; }
int idahip_ls_num_iters(const idahip_ctx* c) { return c ? 0 : -1; }
double idahip_ls_res_norm(const idahip_ctx* c) { (void)c; return 0.0; }

#ifdef IDAHIP_STAMPS /* (only accepted together with -DIDAHIP_TIMING_BUILD: exp_switches.hpp) */
/* timing builds: a device buffer for in-kernel time stamps (8 per workgroup of the instrumented launch) */
void* idahip_debug_stamps(idahip_ctx* c, size_t words) {
    if (!c) return nullptr;
    if (!c->dbg_stamps && hipMalloc((void**)&c->dbg_stamps, words * 8) != hipSuccess) return nullptr;
    (void)hipMemset(c->dbg_stamps, 0, words * 8);
    return c->dbg_stamps;
}
#endif

int idahip_set_lu_variant(idahip_ctx* c, int variant) {
    DevGuard dev_guard__(c);
    if (!c || variant < 3 || variant > 4) return -1;  // 3: panel + narrow update kernels, 4: wave-per-matrix panel (default)
    c->lu_variant = variant;
    return 0;
}

// ------------------------------------------------------------------------------------------------ measurement
int idahip_timing_enable(idahip_ctx* c, int on) {
    DevGuard dev_guard__(c);
    if (!c) return -1;
    c->timing = on < 0 ? 0 : (on > 2 ? 2 : on);
    return 0;
}
int idahip_timing_get(idahip_ctx* c, idahip_kclass k, double* ms, int64_t* launches, int64_t* systems) {
    DevGuard dev_guard__(c);
    if (!c || k < 0 || k >= IDAHIP_K_COUNT) return -1;
    if (ms) *ms = c->k_ms[k];
    if (launches) *launches = c->k_launches[k];
    if (systems) *systems = c->k_systems[k];
    return 0;
}
int idahip_timing_reset(idahip_ctx* c) {
    DevGuard dev_guard__(c);
    if (!c) return -1;
    for (int k = 0; k < IDAHIP_K_COUNT; ++k) {
        c->k_ms[k] = 0.0;
        c->k_launches[k] = 0;
        c->k_systems[k] = 0;
    }
    return 0;
}

}  // extern "C"
