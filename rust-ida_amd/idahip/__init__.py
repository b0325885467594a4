"""ctypes bindings of libidahip.so (include/ida_hip.h) and libidaens.so (include/ida_ensemble.h).

The HIP path is the only path: importing this package without the built libraries raises (there is no CPU
fallback and the CPU oracle is never imported from here). Build with `python __graft_entry__.py` or
`make -C rust-ida_amd/csrc && make -C rust-ida_amd/host`.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG = os.path.dirname(_HERE)
LIB_HIP = os.environ.get("IDAHIP_LIB_HIP", os.path.join(_PKG, "csrc", "libidahip.so"))  # the override is for A/B builds (tools/)
LIB_ENS = os.path.join(_PKG, "host", "libidaens.so")

dp = C.POINTER(C.c_double)
i32p = C.POINTER(C.c_int32)
i64p = C.POINTER(C.c_int64)

ROBERTS, LORENZ63, LINEAR_DENSE, HEAT1D, HOST_CALLBACK = 0, 1, 2, 3, 4
KIND = {"roberts": ROBERTS, "lorenz63": LORENZ63, "linear_dense": LINEAR_DENSE, "heat1d": HEAT1D, "host_callback": HOST_CALLBACK}
RES_FN = C.CFUNCTYPE(C.c_int, C.c_int, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p)
JAC_FN = C.CFUNCTYPE(C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                     C.POINTER(C.c_double), C.c_void_p)
ROOT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int32, C.POINTER(C.c_double))
F_YY, F_YP, F_YYPREDICT, F_YPPREDICT, F_EWT, F_EE, F_DELTA, F_SAVRES, F_PHI0 = range(9)
K_NEWTON_ITER, K_SYS, K_JAC, K_LU, K_VECTOR, K_SOLVE, K_SYS_JAC = range(7)
K_NAMES = ["newton_iter", "sys", "jac", "lu", "vector", "solve", "sys_jac", "lu_panel", "lu_trail", "lu_finalize"]

# symbols declared in include/ida_hip.h / include/ida_ensemble.h (checked by tests/test_abi_symbols.py)
HIP_SYMBOLS = [
    "idahip_create", "idahip_destroy", "idahip_last_error", "idahip_sync", "idahip_n", "idahip_batch", "idahip_kind", "idahip_set_tolerances",
    "idahip_set_problem_params", "idahip_set_linear_dense", "idahip_set_host_problem", "idahip_upload", "idahip_download", "idahip_download_lu",
    "idahip_dev_alloc", "idahip_dev_free", "idahip_memcpy_h2d", "idahip_memcpy_d2h", "idahip_ls_setup", "idahip_ls_solve",
    "idahip_wrms", "idahip_nls_sys", "idahip_nls_lsetup", "idahip_nls_sys_setup", "idahip_newton_iter", "idahip_newton_iter2", "idahip_init_first", "idahip_scale_phi1",
    "idahip_predict", "idahip_post_newton", "idahip_restore", "idahip_complete_step", "idahip_get_solution", "idahip_get_dky",
    "idahip_timing_enable", "idahip_timing_get", "idahip_timing_reset", "idahip_set_lu_variant", "idahip_snapshot_initial",
    "idahip_tiny_solve", "idahip_pow_batch", "idahip_round_solve", "idahip_lu_variant",
    "idahip_restore_initial", "idahip_ls_type", "idahip_ls_num_iters", "idahip_ls_res_norm", "idahip_timing_build", "idahip_concurrent_streams", "idahip_release_streams", "idahip_stream_pair_share", "idahip_set_lu_superpanel", "idahip_lu_superpanel", "idahip_set_lu_period", "idahip_lu_period",
]
ENS_SYMBOLS = [
    "idaens_create", "idaens_destroy", "idaens_last_error", "idaens_set_max_num_steps", "idaens_set_max_ord", "idaens_set_fused_newton", "idaens_set_device_controller", "idaens_device_controller_active", "idaens_set_roots", "idaens_set_root_fn",
    "idaens_get_roots", "idaens_solve", "idaens_solve_schedule", "idaens_stream", "idaens_stream_group", "idaens_solve_schedule_group",
    "idaens_get_counter", "idaens_get_real", "idaens_get_yy", "idaens_get_yp", "idaens_get_dky", "idaens_total_newton_iters",
    "idaens_total_rounds", "idaens_trace_system", "idaens_trace_len", "idaens_trace_get",
]

_libs = None


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def _i32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32))


def _p(a, t=dp):
    return None if a is None else a.ctypes.data_as(t)


def load():
    """Load both shared libraries; raises OSError if the HIP extension has not been built."""
    global _libs
    if _libs is not None:
        return _libs
    for p in (LIB_HIP, LIB_ENS):
        if not os.path.exists(p):
            raise OSError("%s is missing: the HIP extension is not built (run `python __graft_entry__.py`); "
                          "there is no CPU fallback" % p)
    H = C.CDLL(LIB_HIP, mode=C.RTLD_GLOBAL)
    H.idahip_timing_build.argtypes = []
    if H.idahip_timing_build() != 0 and os.environ.get("IDAHIP_ALLOW_TIMING_BUILD") != "1":
        raise OSError("%s is a timing build (-DIDAHIP_TIMING_BUILD, csrc/exp_switches.hpp): its results are garbage by design; "
                      "only the measurement tools load one, with IDAHIP_ALLOW_TIMING_BUILD=1" % LIB_HIP)
    E = C.CDLL(LIB_ENS)
    vp, ci, cd = C.c_void_p, C.c_int, C.c_double
    H.idahip_create.argtypes = [C.POINTER(vp), ci, ci, ci, ci, vp]
    H.idahip_destroy.argtypes = [vp]
    H.idahip_concurrent_streams.argtypes = [ci, ci, C.POINTER(vp), C.POINTER(ci)]
    H.idahip_release_streams.argtypes = [ci, ci, C.POINTER(vp)]
    H.idahip_stream_pair_share.argtypes = [ci, vp, vp, C.POINTER(cd)]
    H.idahip_set_lu_superpanel.argtypes = [vp, ci]
    H.idahip_lu_superpanel.argtypes = [vp]
    H.idahip_set_lu_period.argtypes = [vp, ci]
    H.idahip_lu_period.argtypes = [vp]
    H.idahip_last_error.argtypes = [vp]
    H.idahip_last_error.restype = C.c_char_p
    H.idahip_sync.argtypes = [vp]
    H.idahip_n.argtypes = [vp]
    H.idahip_batch.argtypes = [vp]
    H.idahip_set_tolerances.argtypes = [vp, cd, dp, ci]
    H.idahip_set_problem_params.argtypes = [vp, ci, ci, dp, ci]
    H.idahip_set_linear_dense.argtypes = [vp, ci, ci, dp, dp, dp]
    H.idahip_upload.argtypes = [vp, ci, ci, ci, dp]
    H.idahip_download.argtypes = [vp, ci, ci, ci, dp]
    H.idahip_download_lu.argtypes = [vp, ci, dp, i64p]
    H.idahip_dev_alloc.argtypes = [vp, C.c_size_t]
    H.idahip_dev_alloc.restype = vp
    H.idahip_dev_free.argtypes = [vp, vp]
    H.idahip_memcpy_h2d.argtypes = [vp, vp, vp, C.c_size_t]
    H.idahip_memcpy_d2h.argtypes = [vp, vp, vp, C.c_size_t]
    H.idahip_ls_setup.argtypes = [vp, vp, vp, i32p, i32p, ci]
    H.idahip_ls_solve.argtypes = [vp, vp, vp, vp, vp, cd, i32p, ci]
    H.idahip_wrms.argtypes = [vp, vp, vp, dp, i32p, ci]
    H.idahip_ls_type.argtypes = [vp]
    H.idahip_ls_num_iters.argtypes = [vp]
    H.idahip_ls_res_norm.argtypes = [vp]
    H.idahip_ls_res_norm.restype = cd
    H.idahip_nls_sys.argtypes = [vp, dp, dp, ci, i32p, ci]
    H.idahip_nls_lsetup.argtypes = [vp, dp, dp, i32p, i32p, ci]
    H.idahip_nls_sys_setup.argtypes = [vp, dp, dp, ci, i32p, i32p, ci]
    H.idahip_newton_iter.argtypes = [vp, dp, dp, i32p, ci]
    H.idahip_init_first.argtypes = [vp, dp, dp, i32p, ci]
    H.idahip_scale_phi1.argtypes = [vp, dp, i32p, ci]
    H.idahip_predict.argtypes = [vp, i32p, dp, dp, i32p, ci]
    H.idahip_post_newton.argtypes = [vp, dp, i32p, dp, i32p, ci]
    H.idahip_restore.argtypes = [vp, i32p, dp, i32p, ci]
    H.idahip_complete_step.argtypes = [vp, i32p, dp, ci, dp, i32p, i32p, ci]
    H.idahip_get_solution.argtypes = [vp, i32p, dp, dp, i32p, ci]
    H.idahip_get_dky.argtypes = [vp, i32p, i32p, dp, dp, i32p, ci]
    H.idahip_set_lu_variant.argtypes = [vp, ci]
    H.idahip_pow_batch.argtypes = [vp, dp, dp, dp, C.c_size_t]
    H.idahip_newton_iter2.argtypes = [vp, dp, dp, dp, dp, dp, dp, dp, i32p, i32p, ci]
    H.idahip_set_host_problem.argtypes = [vp, RES_FN, JAC_FN, vp]
    H.idahip_timing_enable.argtypes = [vp, ci]
    H.idahip_timing_get.argtypes = [vp, ci, dp, i64p, i64p]
    H.idahip_timing_reset.argtypes = [vp]
    E.idaens_create.argtypes = [C.POINTER(vp), vp, dp, dp]
    E.idaens_destroy.argtypes = [vp]
    E.idaens_last_error.argtypes = [vp]
    E.idaens_last_error.restype = C.c_char_p
    E.idaens_set_max_num_steps.argtypes = [vp, C.c_long]
    E.idaens_set_max_ord.argtypes = [vp, ci]
    E.idaens_set_fused_newton.argtypes = [vp, ci]
    E.idaens_set_device_controller.argtypes = [vp, ci]
    E.idaens_device_controller_active.argtypes = [vp]
    E.idaens_stream_group.argtypes = [C.POINTER(vp), ci, dp, ci, C.c_long, C.c_long, C.c_long, i64p]
    E.idaens_solve_schedule_group.argtypes = [C.POINTER(vp), ci, dp, ci, C.POINTER(dp), C.POINTER(i32p), C.POINTER(i32p), C.c_long]
    E.idaens_set_roots.argtypes = [vp, ci, i32p, dp]
    E.idaens_set_root_fn.argtypes = [vp, ci, ROOT_FN, vp]
    E.idaens_get_roots.argtypes = [vp, i32p]
    E.idaens_solve.argtypes = [vp, cd, ci, dp, i32p, C.c_long]
    E.idaens_solve_schedule.argtypes = [vp, dp, ci, dp, i32p, i32p, dp, dp, C.c_long]
    E.idaens_stream.argtypes = [vp, dp, ci, C.c_long, C.c_long, i64p]
    E.idaens_get_counter.argtypes = [vp, ci, i64p]
    E.idaens_get_real.argtypes = [vp, ci, dp]
    E.idaens_get_yy.argtypes = [vp, dp]
    E.idaens_get_dky.argtypes = [vp, C.c_double, ci, dp, i32p]
    E.idaens_get_yp.argtypes = [vp, dp]
    E.idaens_total_newton_iters.argtypes = [vp]
    E.idaens_total_newton_iters.restype = C.c_int64
    E.idaens_total_rounds.argtypes = [vp]
    E.idaens_total_rounds.restype = C.c_int64
    E.idaens_trace_system.argtypes = [vp, ci]
    E.idaens_trace_len.argtypes = [vp]
    E.idaens_trace_len.restype = C.c_long
    E.idaens_trace_get.argtypes = [vp, dp]
    _libs = (H, E)
    return _libs


class IdaHipError(RuntimeError):
    pass


class Ctx:
    """One ensemble context on one device (idahip_ctx)."""

    def __init__(self, kind, n, batch, device=0, stream=None):
        self.H, self.E = load()
        self.n, self.batch = int(n), int(batch)
        self.kind = KIND[kind] if isinstance(kind, str) else int(kind)
        h = C.c_void_p()
        rc = self.H.idahip_create(C.byref(h), int(device), self.n, self.batch, self.kind, stream)
        if rc != 0 or not h.value:
            raise IdaHipError("idahip_create failed (%d) -- is a GPU visible?" % rc)
        self.h = h

    def close(self):
        """idahip_destroy. Refused while an Ensemble created on this ctx is still open: libidaens keeps the raw ctx pointer."""
        if getattr(self, "h", None) is not None and self.h.value:
            if getattr(self, "_ensembles", 0) > 0:
                raise IdaHipError("close() of a Ctx with %d open Ensemble(s): close them first" % self._ensembles)
            self.H.idahip_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self._ensembles = 0  # garbage collection: any Ensemble holds a reference to this object, so none is left
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc < 0:
            self.fail("%s failed (%d): %s" % (what, rc, self.H.idahip_last_error(self.h).decode()))
        return rc

    def fail(self, message):
        """Raise IdaHipError; an exception raised inside a host callback (it could not cross the C boundary) is its cause."""
        cause, self._cb_error = getattr(self, "_cb_error", None), None
        if cause is not None:
            raise IdaHipError(message + " [Python callback raised %s: %s]" % (type(cause).__name__, cause)) from cause
        raise IdaHipError(message)

    def take_callback_error(self):
        """The exception a Python callback (residual, Jacobian or root function) raised since the last call of this method,
        or None; the stored exception is cleared, so that it is not attached to a later, unrelated error."""
        err, self._cb_error = getattr(self, "_cb_error", None), None
        return err

    def all_idx(self):
        return np.arange(self.batch, dtype=np.int32)

    # --- setup
    def set_tolerances(self, rtol, atol):
        a = _f64(np.atleast_1d(atol))
        self._chk(self.H.idahip_set_tolerances(self.h, float(rtol), _p(a), a.size), "set_tolerances")

    def set_problem_params(self, params, first=0):
        p = _f64(params).reshape(-1, _f64(params).shape[-1] if np.ndim(params) > 1 else 1)
        self._chk(self.H.idahip_set_problem_params(self.h, first, p.shape[0], _p(p), p.shape[1]), "set_problem_params")

    def set_linear_dense(self, A, B, c, first=0):
        """A, B: [count][n][n] stored column-major per system, i.e. A[s, j, i] = A_s(i, j); c: [count][n]."""
        A, B, c = _f64(A), _f64(B), _f64(c)
        self._chk(self.H.idahip_set_linear_dense(self.h, first, A.shape[0], _p(A), _p(B), _p(c)), "set_linear_dense")

    def set_host_problem(self, res, jac):
        """IDAHIP_HOST_CALLBACK: res(sys, t, yy, yp) -> residual vector; jac(sys, t, cj, yy, yp, res) -> J[n][n] (row, col), or
        None entries left zero. Python callables are wrapped into the C callbacks of include/ida_hip.h."""
        n = self.n

        self._cb_error = None  # the exception a callback raised: it cannot cross the C boundary, so it is kept and re-raised by _chk

        def c_res(sys, t, yy, yp, out, _user):
            try:
                r = res(sys, t, np.ctypeslib.as_array(yy, (n,)), np.ctypeslib.as_array(yp, (n,)))
                np.copyto(np.ctypeslib.as_array(out, (n,)), np.asarray(r, dtype=np.float64).reshape(n))
                return 0
            except Exception as e:
                self._cb_error = e
                return 1

        def c_jac(sys, t, cj, yy, yp, rv, J, _user):
            try:
                m = jac(sys, t, cj, np.ctypeslib.as_array(yy, (n,)), np.ctypeslib.as_array(yp, (n,)), np.ctypeslib.as_array(rv, (n,)))
                # m[i][j] = dF_i/dy_j (row, column); the library's matrix is column-major: J[j * n + i]
                np.copyto(np.ctypeslib.as_array(J, (n, n)), np.asarray(m, dtype=np.float64).reshape(n, n).T)
                return 0
            except Exception as e:
                self._cb_error = e
                return 1

        self._cb = (RES_FN(c_res), JAC_FN(c_jac))  # keep the thunks alive as long as the ctx
        self._chk(self.H.idahip_set_host_problem(self.h, self._cb[0], self._cb[1], None), "set_host_problem")

    def upload(self, field, arr, first=0):
        a = _f64(arr).reshape(-1, self.n)
        self._chk(self.H.idahip_upload(self.h, field, first, a.shape[0], _p(a)), "upload")

    def download(self, field, first=0, count=None):
        count = self.batch - first if count is None else count
        out = np.empty((count, self.n))
        self._chk(self.H.idahip_download(self.h, field, first, count, _p(out)), "download")
        return out

    def download_lu(self, sys):
        lu = np.empty(self.n * self.n)
        piv = np.empty(self.n, dtype=np.int64)
        self._chk(self.H.idahip_download_lu(self.h, sys, _p(lu), _p(piv, i64p)), "download_lu")
        return lu.reshape(self.n, self.n).T.copy(), piv  # logical (row, col) view of the column-major buffer

    # --- device buffers for the stand-alone LSolver calls
    def dev_array(self, host):
        host = np.ascontiguousarray(host)
        d = self.H.idahip_dev_alloc(self.h, host.nbytes)
        if not d:
            raise IdaHipError("device allocation of %d bytes failed" % host.nbytes)
        self._chk(self.H.idahip_memcpy_h2d(self.h, d, host.ctypes.data_as(C.c_void_p), host.nbytes), "h2d")
        return d

    def dev_empty(self, nbytes):
        d = self.H.idahip_dev_alloc(self.h, nbytes)
        if not d:
            raise IdaHipError("device allocation of %d bytes failed" % nbytes)
        return d

    def to_host(self, d, shape, dtype=np.float64):
        out = np.empty(shape, dtype=dtype)
        self._chk(self.H.idahip_memcpy_d2h(self.h, out.ctypes.data_as(C.c_void_p), d, out.nbytes), "d2h")
        return out

    def dev_free(self, d):
        self.H.idahip_dev_free(self.h, d)

    # --- LSolver
    def ls_setup(self, dA, dPiv, idx=None):
        idx = self.all_idx() if idx is None else _i32(idx)
        info = np.zeros(idx.size, dtype=np.int32)
        rc = self._chk(self.H.idahip_ls_setup(self.h, dA, dPiv, _p(info, i32p), _p(idx, i32p), idx.size), "ls_setup")
        return rc, info

    def ls_solve(self, dLU, dPiv, dX, dB, idx=None):
        idx = self.all_idx() if idx is None else _i32(idx)
        self._chk(self.H.idahip_ls_solve(self.h, dLU, dPiv, dX, dB, 0.0, _p(idx, i32p), idx.size), "ls_solve")
        self._chk(self.H.idahip_sync(self.h), "sync")

    def wrms(self, dX, dW, idx=None):
        idx = self.all_idx() if idx is None else _i32(idx)
        out = np.zeros(idx.size)
        self._chk(self.H.idahip_wrms(self.h, dX, dW, _p(out), _p(idx, i32p), idx.size), "wrms")
        return out

    # --- NLProblem
    def nls_sys(self, tn, cj, reset_ee, idx=None):
        idx = self.all_idx() if idx is None else _i32(idx)
        tn, cj = _f64(np.broadcast_to(tn, idx.shape)), _f64(np.broadcast_to(cj, idx.shape))
        self._chk(self.H.idahip_nls_sys(self.h, _p(tn), _p(cj), int(reset_ee), _p(idx, i32p), idx.size), "nls_sys")

    def nls_lsetup(self, tn, cj, idx=None):
        idx = self.all_idx() if idx is None else _i32(idx)
        tn, cj = _f64(np.broadcast_to(tn, idx.shape)), _f64(np.broadcast_to(cj, idx.shape))
        info = np.zeros(idx.size, dtype=np.int32)
        rc = self._chk(self.H.idahip_nls_lsetup(self.h, _p(tn), _p(cj), _p(info, i32p), _p(idx, i32p), idx.size), "nls_lsetup")
        return rc, info

    def nls_sys_setup(self, tn, cj, reset_ee=True, idx=None):
        """sys immediately followed by setup (one pass over A and B for the linear dense problem)."""
        idx = self.all_idx() if idx is None else _i32(idx)
        tn, cj = _f64(np.broadcast_to(tn, idx.shape)), _f64(np.broadcast_to(cj, idx.shape))
        info = np.zeros(idx.size, dtype=np.int32)
        rc = self._chk(self.H.idahip_nls_sys_setup(self.h, _p(tn), _p(cj), int(bool(reset_ee)), _p(info, i32p), _p(idx, i32p), idx.size),
                       "nls_sys_setup")
        return rc, info

    def newton_iter(self, scale, idx=None):
        idx = self.all_idx() if idx is None else _i32(idx)
        scale = _f64(np.broadcast_to(scale, idx.shape))
        out = np.zeros(idx.size)
        self._chk(self.H.idahip_newton_iter(self.h, _p(scale), _p(out), _p(idx, i32p), idx.size), "newton_iter")
        return out

    # --- stepper vector ops
    def init_first(self, idx=None):
        idx = self.all_idx() if idx is None else _i32(idx)
        a, b = np.zeros(idx.size), np.zeros(idx.size)
        self._chk(self.H.idahip_init_first(self.h, _p(a), _p(b), _p(idx, i32p), idx.size), "init_first")
        return a, b

    def scale_phi1(self, fac, idx=None):
        idx = self.all_idx() if idx is None else _i32(idx)
        fac = _f64(np.broadcast_to(fac, idx.shape))
        self._chk(self.H.idahip_scale_phi1(self.h, _p(fac), _p(idx, i32p), idx.size), "scale_phi1")

    def predict(self, kk, ns, beta, gamma, idx=None):
        idx = self.all_idx() if idx is None else _i32(idx)
        kkns = _i32(np.stack([np.broadcast_to(kk, idx.shape), np.broadcast_to(ns, idx.shape)], axis=1))
        beta = _f64(np.broadcast_to(beta, (idx.size, 6)))
        gamma = _f64(np.broadcast_to(gamma, (idx.size, 6)))
        self._chk(self.H.idahip_predict(self.h, _p(kkns, i32p), _p(beta), _p(gamma), _p(idx, i32p), idx.size), "predict")

    def post_newton(self, cj, kk, idx=None):
        idx = self.all_idx() if idx is None else _i32(idx)
        cj = _f64(np.broadcast_to(cj, idx.shape))
        kk = _i32(np.broadcast_to(kk, idx.shape))
        out = np.zeros((idx.size, 4))
        self._chk(self.H.idahip_post_newton(self.h, _p(cj), _p(kk, i32p), _p(out), _p(idx, i32p), idx.size), "post_newton")
        return out

    def restore(self, kk, ns, cvals, idx=None):
        idx = self.all_idx() if idx is None else _i32(idx)
        kkns = _i32(np.stack([np.broadcast_to(kk, idx.shape), np.broadcast_to(ns, idx.shape)], axis=1))
        cvals = _f64(np.broadcast_to(cvals, (idx.size, 6)))
        self._chk(self.H.idahip_restore(self.h, _p(kkns, i32p), _p(cvals), _p(idx, i32p), idx.size), "restore")

    def complete_step(self, kused, ck, maxord=5, idx=None):
        idx = self.all_idx() if idx is None else _i32(idx)
        kused = _i32(np.broadcast_to(kused, idx.shape))
        ck = _f64(np.broadcast_to(ck, idx.shape))
        nrm = np.zeros(idx.size)
        bad = np.zeros(idx.size, dtype=np.int32)
        self._chk(self.H.idahip_complete_step(self.h, _p(kused, i32p), _p(ck), int(maxord), _p(nrm), _p(bad, i32p), _p(idx, i32p),
                                              idx.size), "complete_step")
        return nrm, bad

    def get_solution(self, kord, cvals, dvals, idx=None):
        idx = self.all_idx() if idx is None else _i32(idx)
        kord = _i32(np.broadcast_to(kord, idx.shape))
        cvals = _f64(np.broadcast_to(cvals, (idx.size, 6)))
        dvals = _f64(np.broadcast_to(dvals, (idx.size, 5)))
        self._chk(self.H.idahip_get_solution(self.h, _p(kord, i32p), _p(cvals), _p(dvals), _p(idx, i32p), idx.size), "get_solution")

    def pow_batch(self, x, y):
        """The device controller's pow (glibc_pow.hpp) for arrays of arguments."""
        x, y = _f64(x), _f64(y)
        out = np.empty_like(x)
        self._chk(self.H.idahip_pow_batch(self.h, _p(x), _p(y), _p(out), x.size), "pow_batch")
        return out

    def set_lu_superpanel(self, on):
        """n > 1024: a super-panel in one launch (banded matrices; default for heat1d) or panel by panel (dense matrices)."""
        self._chk(self.H.idahip_set_lu_superpanel(self.h, int(on)), "set_lu_superpanel")

    def lu_superpanel(self):
        return int(self.H.idahip_lu_superpanel(self.h))

    def set_lu_period(self, rounds):
        """Device lock-step stepper: a round postpones its linear setups unless (k - 1) / k of the stepping systems ask for one, at most k - 1 rounds in a row (results unchanged)."""
        self._chk(self.H.idahip_set_lu_period(self.h, int(rounds)), "set_lu_period")

    def lu_period(self):
        return int(self.H.idahip_lu_period(self.h))

    def set_lu_variant(self, variant):
        self._chk(self.H.idahip_set_lu_variant(self.h, int(variant)), "set_lu_variant")

    # --- measurement
    def timing(self, on):
        self.H.idahip_timing_enable(self.h, int(on))

    def timing_reset(self):
        self.H.idahip_timing_reset(self.h)

    def timing_get(self):
        out = {}
        for k, name in enumerate(K_NAMES):
            ms, la, sy = C.c_double(), C.c_int64(), C.c_int64()
            self.H.idahip_timing_get(self.h, k, C.byref(ms), C.byref(la), C.byref(sy))
            out[name] = {"ms": ms.value, "launches": la.value, "systems": sy.value}
        return out


COUNTERS = {"nst": 0, "nre": 1, "nje": 2, "nsetups": 3, "nni": 4, "netf": 5, "ncfn": 6, "n_attempts": 7, "nls_nconvfails": 8,
            "kused": 9, "kk": 10, "nge": 11, "nlufail": 12, "nconv_jcur": 13, "nfail_first": 14, "nli": 15, "ncfl": 16}
REALS = {"tn": 0, "hused": 1, "hh": 2, "h0u": 3, "tolsf": 4}


class Ensemble:
    """Batched counterpart of the reference's `Ida` object: Ida::new / Ida::solve / getters for every system of a Ctx."""

    def __init__(self, ctx, yy0, yp0):
        self.ctx = ctx
        self.E = ctx.E
        yy0, yp0 = _f64(yy0).reshape(ctx.batch, ctx.n), _f64(yp0).reshape(ctx.batch, ctx.n)
        h = C.c_void_p()
        rc = self.E.idaens_create(C.byref(h), ctx.h, _p(yy0), _p(yp0))
        if rc != 0:
            self.ctx.fail("idaens_create failed (%d): %s" % (rc, ctx.H.idahip_last_error(ctx.h).decode()))
        self.h = h
        ctx._ensembles = getattr(ctx, "_ensembles", 0) + 1

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            self.E.idaens_destroy(self.h)
            self.h = C.c_void_p()
            self.ctx._ensembles -= 1

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_max_num_steps(self, mxstep):
        self.E.idaens_set_max_num_steps(self.h, int(mxstep))

    def set_roots(self, comps, thresholds):
        """Root functions g_i = y[comps[i]] - thresholds[i] for every system (before the first solve)."""
        comps, thr = _i32(comps), _f64(thresholds)
        if self.E.idaens_set_roots(self.h, comps.size, _p(comps, i32p), _p(thr)) != 0:
            raise IdaHipError("set_roots: %s" % (self.E.idaens_last_error(self.h) or b"").decode())
        self.nroots = comps.size

    def set_root_fn(self, nroots, fn):
        """Root::root for any function: fn(sys, t, yy, yp) -> nroots values g_i (yy, yp: numpy views valid during the call).
        An exception inside fn fails that system with IDAENS_RTFUNC_FAIL (-12); it is kept on the Ctx and handed out -- once --
        by `Ctx.take_callback_error()` (solve / solve_schedule do not raise for a per-system status)."""
        ens = self

        def thunk(user, sys, t, yy, yp, nr, gout):
            try:
                n = ens.ctx.n
                g = fn(int(sys), float(t), np.ctypeslib.as_array(yy, (n,)), np.ctypeslib.as_array(yp, (n,)))
                np.copyto(np.ctypeslib.as_array(gout, (nr,)), np.asarray(g, dtype=np.float64).reshape(nr))
                return 0
            except Exception as e:  # cannot cross the C boundary
                ens.ctx._cb_error = e
                return 1

        self._root_thunk = ROOT_FN(thunk)  # keep it alive as long as the ensemble
        if self.E.idaens_set_root_fn(self.h, int(nroots), self._root_thunk, None) != 0:
            raise IdaHipError("set_root_fn: %s" % (self.E.idaens_last_error(self.h) or b"").decode())
        self.nroots = int(nroots)

    def roots_found(self):
        out = np.zeros((self.ctx.batch, getattr(self, "nroots", 0)), dtype=np.int32)
        if out.size:
            self.E.idaens_get_roots(self.h, _p(out, i32p))
        return out

    def set_fused_newton(self, on):
        self.E.idaens_set_fused_newton(self.h, int(on))

    def set_device_controller(self, on):
        """Small device problems: the whole of Ida::solve in one launch (default on) or the lock-step host stepper."""
        rc = self.E.idaens_set_device_controller(self.h, int(on))
        if rc != 0:  # refused: the device pow does not have this host's bits, or its self-check could not run
            raise IdaHipError("set_device_controller(%d) refused (%d): %s" % (int(on), rc, self.E.idaens_last_error(self.h).decode()))

    def device_controller_active(self):
        """0 = host stepper, 1 = device stepper with one thread per system, 2 = device lock-step rounds (what a NORMAL solve would use)."""
        return int(self.E.idaens_device_controller_active(self.h))

    def set_max_ord(self, maxord):
        if self.E.idaens_set_max_ord(self.h, int(maxord)) != 0:
            raise IdaHipError("set_max_ord(%d) rejected" % maxord)

    def solve(self, tout, itask=0, max_rounds=0):
        tret = np.zeros(self.ctx.batch)
        status = np.zeros(self.ctx.batch, dtype=np.int32)
        rc = self.E.idaens_solve(self.h, float(tout), int(itask), _p(tret), _p(status, i32p), int(max_rounds))
        if rc < 0:
            self.ctx.fail("idaens_solve failed (%d): %s" % (rc, self.E.idaens_last_error(self.h).decode()))
        return status, tret

    def solve_schedule(self, touts, max_rounds=0, outputs=False):
        """Ida::solve(touts[0]), solve(touts[1]), ... per system without the systems waiting for each other.
        -> (status, tret, reached[, yy_out, yp_out]) with yy_out/yp_out of shape (ntout, batch, n) when outputs is set."""
        touts = _f64(np.atleast_1d(touts))
        B, n = self.ctx.batch, self.ctx.n
        tret = np.zeros(B)
        status = np.zeros(B, dtype=np.int32)
        reached = np.zeros(B, dtype=np.int32)
        yo = np.full((touts.size, B, n), np.nan) if outputs else None
        ypo = np.full((touts.size, B, n), np.nan) if outputs else None
        rc = self.E.idaens_solve_schedule(self.h, _p(touts), touts.size, _p(tret), _p(status, i32p), _p(reached, i32p),
                                          _p(yo) if outputs else None, _p(ypo) if outputs else None, int(max_rounds))
        if rc < 0:
            self.ctx.fail("idaens_solve_schedule failed (%d): %s" % (rc, (self.E.idaens_last_error(self.h) or b"").decode()))
        return (status, tret, reached, yo, ypo) if outputs else (status, tret, reached)

    def stream(self, touts, max_rounds, stagger_rounds=0):
        """Throughput mode: max_rounds lock-step rounds of the schedule with finished systems restarting at once
        (stagger_rounds: spread the first starts over that many rounds). -> integrations completed since the ensemble
        was created."""
        touts = _f64(np.atleast_1d(touts))
        done = C.c_int64(0)
        rc = self.E.idaens_stream(self.h, _p(touts), touts.size, int(max_rounds), int(stagger_rounds), C.byref(done))
        if rc < 0:
            self.ctx.fail("idaens_stream failed (%d): %s" % (rc, (self.E.idaens_last_error(self.h) or b"").decode()))
        return done.value

    def counter(self, name):
        out = np.zeros(self.ctx.batch, dtype=np.int64)
        assert self.E.idaens_get_counter(self.h, COUNTERS[name], _p(out, i64p)) == 0
        return out

    def counters(self):
        return {k: self.counter(k) for k in COUNTERS}

    def real(self, name):
        out = np.zeros(self.ctx.batch)
        assert self.E.idaens_get_real(self.h, REALS[name], _p(out)) == 0
        return out

    def yy(self):
        out = np.empty((self.ctx.batch, self.ctx.n))
        assert self.E.idaens_get_yy(self.h, _p(out)) == 0
        return out

    def yp(self):
        out = np.empty((self.ctx.batch, self.ctx.n))
        assert self.E.idaens_get_yp(self.h, _p(out)) == 0
        return out

    def get_dky(self, t, k):
        """Ida::get_dky(t, k) for every system -> (status[batch], dky[batch][n]); rows with a bad status are NaN."""
        out = np.full((self.ctx.batch, self.ctx.n), np.nan)
        status = np.zeros(self.ctx.batch, dtype=np.int32)
        rc = self.E.idaens_get_dky(self.h, float(t), int(k), _p(out), _p(status, i32p))
        if rc < 0:
            self.ctx.fail("idaens_get_dky failed (%d): %s" % (rc, (self.E.idaens_last_error(self.h) or b"").decode()))
        return status, out

    def total_newton_iters(self):
        return int(self.E.idaens_total_newton_iters(self.h))

    def total_rounds(self):
        return int(self.E.idaens_total_rounds(self.h))

    def trace_system(self, sys):
        self.E.idaens_trace_system(self.h, int(sys))

    def trace(self):
        k = self.E.idaens_trace_len(self.h)
        out = np.zeros((k, 3))
        if k:
            self.E.idaens_trace_get(self.h, _p(out))
        return out


def _group_handles(ensembles):
    arr = (C.c_void_p * len(ensembles))(*[e.h for e in ensembles])
    return arr


def stream_group(ensembles, touts, max_rounds, stagger_rounds=0, offset_us=0):
    """idaens_stream_group: Ensemble.stream for every ensemble of the group at once (one host thread and one HIP stream per
    ensemble, all on one device; group g starts g * offset_us microseconds after group 0). -> integrations completed, per group."""
    touts = _f64(np.atleast_1d(touts))
    E = ensembles[0].E
    done = np.zeros(len(ensembles), dtype=np.int64)
    rc = E.idaens_stream_group(_group_handles(ensembles), len(ensembles), _p(touts), touts.size, int(max_rounds), int(stagger_rounds),
                               int(offset_us), _p(done, i64p))
    if rc < 0:
        texts = [(e.E.idaens_last_error(e.h) or b"").decode() for e in ensembles]
        raise IdaHipError("idaens_stream_group failed (%d): %s" % (rc, " | ".join(t for t in texts if t)))
    return done


def solve_schedule_group(ensembles, touts, max_rounds=0):
    """idaens_solve_schedule_group: Ensemble.solve_schedule for every ensemble of the group at once.
    -> [(status, tret, reached)] per group."""
    touts = _f64(np.atleast_1d(touts))
    E = ensembles[0].E
    G = len(ensembles)
    tret = [np.zeros(e.ctx.batch) for e in ensembles]
    status = [np.zeros(e.ctx.batch, dtype=np.int32) for e in ensembles]
    reached = [np.zeros(e.ctx.batch, dtype=np.int32) for e in ensembles]
    pt = (dp * G)(*[_p(a) for a in tret])
    ps = (i32p * G)(*[_p(a, i32p) for a in status])
    pr = (i32p * G)(*[_p(a, i32p) for a in reached])
    rc = E.idaens_solve_schedule_group(_group_handles(ensembles), G, _p(touts), touts.size, pt, ps, pr, int(max_rounds))
    if rc < 0:
        texts = [(e.E.idaens_last_error(e.h) or b"").decode() for e in ensembles]
        raise IdaHipError("idaens_solve_schedule_group failed (%d): %s" % (rc, " | ".join(t for t in texts if t)))
    return list(zip(status, tret, reached))


def concurrent_streams(count, device=0):
    """idahip_concurrent_streams: `count` HIP streams for the contexts of a group, chosen so that the device really runs them
    side by side. -> (list of stream handles for Ctx(..., stream=s), number of mutually concurrent ones)."""
    H, _ = load()
    arr = (C.c_void_p * count)()
    nc = C.c_int(0)
    rc = H.idahip_concurrent_streams(int(device), int(count), arr, C.byref(nc))
    if rc != 0:
        raise IdaHipError("idahip_concurrent_streams failed (%d)" % rc)
    return [C.c_void_p(arr[i]) for i in range(count)], nc.value


def release_streams(streams, device=0):
    H, _ = load()
    arr = (C.c_void_p * len(streams))(*[s.value if isinstance(s, C.c_void_p) else s for s in streams])
    H.idahip_release_streams(int(device), len(streams), arr)


def stream_pair_share(a, b, device=0):
    """idahip_stream_pair_share: the part of two chip-filling probe grids' joint span in which both streams had workgroups running."""
    H, _ = load()
    f = C.c_double(0.0)
    rc = H.idahip_stream_pair_share(int(device), a, b, C.byref(f))
    if rc != 0:
        raise IdaHipError("idahip_stream_pair_share failed (%d)" % rc)
    return f.value
