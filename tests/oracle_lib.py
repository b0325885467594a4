"""ctypes loader for the CPU oracle (oracle/libida_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product path (rust-ida_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_LIB = None

dp = C.POINTER(C.c_double)
i64p = C.POINTER(C.c_int64)
i32p = C.POINTER(C.c_int32)


def build(force=False):
    so = os.path.join(ORACLE_DIR, "libida_oracle.so")
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("capi.cpp", "dense.hpp", "newton.hpp", "problems.hpp", "ida.hpp", "Makefile")]
    stale = force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-B", "libida_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def _ptr(a, t=dp):
    return None if a is None else a.ctypes.data_as(t)


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    so = build()
    try:
        L = C.CDLL(so)
    except OSError:
        L = C.CDLL(build(force=True))
    L.oracle_dense_getrf.argtypes = [dp, C.c_int, C.c_int, i64p]
    L.oracle_dense_getrf.restype = C.c_int
    L.oracle_dense_getrs.argtypes = [dp, C.c_int, i64p, dp]
    L.oracle_dense_getrs.restype = None
    L.oracle_norm_wrms.argtypes = [dp, dp, C.c_int]
    L.oracle_norm_wrms.restype = C.c_double
    L.oracle_norm_wrms_masked.argtypes = [dp, dp, C.POINTER(C.c_uint8), C.c_int]
    L.oracle_norm_wrms_masked.restype = C.c_double
    L.oracle_dense_lsolver.argtypes = [dp, C.c_int, dp, dp, i64p]
    L.oracle_dense_lsolver.restype = C.c_int
    L.oracle_newton_test.argtypes = [dp, dp, C.c_double, C.c_int, dp, C.POINTER(C.c_long), C.POINTER(C.c_long)]
    L.oracle_newton_test.restype = C.c_int
    L.oracle_ida_create.argtypes = [C.c_int, C.c_int, dp, dp, dp, dp, dp, dp, C.c_double, dp, C.c_int]
    L.oracle_ida_create.restype = C.c_void_p
    L.oracle_ida_destroy.argtypes = [C.c_void_p]
    L.oracle_ida_destroy.restype = None
    L.oracle_ida_solve.argtypes = [C.c_void_p, C.c_double, dp, C.c_int]
    L.oracle_ida_solve.restype = C.c_int
    L.oracle_ida_get_scalar.argtypes = [C.c_void_p, C.c_char_p, dp]
    L.oracle_ida_get_scalar.restype = C.c_int
    L.oracle_ida_set_scalar.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
    L.oracle_ida_set_scalar.restype = C.c_int
    L.oracle_ida_get_vec.argtypes = [C.c_void_p, C.c_char_p, dp, C.c_int]
    L.oracle_ida_get_vec.restype = C.c_int
    L.oracle_ida_set_vec.argtypes = [C.c_void_p, C.c_char_p, dp, C.c_int]
    L.oracle_ida_set_vec.restype = C.c_int
    L.oracle_ida_set_coeffs.argtypes = [C.c_void_p]
    L.oracle_ida_set_coeffs.restype = C.c_double
    L.oracle_ida_predict.argtypes = [C.c_void_p]
    L.oracle_ida_predict.restype = None
    L.oracle_ida_restore.argtypes = [C.c_void_p, C.c_double]
    L.oracle_ida_restore.restype = None
    L.oracle_ida_test_error.argtypes = [C.c_void_p, C.c_double, dp, dp]
    L.oracle_ida_test_error.restype = C.c_int
    L.oracle_ida_complete_step.argtypes = [C.c_void_p, C.c_double, C.c_double]
    L.oracle_ida_complete_step.restype = None
    L.oracle_ida_get_solution.argtypes = [C.c_void_p, C.c_double]
    L.oracle_ida_get_solution.restype = C.c_int
    L.oracle_ida_get_dky.argtypes = [C.c_void_p, C.c_double, C.c_int, C.POINTER(C.c_double), C.c_int]
    L.oracle_ida_get_dky.restype = C.c_int
    L.oracle_ida_nonlinear_solve.argtypes = [C.c_void_p]
    L.oracle_ida_nonlinear_solve.restype = C.c_int
    L.oracle_ida_lsetup.argtypes = [C.c_void_p]
    L.oracle_ida_lsetup.restype = C.c_int
    L.oracle_ida_record_steps.argtypes = [C.c_void_p, C.c_int]
    L.oracle_ida_record_steps.restype = None
    L.oracle_ida_num_recorded.argtypes = [C.c_void_p]
    L.oracle_ida_num_recorded.restype = C.c_long
    L.oracle_ida_get_recorded.argtypes = [C.c_void_p, dp]
    L.oracle_ida_get_recorded.restype = None
    L.oracle_run_ensemble.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, dp, dp, dp, dp, dp, dp, C.c_double, dp, C.c_int,
                                      dp, C.c_int, C.c_int, dp, dp, dp, i32p, dp]
    L.oracle_run_ensemble.restype = C.c_double
    L.oracle_time_lu_solve.argtypes = [dp, dp, C.c_int, C.c_int, C.c_int, i32p]
    L.oracle_time_lu_solve.restype = C.c_double
    L.oracle_hardware_concurrency.argtypes = []
    L.oracle_hardware_concurrency.restype = C.c_int
    _LIB = L
    return L


KIND = {"roberts": 0, "lorenz63": 1, "linear_dense": 2, "heat1d": 3, "dummy": 4}


def f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def getrf(a_rowmajor_logical):
    """a: (m, n) logical matrix. Returns (info, LU logical (m,n), pivots)."""
    a = f64(a_rowmajor_logical)
    m, n = a.shape
    cm = np.asfortranarray(a).copy(order="F")  # column-major storage
    piv = np.zeros(n, dtype=np.int64)
    info = lib().oracle_dense_getrf(cm.ctypes.data_as(dp), m, n, _ptr(piv, i64p))
    return info, np.array(cm), piv


def getrs(lu_logical, piv, b):
    lu = np.asfortranarray(f64(lu_logical)).copy(order="F")
    n = lu.shape[1]
    x = f64(b).copy()
    piv = np.ascontiguousarray(piv, dtype=np.int64)
    lib().oracle_dense_getrs(lu.ctypes.data_as(dp), n, _ptr(piv, i64p), _ptr(x))
    return x


def wrms(x, w):
    x, w = f64(x), f64(w)
    return lib().oracle_norm_wrms(_ptr(x), _ptr(w), x.size)


class OracleIda:
    """One reference-style `Ida` object (one IVP)."""

    def __init__(self, kind, n, yy0, yp0, rtol, atol, params=None, A=None, B=None, c=None):
        self.L = lib()
        self.n = n
        self._keep = [f64(yy0), f64(yp0), f64(np.atleast_1d(atol)),
                      None if params is None else f64(params),
                      None if A is None else f64(A), None if B is None else f64(B), None if c is None else f64(c)]
        yy0_, yp0_, atol_, p_, A_, B_, c_ = self._keep
        self.h = self.L.oracle_ida_create(KIND[kind], n, _ptr(p_), _ptr(A_), _ptr(B_), _ptr(c_), _ptr(yy0_), _ptr(yp0_),
                                          float(rtol), _ptr(atol_), atol_.size)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.oracle_ida_destroy(self.h)
            self.h = None

    def solve(self, tout, itask=0):
        tret = C.c_double(0.0)
        st = self.L.oracle_ida_solve(self.h, float(tout), C.byref(tret), itask)
        return st, tret.value

    def get(self, name):
        out = C.c_double(0.0)
        assert self.L.oracle_ida_get_scalar(self.h, name.encode(), C.byref(out)) == 0, name
        return out.value

    def set(self, name, v):
        assert self.L.oracle_ida_set_scalar(self.h, name.encode(), float(v)) == 0, name

    def getv(self, name):
        buf = np.zeros(max(6 * self.n, self.n * self.n, 8), dtype=np.float64)
        k = self.L.oracle_ida_get_vec(self.h, name.encode(), _ptr(buf), buf.size)
        assert k >= 0, name
        return buf[:k].copy()

    def setv(self, name, v):
        v = f64(v).ravel()
        assert self.L.oracle_ida_set_vec(self.h, name.encode(), _ptr(v), v.size) == 0, name

    def get_dky(self, t, k, literal_q9=False):
        """IDAGetDky (src/lib.rs:424-529) -> (status, dky)."""
        out = np.zeros(self.n)
        st = self.L.oracle_ida_get_dky(self.h, float(t), int(k), _ptr(out), int(literal_q9))
        return st, out

    def counters(self):
        return {k: int(self.get(k)) for k in ("nst", "nre", "nje", "nsetups", "nni", "netf", "ncfn", "nge", "n_attempts",
                                              "nls_nconvfails")}

    def recorded_steps(self):
        k = self.L.oracle_ida_num_recorded(self.h)
        out = np.zeros((k, 5))
        if k:
            self.L.oracle_ida_get_recorded(self.h, _ptr(out))
        return out


def run_ensemble(kind, n, yy0, yp0, rtol, atol, touts, params=None, A=None, B=None, c=None, nthreads=1):
    """Integrate nsys independent IVPs with the oracle. Returns dict(yy, yp, counters, status, kused, hused, seconds)."""
    L = lib()
    yy0, yp0 = f64(yy0), f64(yp0)
    nsys = yy0.shape[0]
    atol_ = f64(np.atleast_1d(atol))
    touts = f64(np.atleast_1d(touts))
    p_ = None if params is None else f64(params).reshape(nsys, -1)
    A_ = None if A is None else f64(A)
    B_ = None if B is None else f64(B)
    c_ = None if c is None else f64(c)
    yy = np.zeros((touts.size, nsys, n))
    yp = np.zeros((touts.size, nsys, n))
    counters = np.zeros((nsys, 8))
    status = np.zeros(nsys, dtype=np.int32)
    kh = np.zeros((nsys, 2))
    secs = L.oracle_run_ensemble(KIND[kind], n, nsys, 0 if p_ is None else p_.shape[1], _ptr(p_), _ptr(A_), _ptr(B_), _ptr(c_),
                                 _ptr(yy0), _ptr(yp0), float(rtol), _ptr(atol_), atol_.size, _ptr(touts), touts.size,
                                 int(nthreads), _ptr(yy), _ptr(yp), _ptr(counters), _ptr(status, i32p), _ptr(kh))
    names = ["nst", "nre", "nje", "nsetups", "nni", "netf", "ncfn", "n_attempts"]
    return {"yy": yy, "yp": yp, "counters": {k: counters[:, i].astype(np.int64) for i, k in enumerate(names)},
            "status": status, "kused": kh[:, 0].astype(np.int64), "hused": kh[:, 1], "seconds": secs}
