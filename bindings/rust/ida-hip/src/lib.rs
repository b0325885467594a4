//! rust-ida's solver traits on MI355X.
//!
//! * [`HipDense`] implements `linear::LSolver<f64, D>` (crates/linear/src/traits.rs:27-91) with the batched LU and
//!   triangular solves of libidahip at batch = 1: `setup` = `dense_get_rf` (dense.rs:86-158), `solve` = `dense_get_rs`
//!   (dense.rs:165-206), bit-identical factors, pivots and solutions.
//! * [`HipEnsemble`] is the batched counterpart of `Ida`: B independent IVPs stepped in lock-step rounds by libidaens, each
//!   system taking exactly the steps the reference's `Ida::solve` takes for it (same step sizes, orders, counters, bits).
//! * [`HostProblem`] carries any `IdaProblem`-style residual / Jacobian pair (src/traits.rs:12-70) across the C ABI as
//!   host callbacks (problem kind `IDAHIP_HOST_CALLBACK`); the four built-in device problems are selected by [`Problem`].
//! * [`nls::HipNlsProblem`] implements `nonlinear::NLProblem<f64, D>` (crates/nonlinear/src/traits.rs:5-127) on the
//!   ctx-resident `IdaNLProblem` state at batch = 1, [`nls::HipNewton`] implements `nonlinear::NLSolver<f64, D>`
//!   (traits.rs:129-209) -- the call shapes of src/lib.rs:819,833-840 -- and, with the cargo feature `ida-problem`,
//!   [`problem::IdaProblemAdapter`] turns any `ida::traits::IdaProblem` (src/traits.rs:12-94) into a [`HostProblem`].
//!
//! STATUS: experimental, never compiled. The build image of this repository has no Rust toolchain; the crate is checked only
//! lexically and against the C ABI (tests/test_rust_bindings.py: every `sys::` item exists in the generated bindings, which
//! match the headers and the exported symbols). Expect trait-bound and lifetime fixes at the first `cargo check`.
//!
//! Errors follow the reference: `linear::Error::LUFactFail { col }` with the 1-based column (dense.rs:121) for a zero pivot;
//! anything else the library reports (< 0: bad argument, HIP failure) is a programming or environment error and surfaces
//! as [`Error::Library`] with the library's message. Nothing unwinds across the FFI: a panic inside a user callback is
//! caught and turned into a failed call.
use std::ffi::CStr;
use std::marker::PhantomData;
use std::os::raw::{c_double, c_int, c_long, c_void};
use std::panic::{catch_unwind, AssertUnwindSafe};
use std::ptr;

pub mod nls;
#[cfg(feature = "ida-problem")]
pub mod problem;

use ida_hip_sys as sys;
use linear::{LSolver, LSolverType};
use nalgebra::{allocator::Allocator, DefaultAllocator, DimName, Matrix, StorageMut, U1};

/// Failure of a library call.
#[derive(Debug)]
pub enum Error {
    /// `linear::Error::LUFactFail` of one or more systems (1-based column, dense.rs:121), by system index.
    Singular(Vec<(usize, usize)>),
    /// The library refused the call or the device failed (`idahip_last_error` / `idaens_last_error`).
    Library { code: i32, message: String },
}

impl std::fmt::Display for Error {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        match self {
            Error::Singular(v) => write!(f, "singular matrix in LU factorisation (system, column): {:?}", v),
            Error::Library { code, message } => write!(f, "libidahip error {}: {}", code, message),
        }
    }
}
impl std::error::Error for Error {}

/// The problem a context integrates (`idahip_problem`).
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum Problem {
    /// src/sample_problems/roberts.rs (n = 3)
    Roberts,
    /// tests/lorenz63.rs parameters, index-0 DAE (n = 3); parameters [p, r, b] per system
    Lorenz63,
    /// F = A y' + B y - c with dense A, B per system (column-major)
    LinearDense,
    /// 1-D heat equation by the method of lines; parameter kappa / dx^2 per system
    Heat1D,
    /// any residual / Jacobian pair evaluated on the host ([`HostProblem`])
    HostCallback,
}

impl Problem {
    fn code(self) -> c_int {
        match self {
            Problem::Roberts => sys::IDAHIP_ROBERTS,
            Problem::Lorenz63 => sys::IDAHIP_LORENZ63,
            Problem::LinearDense => sys::IDAHIP_LINEAR_DENSE,
            Problem::Heat1D => sys::IDAHIP_HEAT1D,
            Problem::HostCallback => sys::IDAHIP_HOST_CALLBACK,
        }
    }
}

/// Residual and Jacobian of a user problem, evaluated on the host for one system at a time
/// (`Residual::res`, `Jacobian::jac`: src/traits.rs:28-37, 58-69). `jac` receives a zeroed column-major n x n matrix.
pub trait HostProblem {
    fn res(&self, sys: usize, tt: f64, yy: &[f64], yp: &[f64], resval: &mut [f64]);
    fn jac(&self, sys: usize, tt: f64, cj: f64, yy: &[f64], yp: &[f64], resvec: &[f64], jac_colmajor: &mut [f64]);
}

struct CallbackBox {
    n: usize,
    problem: Box<dyn HostProblem>,
}

unsafe extern "C" fn res_trampoline(s: c_int, tt: c_double, yy: *const c_double, yp: *const c_double, r: *mut c_double, user: *mut c_void) -> c_int {
    let cb = &*(user as *const CallbackBox);
    let n = cb.n;
    let out = catch_unwind(AssertUnwindSafe(|| {
        cb.problem.res(
            s as usize,
            tt,
            std::slice::from_raw_parts(yy, n),
            std::slice::from_raw_parts(yp, n),
            std::slice::from_raw_parts_mut(r, n),
        )
    }));
    if out.is_ok() {
        0
    } else {
        1
    }
}

unsafe extern "C" fn jac_trampoline(
    s: c_int,
    tt: c_double,
    cj: c_double,
    yy: *const c_double,
    yp: *const c_double,
    rv: *const c_double,
    j: *mut c_double,
    user: *mut c_void,
) -> c_int {
    let cb = &*(user as *const CallbackBox);
    let n = cb.n;
    let out = catch_unwind(AssertUnwindSafe(|| {
        cb.problem.jac(
            s as usize,
            tt,
            cj,
            std::slice::from_raw_parts(yy, n),
            std::slice::from_raw_parts(yp, n),
            std::slice::from_raw_parts(rv, n),
            std::slice::from_raw_parts_mut(j, n * n),
        )
    }));
    if out.is_ok() {
        0
    } else {
        1
    }
}

/// One device context: all device memory of an ensemble of `batch` systems of size `n` (RAII over `idahip_create`).
pub struct Ctx {
    raw: *mut sys::idahip_ctx,
    n: usize,
    batch: usize,
    callbacks: Option<Box<CallbackBox>>,
}

// A ctx may move between threads; every method takes `&mut self` like the reference's solver objects, so it is not Sync.
unsafe impl Send for Ctx {}

impl Ctx {
    pub fn new(device: i32, n: usize, batch: usize, problem: Problem) -> Result<Self, Error> {
        Self::new_on_stream(device, n, batch, problem, ptr::null_mut())
    }

    /// A context on a HIP stream of the caller's (one of `concurrent_streams`' for the ensembles of a group); null = its own.
    pub fn new_on_stream(device: i32, n: usize, batch: usize, problem: Problem, hip_stream: *mut c_void) -> Result<Self, Error> {
        let mut raw: *mut sys::idahip_ctx = ptr::null_mut();
        let rc = unsafe { sys::idahip_create(&mut raw, device, n as c_int, batch as c_int, problem.code(), hip_stream) };
        if rc != 0 || raw.is_null() {
            return Err(Error::Library { code: rc, message: "idahip_create failed (no GPU visible, or bad size)".to_string() });
        }
        Ok(Ctx { raw, n, batch, callbacks: None })
    }

    pub fn n(&self) -> usize {
        self.n
    }
    pub fn batch(&self) -> usize {
        self.batch
    }
    pub fn as_raw(&mut self) -> *mut sys::idahip_ctx {
        self.raw
    }

    pub(crate) fn last_error(&self) -> String {
        unsafe { CStr::from_ptr(sys::idahip_last_error(self.raw)).to_string_lossy().into_owned() }
    }
    /// host -> device copy through the library (no HIP runtime needed by the caller); the return code is checked
    pub(crate) fn h2d<T>(&self, d: *mut T, h: &[T]) -> Result<(), Error> {
        let rc = unsafe { sys::idahip_memcpy_h2d(self.raw, d as *mut c_void, h.as_ptr() as *const c_void, std::mem::size_of_val(h)) };
        self.check(rc).map(|_| ())
    }
    /// device -> host copy through the library; the return code is checked
    pub(crate) fn d2h<T>(&self, h: &mut [T], d: *const T) -> Result<(), Error> {
        let rc = unsafe { sys::idahip_memcpy_d2h(self.raw, h.as_mut_ptr() as *mut c_void, d as *const c_void, std::mem::size_of_val(h)) };
        self.check(rc).map(|_| ())
    }
    pub(crate) fn check(&self, rc: c_int) -> Result<c_int, Error> {
        if rc < 0 {
            Err(Error::Library { code: rc, message: self.last_error() })
        } else {
            Ok(rc)
        }
    }

    /// `TolControlSS` / `TolControlSV` (src/tol_control.rs): one absolute tolerance, or one per component.
    pub fn set_tolerances(&mut self, rtol: f64, atol: &[f64]) -> Result<(), Error> {
        let rc = unsafe { sys::idahip_set_tolerances(self.raw, rtol, atol.as_ptr(), atol.len() as c_int) };
        self.check(rc).map(|_| ())
    }

    /// Matrices and right-hand sides of systems `first .. first + count` of a [`Problem::LinearDense`] context
    /// (column-major per system).
    pub fn set_linear_dense(&mut self, first: usize, a: &[f64], b: &[f64], c: &[f64]) -> Result<(), Error> {
        let nn = self.n * self.n;
        assert!(a.len() == b.len() && a.len() % nn == 0 && c.len() == a.len() / self.n);
        let count = a.len() / nn;
        let rc = unsafe { sys::idahip_set_linear_dense(self.raw, first as c_int, count as c_int, a.as_ptr(), b.as_ptr(), c.as_ptr()) };
        self.check(rc).map(|_| ())
    }

    pub fn set_problem_params(&mut self, first: usize, params: &[f64], nparam: usize) -> Result<(), Error> {
        assert!(nparam > 0 && params.len() % nparam == 0);
        let rc = unsafe {
            sys::idahip_set_problem_params(self.raw, first as c_int, (params.len() / nparam) as c_int, params.as_ptr(), nparam as c_int)
        };
        self.check(rc).map(|_| ())
    }

    /// The user problem of a [`Problem::HostCallback`] context.
    pub fn set_host_problem(&mut self, problem: Box<dyn HostProblem>) -> Result<(), Error> {
        let cb = Box::new(CallbackBox { n: self.n, problem });
        let user = &*cb as *const CallbackBox as *mut c_void;
        let rc = unsafe { sys::idahip_set_host_problem(self.raw, Some(res_trampoline), Some(jac_trampoline), user) };
        self.check(rc)?;
        self.callbacks = Some(cb); // keeps the callbacks alive as long as the ctx
        Ok(())
    }

    /// 4 = default, 3 = cross-check pipeline; both bit-identical to dense_get_rf.
    pub fn set_lu_variant(&mut self, variant: i32) -> Result<(), Error> {
        let rc = unsafe { sys::idahip_set_lu_variant(self.raw, variant) };
        self.check(rc).map(|_| ())
    }

    /// n > 1024: a 64-column super-panel in one launch (banded Jacobians) or panel by panel (dense ones); results identical.
    pub fn set_lu_superpanel(&mut self, on: bool) -> Result<(), Error> {
        let rc = unsafe { sys::idahip_set_lu_superpanel(self.raw, on as i32) };
        self.check(rc).map(|_| ())
    }

    /// Device lock-step stepper: linear setups batched over rounds. With `rounds` = k > 1 a round postpones its setups unless
    /// (k - 1) / k of the stepping systems ask for one, at most k - 1 rounds in a row; results identical (a scheduling choice
    /// for ensembles whose factorisations are latency bound).
    pub fn set_lu_period(&mut self, rounds: i32) -> Result<(), Error> {
        let rc = unsafe { sys::idahip_set_lu_period(self.raw, rounds) };
        self.check(rc).map(|_| ())
    }
}

impl Drop for Ctx {
    fn drop(&mut self) {
        unsafe {
            sys::idahip_destroy(self.raw);
        }
    }
}

/// `linear::LSolver` on the device: the reference's `Dense<D>` (crates/linear/src/dense.rs:15-64) with the factorisation and
/// the substitutions run by libidahip. The matrix crosses PCIe on `setup` (factors come back in place, as the trait's
/// `&mut mat_a` demands) and on `solve`; inside an ensemble integration nothing does -- use [`HipEnsemble`] for that.
pub struct HipDense<D: DimName> {
    ctx: Ctx,
    d_a: *mut c_double,
    d_piv: *mut i64,
    d_x: *mut c_double,
    d_b: *mut c_double,
    pivots: Vec<i64>,
    _dim: PhantomData<D>,
}

impl<D: DimName> HipDense<D> {
    fn bytes(count: usize) -> usize {
        count * std::mem::size_of::<f64>()
    }
}

impl<D> LSolver<f64, D> for HipDense<D>
where
    D: DimName,
    DefaultAllocator: Allocator<f64, D> + Allocator<usize, D>,
{
    fn new() -> Self {
        let n = D::dim();
        let mut ctx = Ctx::new(0, n, 1, Problem::HostCallback).expect("no MI355X visible");
        let raw = ctx.as_raw();
        let (d_a, d_piv, d_x, d_b) = unsafe {
            (
                sys::idahip_dev_alloc(raw, Self::bytes(n * n)) as *mut c_double,
                sys::idahip_dev_alloc(raw, Self::bytes(n)) as *mut i64,
                sys::idahip_dev_alloc(raw, Self::bytes(n)) as *mut c_double,
                sys::idahip_dev_alloc(raw, Self::bytes(n)) as *mut c_double,
            )
        };
        assert!(!d_a.is_null() && !d_piv.is_null() && !d_x.is_null() && !d_b.is_null(), "device allocation failed");
        HipDense { ctx, d_a, d_piv, d_x, d_b, pivots: vec![0; n], _dim: PhantomData }
    }

    fn get_type(&self) -> LSolverType {
        // the library states its solver's type (include/ida_hip.h: idahip_ls_type); idaLsSolve branches on it (src/ida_ls.rs:316)
        match unsafe { sys::idahip_ls_type(self.ctx.as_raw()) } {
            1 => LSolverType::Iterative,
            2 => LSolverType::MatrixIterative,
            _ => LSolverType::Direct,
        }
    }

    fn setup<S>(&mut self, mat_a: &mut Matrix<f64, D, D, S>) -> Result<(), linear::Error>
    where
        S: StorageMut<f64, D, D>,
    {
        let n = D::dim();
        let raw = self.ctx.as_raw();
        // nalgebra stores column-major, which is the device layout of one system (include/ida_hip.h)
        let mut host: Vec<f64> = mat_a.iter().copied().collect();
        let idx = [0i32];
        let mut info = [0i32];
        let lib = |e: Error| -> linear::Error { panic!("{}", e) }; // the trait has no variant for a device failure
        self.ctx.h2d(self.d_a, &host).map_err(lib)?;
        let rc = unsafe { sys::idahip_ls_setup(raw, self.d_a, self.d_piv, info.as_mut_ptr(), idx.as_ptr(), 1) };
        assert!(rc >= 0, "{}", self.ctx.last_error());
        // dense_get_rf works in place: on a zero pivot the caller's matrix holds the columns eliminated so far (dense.rs:120-122),
        // so the (partially) factored matrix and the pivots go back in both cases
        self.ctx.d2h(&mut host, self.d_a as *const f64).map_err(lib)?;
        self.ctx.d2h(&mut self.pivots, self.d_piv as *const i64).map_err(lib)?;
        for (dst, src) in mat_a.iter_mut().zip(host.iter()) {
            *dst = *src;
        }
        if info[0] != 0 {
            return Err(linear::Error::LUFactFail { col: info[0] as usize }); // 1-based (dense.rs:121)
        }
        Ok(())
    }

    fn solve<SA, SB, SC>(
        &self,
        mat_a: &Matrix<f64, D, D, SA>,
        x: &mut Matrix<f64, D, U1, SB>,
        b: &Matrix<f64, D, U1, SC>,
        _tol: f64,
    ) -> Result<(), linear::Error>
    where
        SA: StorageMut<f64, D, D>,
        SB: StorageMut<f64, D>,
        SC: StorageMut<f64, D>,
    {
        let n = D::dim();
        let raw = self.ctx.raw;
        let lu: Vec<f64> = mat_a.iter().copied().collect();
        let rhs: Vec<f64> = b.iter().copied().collect();
        let mut out = vec![0.0f64; n];
        let idx = [0i32];
        let lib = |e: Error| -> linear::Error { panic!("{}", e) };
        // the trait hands the factors back in on every call: they are uploaded again, like Dense::solve reads mat_a
        self.ctx.h2d(self.d_a, &lu).map_err(lib)?;
        self.ctx.h2d(self.d_piv, &self.pivots).map_err(lib)?;
        self.ctx.h2d(self.d_b, &rhs).map_err(lib)?;
        let rc = unsafe { sys::idahip_ls_solve(raw, self.d_a, self.d_piv, self.d_x, self.d_b, 0.0, idx.as_ptr(), 1) };
        assert!(rc >= 0, "{}", self.ctx.last_error());
        self.ctx.d2h(&mut out, self.d_x as *const f64).map_err(lib)?;
        for (dst, src) in x.iter_mut().zip(out.iter()) {
            *dst = *src;
        }
        Ok(())
    }
}

impl<D: DimName> Drop for HipDense<D> {
    fn drop(&mut self) {
        let raw = self.ctx.raw;
        unsafe {
            sys::idahip_dev_free(raw, self.d_a as *mut c_void);
            sys::idahip_dev_free(raw, self.d_piv as *mut c_void);
            sys::idahip_dev_free(raw, self.d_x as *mut c_void);
            sys::idahip_dev_free(raw, self.d_b as *mut c_void);
        }
    }
}

/// Status of one system after a `solve` call (`IdaSolveStatus`, src/lib.rs:58-64, and the `IdaError` codes).
pub type Status = i32;

/// Counters of `Ida` (src/ida_io.rs:11-117) that `HipEnsemble::counter` returns per system.
#[derive(Clone, Copy, Debug)]
pub enum Counter {
    NumSteps,
    NumResEvals,
    NumJacEvals,
    NumLinSolvSetups,
    NumNonlinSolvIters,
    NumErrTestFails,
    NumNonlinSolvConvFails,
    LastOrder,
}

impl Counter {
    fn code(self) -> c_int {
        match self {
            Counter::NumSteps => sys::IDAENS_C_NST,
            Counter::NumResEvals => sys::IDAENS_C_NRE,
            Counter::NumJacEvals => sys::IDAENS_C_NJE,
            Counter::NumLinSolvSetups => sys::IDAENS_C_NSETUPS,
            Counter::NumNonlinSolvIters => sys::IDAENS_C_NNI,
            Counter::NumErrTestFails => sys::IDAENS_C_NETF,
            Counter::NumNonlinSolvConvFails => sys::IDAENS_C_NCFN,
            Counter::LastOrder => sys::IDAENS_C_KUSED,
        }
    }
}

/// `Ida` for a whole batch: `Ida::new` + `Ida::solve` + getters for every system of a [`Ctx`] (include/ida_ensemble.h).
pub struct HipEnsemble {
    raw: *mut sys::idaens,
    ctx: Ctx,
    root: Option<Box<RootBox>>, // the user's `Root` function, kept alive while the library holds a pointer to it
}

/// `Root::root` (src/traits.rs:72-90) for a batch: `gout = g(t, yy, yp)` of system `sys`. Evaluated on the host with y(t), y'(t)
/// interpolated on the device (`idaens_set_root_fn`); a panic fails that system with `IDAENS_RTFUNC_FAIL`.
pub trait HostRoot {
    fn num_roots(&self) -> usize;
    fn root(&self, sys: usize, t: f64, yy: &[f64], yp: &[f64], gout: &mut [f64]);
}

struct RootBox {
    n: usize,
    root: Box<dyn HostRoot>,
}

unsafe extern "C" fn root_trampoline(user: *mut c_void, s: i32, t: c_double, yy: *const c_double, yp: *const c_double, nroots: i32, gout: *mut c_double) -> c_int {
    let cb = &*(user as *const RootBox);
    let out = catch_unwind(AssertUnwindSafe(|| {
        cb.root.root(
            s as usize,
            t,
            std::slice::from_raw_parts(yy, cb.n),
            std::slice::from_raw_parts(yp, cb.n),
            std::slice::from_raw_parts_mut(gout, nroots as usize),
        )
    }));
    if out.is_ok() {
        0
    } else {
        1
    }
}

unsafe impl Send for HipEnsemble {}

impl HipEnsemble {
    /// `Ida::new(problem, yy0, yp0, ...)` for every system; `yy0`, `yp0` are `[batch][n]`.
    pub fn new(mut ctx: Ctx, yy0: &[f64], yp0: &[f64]) -> Result<Self, Error> {
        assert_eq!(yy0.len(), ctx.n * ctx.batch);
        assert_eq!(yp0.len(), ctx.n * ctx.batch);
        let mut raw: *mut sys::idaens = ptr::null_mut();
        let rc = unsafe { sys::idaens_create(&mut raw, ctx.as_raw(), yy0.as_ptr(), yp0.as_ptr()) };
        if rc != 0 || raw.is_null() {
            return Err(Error::Library { code: rc, message: ctx.last_error() });
        }
        Ok(HipEnsemble { raw, ctx, root: None })
    }

    fn last_error(&self) -> String {
        unsafe {
            let p = sys::idaens_last_error(self.raw);
            if p.is_null() {
                String::new()
            } else {
                CStr::from_ptr(p).to_string_lossy().into_owned()
            }
        }
    }

    /// `Ida::solve(tout, &mut tret, IdaTask::Normal)` for every system: (status, tret) per system.
    pub fn solve(&mut self, tout: f64) -> Result<(Vec<Status>, Vec<f64>), Error> {
        let b = self.ctx.batch;
        let mut tret = vec![0.0f64; b];
        let mut status = vec![0i32; b];
        let rc = unsafe { sys::idaens_solve(self.raw, tout, sys::IDAENS_NORMAL, tret.as_mut_ptr(), status.as_mut_ptr(), 0) };
        if rc < 0 {
            return Err(Error::Library { code: rc, message: self.last_error() });
        }
        Ok((status, tret))
    }

    /// Root functions g_i = y[comps[i]] - thresholds[i] (the family of examples/roberts.rs), before the first `solve`.
    pub fn set_roots(&mut self, comps: &[i32], thresholds: &[f64]) -> Result<(), Error> {
        assert_eq!(comps.len(), thresholds.len());
        let rc = unsafe { sys::idaens_set_roots(self.raw, comps.len() as c_int, comps.as_ptr(), thresholds.as_ptr()) };
        if rc != 0 {
            return Err(Error::Library { code: rc, message: self.last_error() });
        }
        Ok(())
    }

    /// Any `Root` implementor (src/traits.rs:72-90) as the root functions of every system, before the first `solve`.
    pub fn set_root_fn(&mut self, root: Box<dyn HostRoot>) -> Result<(), Error> {
        let nroots = root.num_roots();
        let mut boxed = Box::new(RootBox { n: self.ctx.n, root });
        let user = &mut *boxed as *mut RootBox as *mut c_void;
        let rc = unsafe { sys::idaens_set_root_fn(self.raw, nroots as c_int, Some(root_trampoline), user) };
        if rc != 0 {
            return Err(Error::Library { code: rc, message: self.last_error() });
        }
        self.root = Some(boxed);
        Ok(())
    }

    /// `Ida::get_yy` of every system, `[batch][n]`.
    pub fn yy(&mut self) -> Vec<f64> {
        let mut out = vec![0.0f64; self.ctx.n * self.ctx.batch];
        unsafe {
            sys::idaens_get_yy(self.raw, out.as_mut_ptr());
        }
        out
    }

    /// `Ida::get_yp` of every system, `[batch][n]`.
    pub fn yp(&mut self) -> Vec<f64> {
        let mut out = vec![0.0f64; self.ctx.n * self.ctx.batch];
        unsafe {
            sys::idaens_get_yp(self.raw, out.as_mut_ptr());
        }
        out
    }

    /// `Ida::get_dky(t, k, ..)` of every system: (status, `[batch][n]`).
    pub fn get_dky(&mut self, t: f64, k: usize) -> Result<(Vec<Status>, Vec<f64>), Error> {
        let mut out = vec![f64::NAN; self.ctx.n * self.ctx.batch];
        let mut status = vec![0i32; self.ctx.batch];
        let rc = unsafe { sys::idaens_get_dky(self.raw, t, k as c_int, out.as_mut_ptr(), status.as_mut_ptr()) };
        if rc < 0 {
            return Err(Error::Library { code: rc, message: self.last_error() });
        }
        Ok((status, out))
    }

    pub fn counter(&self, which: Counter) -> Vec<i64> {
        let mut out = vec![0i64; self.ctx.batch];
        unsafe {
            sys::idaens_get_counter(self.raw, which.code(), out.as_mut_ptr());
        }
        out
    }

    /// The context the ensemble integrates on (read-only: the ensemble keeps a pointer into it, so it cannot be replaced).
    pub fn ctx(&self) -> &Ctx {
        &self.ctx
    }

    /// `Ctx::set_lu_variant` of the ensemble's context.
    pub fn set_lu_variant(&mut self, variant: i32) -> Result<(), Error> {
        self.ctx.set_lu_variant(variant)
    }

    /// The device-resident steppers (default on) or the lock-step host stepper. `Err` when the library refuses to switch them
    /// on: its device `pow` does not have this host's `powf` bits, or the self-check could not run (`last_error` says which).
    pub fn set_device_controller(&mut self, on: bool) -> Result<(), Error> {
        let rc = unsafe { sys::idaens_set_device_controller(self.raw, on as c_int) };
        if rc == 0 {
            Ok(())
        } else {
            Err(Error::Library { code: rc, message: self.last_error() })
        }
    }

    /// 0 = host stepper, 1 = device stepper with one thread per system, 2 = device lock-step rounds: what `solve` would run on.
    pub fn device_controller_active(&self) -> i32 {
        unsafe { sys::idaens_device_controller_active(self.raw) as i32 }
    }
}

/// `count` HIP streams that the device runs side by side (`idahip_concurrent_streams`: the HIP runtime may put two ordinary
/// streams on one hardware queue); the second value is how many of them, from the front, are mutually concurrent. The caller
/// owns the handles: `release_streams` after the contexts created on them are gone.
pub fn concurrent_streams(device: i32, count: usize) -> Result<(Vec<*mut c_void>, usize), Error> {
    let mut out: Vec<*mut c_void> = vec![ptr::null_mut(); count];
    let mut nc: c_int = 0;
    let rc = unsafe { sys::idahip_concurrent_streams(device, count as c_int, out.as_mut_ptr(), &mut nc) };
    if rc != 0 {
        return Err(Error::Library { code: rc, message: "idahip_concurrent_streams failed".to_string() });
    }
    Ok((out, nc as usize))
}

pub fn release_streams(device: i32, streams: &mut [*mut c_void]) {
    unsafe {
        sys::idahip_release_streams(device, streams.len() as c_int, streams.as_mut_ptr());
    }
}

/// `idaens_stream_group`: throughput mode for several ensembles of ONE device at once, one host thread and HIP stream each
/// (every system is integrated exactly as alone; the groups fill each other's idle stretches of a lock-step round).
/// Returns the integrations completed per group.
pub fn stream_group(ens: &mut [HipEnsemble], touts: &[f64], max_rounds: i64, stagger_rounds: i64, offset_us: i64) -> Result<Vec<i64>, Error> {
    let mut raws: Vec<*mut sys::idaens> = ens.iter().map(|e| e.raw).collect();
    let mut done = vec![0i64; ens.len()];
    let rc = unsafe {
        sys::idaens_stream_group(raws.as_mut_ptr(), raws.len() as c_int, touts.as_ptr(), touts.len() as c_int, max_rounds as c_long,
                                 stagger_rounds as c_long, offset_us as c_long, done.as_mut_ptr())
    };
    if rc < 0 {
        let text = ens.iter().map(|e| e.last_error()).filter(|t| !t.is_empty()).collect::<Vec<_>>().join(" | ");
        return Err(Error::Library { code: rc, message: text });
    }
    Ok(done)
}

impl Drop for HipEnsemble {
    fn drop(&mut self) {
        unsafe {
            sys::idaens_destroy(self.raw); // before the ctx it points into (field order: `ctx` drops after this body)
        }
    }
}
