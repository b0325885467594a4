//! `nonlinear::NLProblem` and `nonlinear::NLSolver` on the device (crates/nonlinear/src/traits.rs:5-209).
//!
//! [`HipNlsProblem`] is the reference's `IdaNLProblem` (src/ida_nls.rs:27-266) with its vectors and its factored Jacobian
//! resident in a [`Ctx`] of batch = 1: `sys` = idaNlsResidual, `setup` = idaNlsLSetup + idaLsSetup, `solve` = idaNlsLSolve +
//! idaLsSolve, `ctest` = idaNlsConvTest. It can be handed to the reference's own `nonlinear::Newton` unchanged, or to
//! [`HipNewton`], whose `solve_on_device` runs each Newton iteration as ONE device call (`idahip_newton_iter`: negate, getrs,
//! cjratio scaling, `y += delta`, WRMS norm -- newton.rs:98-110 fused).
//!
//! Experimental, never compiled (see the crate documentation).
use std::marker::PhantomData;
use std::os::raw::{c_double, c_void};

use ida_hip_sys as sys;
use nalgebra::{allocator::Allocator, DefaultAllocator, DimName, Matrix, OVector, Storage, StorageMut, U1};
use nonlinear::{NLProblem, NLSolver};

use crate::{Ctx, Error};

const RATEMAX: f64 = 0.9; // src/ida_nls.rs:15

/// `IdaNLProblem` (src/ida_nls.rs:27-59) on a device context of batch = 1. The integrator sets `tn`, `cj`, `cjratio`, `ss`,
/// `toldel`, `eps_newt` before a nonlinear solve exactly as `Ida::nonlinear_solve` does (src/lib.rs:787-812), after uploading
/// `yypredict`, `yppredict` and `ewt` (`Ctx::upload`).
pub struct HipNlsProblem<D: DimName> {
    /// private: `d_x` / `d_w` below were allocated on this context and `Drop` frees them through it
    ctx: Ctx,
    pub tn: f64,
    pub cj: f64,
    pub cjold: f64,
    pub cjratio: f64,
    pub ss: f64,
    pub oldnrm: f64,
    pub toldel: f64,
    pub eps_newt: f64,
    /// residual evaluations / linear setups (ida_nre, ida_nsetups: src/ida_nls.rs:150,168)
    pub nre: usize,
    pub nsetups: usize,
    d_x: *mut c_double,
    d_w: *mut c_double,
    _dim: PhantomData<D>,
}

impl<D: DimName> HipNlsProblem<D> {
    /// `ctx` must have batch = 1 and n = D.
    pub fn new(mut ctx: Ctx) -> Self {
        assert_eq!(ctx.batch(), 1);
        assert_eq!(ctx.n(), D::dim());
        let bytes = D::dim() * std::mem::size_of::<f64>();
        let raw = ctx.as_raw();
        let (d_x, d_w) = unsafe { (sys::idahip_dev_alloc(raw, bytes) as *mut c_double, sys::idahip_dev_alloc(raw, bytes) as *mut c_double) };
        assert!(!d_x.is_null() && !d_w.is_null(), "device allocation failed");
        HipNlsProblem { ctx, tn: 0.0, cj: 0.0, cjold: 0.0, cjratio: 1.0, ss: 20.0, oldnrm: 0.0, toldel: 0.0, eps_newt: 0.0, nre: 0, nsetups: 0, d_x, d_w, _dim: PhantomData }
    }

    /// The device context (read-only: the problem's device buffers belong to it).
    pub fn ctx(&self) -> &Ctx {
        &self.ctx
    }

    fn upload_field(&mut self, field: i32, v: &[f64]) -> Result<(), Error> {
        let rc = unsafe { sys::idahip_upload(self.ctx.as_raw(), field, 0, 1, v.as_ptr()) };
        self.ctx.check(rc).map(|_| ())
    }
    fn download_field(&mut self, field: i32, v: &mut [f64]) -> Result<(), Error> {
        let rc = unsafe { sys::idahip_download(self.ctx.as_raw(), field, 0, 1, v.as_mut_ptr()) };
        self.ctx.check(rc).map(|_| ())
    }

    /// One Newton iteration body on the device (newton.rs:98-110): `delta = -delta; lsolve; y += delta`; returns
    /// `||delta||_wrms(ewt)` for the convergence test. `delta` (the residual of the last `sys`) and `ee` (= y) are ctx-resident.
    pub fn newton_iter_on_device(&mut self) -> Result<f64, Error> {
        let scale = [if self.cjratio != 1.0 { 2.0 / (1.0 + self.cjratio) } else { 1.0 }]; // src/ida_ls.rs:406-410
        let mut delnrm = [0.0f64];
        let idx = [0i32];
        let rc = unsafe { sys::idahip_newton_iter(self.ctx.as_raw(), scale.as_ptr(), delnrm.as_mut_ptr(), idx.as_ptr(), 1) };
        self.ctx.check(rc)?;
        Ok(delnrm[0])
    }

    /// idaNlsConvTest's scalar part (src/ida_nls.rs:243-262) for iteration `m` with the norm already known.
    pub fn conv_test(&mut self, m: usize, delnrm: f64) -> Result<bool, nonlinear::Error> {
        if m == 0 {
            self.oldnrm = delnrm;
            if delnrm <= 0.0001 * self.toldel {
                return Ok(true);
            }
        } else {
            let rate = (delnrm / self.oldnrm).powf(1.0 / m as f64);
            if rate > RATEMAX {
                return Err(nonlinear::Error::ConvergenceRecover {});
            }
            self.ss = rate / (1.0 - rate);
        }
        Ok(self.ss * delnrm <= self.eps_newt)
    }
}

impl<D: DimName> Drop for HipNlsProblem<D> {
    fn drop(&mut self) {
        let raw = self.ctx.as_raw();
        unsafe {
            sys::idahip_dev_free(raw, self.d_x as *mut c_void);
            sys::idahip_dev_free(raw, self.d_w as *mut c_void);
        }
    }
}

fn fatal(e: Error) -> nonlinear::Error {
    // the trait's error type has no variant for a device failure: IllegalInput is its unrecoverable error
    eprintln!("ida-hip: {}", e);
    nonlinear::Error::IllegalInput {}
}

impl<D> NLProblem<f64, D> for HipNlsProblem<D>
where
    D: DimName,
    DefaultAllocator: Allocator<f64, D>,
{
    /// idaNlsResidual (src/ida_nls.rs:118-153): `y` is the accumulated correction ycor; f = F(tn, yypredict + ycor, yppredict + cj ycor).
    fn sys<SB1, SB2>(&mut self, y: &Matrix<f64, D, U1, SB1>, f: &mut Matrix<f64, D, U1, SB2>) -> Result<(), nonlinear::Error>
    where
        SB1: Storage<f64, D, U1>,
        SB2: StorageMut<f64, D, U1>,
    {
        let ycor: Vec<f64> = y.iter().copied().collect();
        self.upload_field(sys::IDAHIP_F_EE, &ycor).map_err(fatal)?;
        let (tn, cj, idx) = ([self.tn], [self.cj], [0i32]);
        let rc = unsafe { sys::idahip_nls_sys(self.ctx.as_raw(), tn.as_ptr(), cj.as_ptr(), 0, idx.as_ptr(), 1) };
        self.ctx.check(rc).map_err(fatal)?;
        self.nre += 1;
        let mut res = vec![0.0f64; D::dim()];
        self.download_field(sys::IDAHIP_F_DELTA, &mut res).map_err(fatal)?;
        for (dst, src) in f.iter_mut().zip(res.iter()) {
            *dst = *src;
        }
        Ok(())
    }

    /// idaNlsLSetup + idaLsSetup (src/ida_nls.rs:156-187, src/ida_ls.rs:232-290): Jacobian at the current yy, yp, cj and its LU.
    fn setup<SA, SB>(&mut self, _y: &Matrix<f64, D, U1, SA>, _f: &Matrix<f64, D, U1, SB>, _jbad: bool) -> Result<bool, nonlinear::Error>
    where
        SA: Storage<f64, D, U1>,
        SB: Storage<f64, D, U1>,
    {
        let (tn, cj, idx) = ([self.tn], [self.cj], [0i32]);
        let mut info = [0i32];
        let rc = unsafe { sys::idahip_nls_lsetup(self.ctx.as_raw(), tn.as_ptr(), cj.as_ptr(), info.as_mut_ptr(), idx.as_ptr(), 1) };
        self.ctx.check(rc).map_err(fatal)?;
        self.nsetups += 1;
        if info[0] != 0 {
            return Err(nonlinear::Error::LinearSetupFailed { source: linear::Error::LUFactFail { col: info[0] as usize } });
        }
        self.cjold = self.cj; // src/ida_nls.rs:177-179
        self.cjratio = 1.0;
        self.ss = 20.0;
        Ok(true)
    }

    /// idaNlsLSolve + idaLsSolve (src/ida_nls.rs:190-215, src/ida_ls.rs:298-455): b <- J^-1 b, scaled by 2 / (1 + cjratio).
    /// Runs `idahip_newton_iter` on -b (its negation gives b back); the device's `ee` is overwritten by the next `sys`.
    fn solve<SA, SB>(&mut self, _y: &Matrix<f64, D, U1, SA>, b: &mut Matrix<f64, D, U1, SB>) -> Result<(), nonlinear::Error>
    where
        SA: Storage<f64, D, U1>,
        SB: StorageMut<f64, D, U1>,
    {
        let neg: Vec<f64> = b.iter().map(|v| -*v).collect();
        self.upload_field(sys::IDAHIP_F_DELTA, &neg).map_err(fatal)?;
        self.newton_iter_on_device().map_err(fatal)?;
        let mut x = vec![0.0f64; D::dim()];
        self.download_field(sys::IDAHIP_F_DELTA, &mut x).map_err(fatal)?;
        for (dst, src) in b.iter_mut().zip(x.iter()) {
            *dst = *src;
        }
        Ok(())
    }

    /// idaNlsConvTest (src/ida_nls.rs:218-266): delnrm = ||del||_wrms(ewt) on the device (sequential sum, norm_rms.rs:31-38).
    fn ctest<NLS, SA, SB, SC>(
        &mut self,
        solver: &NLS,
        _y: &Matrix<f64, D, U1, SA>,
        del: &Matrix<f64, D, U1, SB>,
        _tol: f64,
        ewt: &Matrix<f64, D, U1, SC>,
    ) -> Result<bool, nonlinear::Error>
    where
        NLS: NLSolver<f64, D>,
        SA: Storage<f64, D, U1>,
        SB: Storage<f64, D, U1>,
        SC: Storage<f64, D, U1>,
    {
        let (d, w): (Vec<f64>, Vec<f64>) = (del.iter().copied().collect(), ewt.iter().copied().collect());
        self.ctx.h2d(self.d_x, &d).map_err(fatal)?;
        self.ctx.h2d(self.d_w, &w).map_err(fatal)?;
        let mut nrm = [0.0f64];
        let idx = [0i32];
        let rc = unsafe { sys::idahip_wrms(self.ctx.as_raw(), self.d_x, self.d_w, nrm.as_mut_ptr(), idx.as_ptr(), 1) };
        self.ctx.check(rc).map_err(fatal)?;
        self.conv_test(solver.get_cur_iter(), nrm[0])
    }
}

/// `nonlinear::NLSolver` (crates/nonlinear/src/traits.rs:129-209): the Newton iteration of crates/nonlinear/src/newton.rs:51-167
/// for any `NLProblem`, and [`HipNewton::solve_on_device`] for a [`HipNlsProblem`]. One deviation from newton.rs, stated in
/// SURVEY.md (quirk Q3): on `ConvergenceRecover` with a current Jacobian the loop ends with that error, as in C IDA (the match
/// arm at newton.rs:146-153 falls through and would repeat the solve forever).
pub struct HipNewton<D>
where
    D: DimName,
    DefaultAllocator: Allocator<f64, D>,
{
    delta: OVector<f64, D>,
    jcur: bool,
    curiter: usize,
    maxiters: usize,
    niters: usize,
    nconvfails: usize,
}

impl<D> HipNewton<D>
where
    D: DimName,
    DefaultAllocator: Allocator<f64, D>,
{
    /// `Newton::solve` with every iteration one device call; `y0 = 0` (the correction starts from zero, newton.rs:73-93 as
    /// `Ida::nonlinear_solve` calls it, src/lib.rs:826-840). On success the accumulated correction is the ctx's `ee`.
    pub fn solve_on_device(&mut self, problem: &mut HipNlsProblem<D>, call_lsetup: bool) -> Result<(), nonlinear::Error> {
        let mut call_lsetup = call_lsetup;
        let zero = OVector::<f64, D>::zeros();
        loop {
            let mut f = OVector::<f64, D>::zeros();
            problem.sys(&zero, &mut f)?; // ee = 0; delta = F(yypredict, yppredict)
            if call_lsetup {
                match problem.setup(&zero, &f, false) {
                    Ok(j) => self.jcur = j,
                    Err(e) => {
                        self.nconvfails += 1;
                        return Err(e);
                    }
                }
            }
            self.curiter = 0;
            let outcome = loop {
                self.niters += 1;
                let delnrm = problem.newton_iter_on_device().map_err(fatal)?;
                match problem.conv_test(self.curiter, delnrm) {
                    Ok(true) => {
                        self.jcur = false;
                        break Ok(());
                    }
                    Ok(false) => {
                        self.curiter += 1;
                        if self.curiter >= self.maxiters {
                            break Err(nonlinear::Error::ConvergenceRecover {});
                        }
                        // NLProblem::sys at the current iterate: the ctx's `ee` already holds y (reset_ee = 0)
                        let (tn, cj, idx) = ([problem.tn], [problem.cj], [0i32]);
                        let rc = unsafe { sys::idahip_nls_sys(problem.ctx.as_raw(), tn.as_ptr(), cj.as_ptr(), 0, idx.as_ptr(), 1) };
                        problem.ctx.check(rc).map_err(fatal)?;
                        problem.nre += 1;
                    }
                    Err(e) => break Err(e),
                }
            };
            match outcome {
                Ok(()) => return Ok(()),
                Err(nonlinear::Error::ConvergenceRecover {}) if !self.jcur => {
                    self.nconvfails += 1;
                    call_lsetup = true;
                }
                Err(e) => {
                    self.nconvfails += 1;
                    return Err(e);
                }
            }
        }
    }
}

impl<D> NLSolver<f64, D> for HipNewton<D>
where
    D: DimName,
    DefaultAllocator: Allocator<f64, D>,
{
    fn new(maxiters: usize) -> Self {
        HipNewton { delta: OVector::zeros(), jcur: false, curiter: 0, maxiters, niters: 0, nconvfails: 0 }
    }

    fn solve<NLP, SA, SB, SC>(
        &mut self,
        problem: &mut NLP,
        y0: &Matrix<f64, D, U1, SA>,
        y: &mut Matrix<f64, D, U1, SB>,
        w: &Matrix<f64, D, U1, SC>,
        tol: f64,
        call_lsetup: bool,
    ) -> Result<(), nonlinear::Error>
    where
        NLP: NLProblem<f64, D>,
        SA: Storage<f64, D, U1>,
        SB: StorageMut<f64, D, U1>,
        SC: Storage<f64, D, U1>,
    {
        let mut jbad = false;
        let mut call_lsetup = call_lsetup;
        loop {
            // residual at y0, linear setup if asked for (newton.rs:73-96)
            let mut delta = std::mem::replace(&mut self.delta, OVector::zeros());
            let mut attempt = problem.sys(y0, &mut delta);
            if attempt.is_ok() && call_lsetup {
                attempt = problem.setup(y0, &delta, jbad).map(|jcur| self.jcur = jcur);
            }
            if attempt.is_ok() {
                self.curiter = 0;
                y.copy_from(y0);
                attempt = loop {
                    self.niters += 1;
                    delta.neg_mut();
                    if let Err(e) = problem.solve(y, &mut delta) {
                        break Err(e);
                    }
                    *y += &delta;
                    self.delta = delta; // ctest reads the solver (get_cur_iter) and the update
                    let verdict = problem.ctest(self, y, &self.delta, tol, w);
                    delta = std::mem::replace(&mut self.delta, OVector::zeros());
                    match verdict {
                        Ok(true) => {
                            self.jcur = false;
                            break Ok(());
                        }
                        Ok(false) => {
                            self.curiter += 1;
                            if self.curiter >= self.maxiters {
                                break Err(nonlinear::Error::ConvergenceRecover {});
                            }
                            if let Err(e) = problem.sys(y, &mut delta) {
                                break Err(e);
                            }
                        }
                        Err(e) => break Err(e),
                    }
                };
            }
            self.delta = delta;
            match attempt {
                Ok(()) => return Ok(()),
                Err(nonlinear::Error::ConvergenceRecover {}) if !self.jcur => {
                    self.nconvfails += 1;
                    call_lsetup = true;
                    jbad = true;
                }
                Err(e) => {
                    self.nconvfails += 1;
                    return Err(e);
                }
            }
        }
    }

    fn get_num_iters(&self) -> usize {
        self.niters
    }
    fn get_cur_iter(&self) -> usize {
        self.curiter
    }
    fn get_num_conv_fails(&self) -> usize {
        self.nconvfails
    }
}
