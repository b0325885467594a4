//! `ida::traits::IdaProblem` (src/traits.rs:12-94: `Residual + Jacobian + Root`) as a [`HostProblem`]: the reference's own
//! problem structs (e.g. src/sample_problems/roberts.rs) run unchanged behind the C ABI's host callbacks
//! (`IDAHIP_HOST_CALLBACK`), one system at a time; LU, triangular solves, norms and the stepper's vectors stay on the device.
//! Feature `ida-problem` (needs the reference's root crate, which upstream ships without a `[package]` section).
//!
//! Experimental, never compiled (see the crate documentation).
use ida::traits::{Jacobian, Residual};
use ndarray::{ArrayView1, ArrayViewMut1, ArrayViewMut2, ShapeBuilder};

use crate::HostProblem;

/// Wraps one problem instance shared by every system of the batch (the reference integrates one `Ida` per problem; an
/// ensemble of different parameter sets uses one adapter per system through `Vec<P>` below).
pub struct IdaProblemAdapter<P> {
    problems: Vec<P>,
}

impl<P> IdaProblemAdapter<P>
where
    P: Residual<Scalar = f64> + Jacobian<Scalar = f64>,
{
    /// The same problem for every system.
    pub fn shared(problem: P) -> Self {
        IdaProblemAdapter { problems: vec![problem] }
    }
    /// `problems[sys]` for system `sys` (a parameter sweep).
    pub fn per_system(problems: Vec<P>) -> Self {
        assert!(!problems.is_empty());
        IdaProblemAdapter { problems }
    }
    fn of(&self, sys: usize) -> &P {
        &self.problems[if self.problems.len() == 1 { 0 } else { sys }]
    }
}

impl<P> HostProblem for IdaProblemAdapter<P>
where
    P: Residual<Scalar = f64> + Jacobian<Scalar = f64>,
{
    /// `Residual::res(tt, yy, yp, rr)` (src/traits.rs:28-37)
    fn res(&self, sys: usize, tt: f64, yy: &[f64], yp: &[f64], resval: &mut [f64]) {
        self.of(sys).res(tt, ArrayView1::from(yy), ArrayView1::from(yp), ArrayViewMut1::from(resval));
    }

    /// `Jacobian::jac(tt, cj, yy, yp, rr, jac)` (src/traits.rs:58-69). The reference indexes `jac[[row, col]]`
    /// (roberts.rs:80-90); the library's matrix is column-major, i.e. Fortran order of an n x n view.
    fn jac(&self, sys: usize, tt: f64, cj: f64, yy: &[f64], yp: &[f64], resvec: &[f64], jac_colmajor: &mut [f64]) {
        let n = yy.len();
        let j = ArrayViewMut2::from_shape((n, n).f(), jac_colmajor).expect("n x n Jacobian");
        self.of(sys).jac(tt, cj, ArrayView1::from(yy), ArrayView1::from(yp), ArrayView1::from(resvec), j);
    }
}
