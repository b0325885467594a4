"""Synthetic problem generators for the configurations of BASELINE.json / SURVEY.md 8(d).

Pure input generation (numpy): matrices, parameters, consistent initial conditions, tolerances, output grids. The
reference ships none of these except Roberts (examples/roberts.rs:64-70) and the Lorenz parameters
(tests/lorenz63.rs:17-25); everything else is this build's choice, fixed here so CPU oracle and GPU path integrate
identical inputs.
"""
import numpy as np


def roberts():
    """Config 1 -- examples/roberts.rs:64-70, 95-136."""
    return {
        "kind": "roberts", "n": 3, "yy0": np.array([[1.0, 0.0, 0.0]]), "yp0": np.array([[-0.04, 0.04, 0.0]]),
        "rtol": 1.0e-4, "atol": np.array([1.0e-8, 1.0e-6, 1.0e-6]), "touts": 0.4 * 10.0 ** np.arange(12),
    }


def lorenz63(batch=1024, seed=63):
    """Config 2 -- Lorenz63 as an index-0 DAE. p, r, b from tests/lorenz63.rs:17-25; ICs/tolerances/horizon are ours:
    y(0) = [1,1,1] + 1e-3*u_b, u_b ~ U(-1,1)^3 (PCG64 seed 63), y'(0) = f(y(0)), rtol 1e-6, atol 1e-9, t = 0.1..5.0."""
    rng = np.random.Generator(np.random.PCG64(seed))
    u = rng.uniform(-1.0, 1.0, size=(batch, 3))
    y0 = 1.0 + 1.0e-3 * u
    p, r, b = 10.0, 28.0, 8.0 / 3.0
    yp0 = np.stack([p * (y0[:, 1] - y0[:, 0]), y0[:, 0] * (r - y0[:, 2]) - y0[:, 1], y0[:, 0] * y0[:, 1] - b * y0[:, 2]], axis=1)
    params = np.tile(np.array([p, r, b]), (batch, 1))
    return {"kind": "lorenz63", "n": 3, "yy0": y0, "yp0": yp0, "params": params, "rtol": 1.0e-6, "atol": np.array([1.0e-9]),
            "touts": 0.1 * np.arange(1, 51)}


def _linear_system(n, b):
    """One system of config 3: returns (A_colmajor, B_colmajor, c, y0, yp0)."""
    rng = np.random.Generator(np.random.PCG64(n * 1000003 + b))
    R = rng.uniform(-1.0, 1.0, size=(n, n))
    G = rng.standard_normal(size=(n, n))
    s = rng.uniform(0.0, 1.0)
    c = rng.uniform(-1.0, 1.0, size=n)
    nd = (3 * n) // 4  # differential unknowns first, algebraic last
    R[nd:, :] = 0.0
    R[:, nd:] = 0.0
    A = 0.01 * R
    A[np.arange(nd), np.arange(nd)] += 1.0
    Bm = -((0.5 / np.sqrt(n)) * G)
    Bm[np.arange(n), np.arange(n)] -= (1.0 + s)
    # consistent initial conditions (the reference has no IDACalcIC: src/lib.rs:328-335)
    y0 = np.zeros(n)
    yp0 = np.zeros(n)
    if nd < n:
        y0[nd:] = np.linalg.solve(Bm[nd:, nd:], c[nd:])
    if nd > 0:
        yp0[:nd] = np.linalg.solve(A[:nd, :nd], c[:nd] - Bm[:nd, nd:] @ y0[nd:])
    return np.ascontiguousarray(A.T), np.ascontiguousarray(Bm.T), c, y0, yp0


_SHARED = {}


def _fill_range(args):
    """Worker: fill systems [lo, hi) of the five output arrays (process-local arrays, or shared memory blocks by name)."""
    n, first, lo, hi, limit_blas, shm_names, shapes = args
    handles = []
    if shm_names is None:
        A, Bm, c, y0, yp0 = _SHARED["arrays"]
    else:
        from multiprocessing import shared_memory
        handles = [shared_memory.SharedMemory(name=nm) for nm in shm_names]
        A, Bm, c, y0, yp0 = [np.ndarray(sh, dtype=np.float64, buffer=h.buf) for sh, h in zip(shapes, handles)]

    def fill():
        for s in range(lo, hi):
            A[s], Bm[s], c[s], y0[s], yp0[s] = _linear_system(n, first + s)

    if limit_blas:  # one BLAS thread per worker process
        try:
            from threadpoolctl import threadpool_limits
            with threadpool_limits(limits=1):
                fill()
        except ImportError:
            fill()
    else:
        fill()
    del A, Bm, c, y0, yp0
    for h in handles:
        h.close()
    return hi - lo




def ensure_fork_server():
    """Start multiprocessing's fork server now (idempotent). The server is started by fork + exec of a fresh interpreter; call
    this BEFORE the process creates its first device context, so that the exec never comes from a process that holds an
    initialised ROCm runtime (the generator pools of linear_dense / linear_dense_slices then fork from the clean server)."""
    import multiprocessing as mp
    from multiprocessing import forkserver
    ctx = mp.get_context("forkserver")
    ctx.set_forkserver_preload(["numpy"])
    forkserver.ensure_running()
    return ctx


def linear_dense(n=512, batch=4096, first=0, procs=1, nthreads=1):
    """Config 3/5 -- synthetic random linear dense index-1 DAE F = A y' + B y - c, systems [first, first+batch).
    Matrices are returned column-major per system (array[s, j, i] = M_s(i, j)).
    `procs` > 1 generates with worker processes of a fork server into named shared memory: safe also after this process
    has initialised the GPU (nothing is forked from it). `nthreads` > 1 generates with a thread pool instead (numpy's
    generators and LAPACK calls release the GIL only in part: about 1.5x on 8 cores)."""
    shapes = [(batch, n, n), (batch, n, n), (batch, n), (batch, n), (batch, n)]
    if nthreads > 1 and procs <= 1 and batch >= 2 * nthreads:
        from concurrent.futures import ThreadPoolExecutor
        arrays = [np.empty(sh) for sh in shapes]
        A, Bm, c, y0, yp0 = arrays

        def fill(lo, hi):
            for s in range(lo, hi):
                A[s], Bm[s], c[s], y0[s], yp0[s] = _linear_system(n, first + s)
            return hi - lo

        step = max(1, batch // (nthreads * 8))
        try:
            from threadpoolctl import threadpool_limits
            limiter = threadpool_limits(limits=1)  # one BLAS thread per pool thread
        except ImportError:
            limiter = None
        try:
            with ThreadPoolExecutor(nthreads) as pool:
                done = sum(pool.map(lambda lo: fill(lo, min(batch, lo + step)), range(0, batch, step)))
        finally:
            if limiter is not None:
                limiter.restore_original_limits()
        assert done == batch
        return {"kind": "linear_dense", "n": n, "A": A, "B": Bm, "c": c, "yy0": y0, "yp0": yp0, "rtol": 1.0e-6,
                "atol": np.array([1.0e-8]), "touts": 0.1 * np.arange(1, 11)}
    if procs > 1 and batch >= 2 * procs:
        # worker processes forked from a clean fork server (itself started by exec, so none of them inherits this
        # process's threads, locks or -- if the GPU is already in use here -- ROCm runtime); results come back through
        # named shared memory, a slice of the batch at a time: a slice is copied into ordinary memory and its blocks are
        # unlinked before the next one is created, so that at most ~2 GB per process sit in /dev/shm however large the
        # batch is (eight ranks of a node generate at the same time)
        import multiprocessing as mp
        from multiprocessing import shared_memory
        arrays = [np.empty(sh) for sh in shapes]
        slice_systems = max(2 * procs, min(batch, (1 << 31) // max(1, 16 * n * n)))  # A and B of a slice: <= 2 GiB
        ctx = mp.get_context("forkserver")
        ctx.set_forkserver_preload(["numpy"])
        with ctx.Pool(procs) as pool:
            for s0 in range(0, batch, slice_systems):
                cnt = min(slice_systems, batch - s0)
                sl_shapes = [(cnt,) + sh[1:] for sh in shapes]
                blocks = [shared_memory.SharedMemory(create=True, size=max(8, int(np.prod(sh)) * 8)) for sh in sl_shapes]
                try:
                    step = max(1, cnt // (procs * 4))
                    names = [b.name for b in blocks]
                    jobs = [(n, first + s0, lo, min(cnt, lo + step), True, names, sl_shapes) for lo in range(0, cnt, step)]
                    assert sum(pool.map(_fill_range, jobs)) == cnt
                    for dst, sh, b in zip(arrays, sl_shapes, blocks):
                        dst[s0:s0 + cnt] = np.ndarray(sh, dtype=np.float64, buffer=b.buf)
                finally:
                    for b in blocks:
                        b.close()
                        b.unlink()
    else:
        arrays = [np.empty(sh) for sh in shapes]
        _SHARED["arrays"] = arrays
        _fill_range((n, first, 0, batch, False, None, shapes))
        _SHARED.pop("arrays")
    A, Bm, c, y0, yp0 = arrays
    return {"kind": "linear_dense", "n": n, "A": A, "B": Bm, "c": c, "yy0": y0, "yp0": yp0, "rtol": 1.0e-6,
            "atol": np.array([1.0e-8]), "touts": 0.1 * np.arange(1, 11)}


def linear_dense_slices(n=512, batch=4096, first=0, procs=1, slice_bytes=1 << 31):
    """Config 3/5 generated a slice at a time: yields (s0, A, B, c, yy0, yp0) for consecutive slices of at most `slice_bytes`
    of matrices (A and B together), systems [first + s0, first + s0 + len) -- the same systems, bit for bit, as linear_dense.
    The consumer (bench.py: upload to the device) sees one slice at a time, so a process never holds more than a slice of its
    shard on the host: eight ranks of a node then need 8 x ~4 GB instead of 8 x 17 GB. The arrays of a slice are only valid
    until the next one is requested."""
    per = max(1, slice_bytes // max(1, 16 * n * n))
    if procs > 1 and batch >= 2 * procs:
        import multiprocessing as mp
        from multiprocessing import shared_memory
        per = max(2 * procs, min(batch, per))
        ctx = mp.get_context("forkserver")
        ctx.set_forkserver_preload(["numpy"])
        with ctx.Pool(procs) as pool:
            for s0 in range(0, batch, per):
                cnt = min(per, batch - s0)
                shapes = [(cnt, n, n), (cnt, n, n), (cnt, n), (cnt, n), (cnt, n)]
                blocks = [shared_memory.SharedMemory(create=True, size=max(8, int(np.prod(sh)) * 8)) for sh in shapes]
                try:
                    step = max(1, cnt // (procs * 4))
                    names = [b.name for b in blocks]
                    jobs = [(n, first + s0, lo, min(cnt, lo + step), True, names, shapes) for lo in range(0, cnt, step)]
                    assert sum(pool.map(_fill_range, jobs)) == cnt
                    views = [np.ndarray(sh, dtype=np.float64, buffer=b.buf) for sh, b in zip(shapes, blocks)]
                    yield (s0,) + tuple(views)
                    del views
                finally:
                    for b in blocks:
                        b.close()
                        b.unlink()
    else:
        for s0 in range(0, batch, per):
            cnt = min(per, batch - s0)
            shapes = [(cnt, n, n), (cnt, n, n), (cnt, n), (cnt, n), (cnt, n)]
            arrays = [np.empty(sh) for sh in shapes]
            _SHARED["arrays"] = arrays
            _fill_range((n, first + s0, 0, cnt, False, None, shapes))
            _SHARED.pop("arrays")
            yield (s0,) + tuple(arrays)


def make_ctxs_linear_dense_streamed(n, first, count, parts, procs=1, device=0, keep=0, slice_bytes=1 << 31):
    """Device contexts for the config-3 systems [first, first + count), generated ONCE, slice by slice (linear_dense_slices), and
    uploaded to every part that covers them. parts: [(offset, cnt, stream)] -- the context of part i holds systems
    [first + offset, first + offset + cnt) on HIP stream `stream` (None: its own); parts may overlap (bench.py: the groups of a
    rank, and one context with all of the rank's systems for the kernel timers). Returns [(ctx, prob)]: prob has everything
    linear_dense returns for its systems except that "A" and "B" only hold the first `keep` systems of the whole range and only
    in parts that start at offset 0 (the calibration sample / the CPU baseline's sample)."""
    from . import Ctx
    if procs > 1 and count >= 2 * procs:
        ensure_fork_server()  # before the device contexts exist: the server's exec must not come from a GPU process
    rtol, atol = 1.0e-6, np.array([1.0e-8])
    ctxs = []
    for off, cnt, stream in parts:
        assert 0 <= off and off + cnt <= count and cnt >= 1
        c_ = Ctx("linear_dense", n, cnt, device=device, stream=stream)
        c_.set_tolerances(rtol, atol)
        ctxs.append(c_)
    c, y0, yp0 = np.empty((count, n)), np.empty((count, n)), np.empty((count, n))
    keep = min(keep, count)
    Ak, Bk = np.empty((keep, n, n)), np.empty((keep, n, n))
    up = max(1, (1 << 28) // (8 * n * n))  # <= 256 MiB per matrix upload
    for s0, A, Bm, cs, ys, yps in linear_dense_slices(n, count, first, procs, slice_bytes):
        cnt_s = A.shape[0]
        for c_, (off, cnt, _) in zip(ctxs, parts):
            lo, hi = max(s0, off), min(s0 + cnt_s, off + cnt)  # the part's systems inside this slice
            for f in range(lo, hi, up):
                g = min(hi, f + up)
                c_.set_linear_dense(A[f - s0:g - s0], Bm[f - s0:g - s0], cs[f - s0:g - s0], first=f - off)
        c[s0:s0 + cnt_s], y0[s0:s0 + cnt_s], yp0[s0:s0 + cnt_s] = cs, ys, yps
        if s0 < keep:
            m = min(cnt_s, keep - s0)
            Ak[s0:s0 + m], Bk[s0:s0 + m] = A[:m], Bm[:m]
        del A, Bm, cs, ys, yps
    out = []
    for c_, (off, cnt, _) in zip(ctxs, parts):
        k_ = min(keep, cnt) if off == 0 else 0
        out.append((c_, {"kind": "linear_dense", "n": n, "A": Ak[:k_], "B": Bk[:k_], "c": c[off:off + cnt], "yy0": y0[off:off + cnt],
                         "yp0": yp0[off:off + cnt], "rtol": rtol, "atol": atol, "touts": 0.1 * np.arange(1, 11), "matrices_on_host": k_}))
    return out


def make_ctx_linear_dense_streamed(n, batch, first=0, procs=1, device=0, stream=None, keep=0, slice_bytes=1 << 31):
    """One device context of `batch` config-3 systems [first, first + batch) whose matrices go from the generator to the device a
    slice at a time. Returns (ctx, prob): see make_ctxs_linear_dense_streamed."""
    return make_ctxs_linear_dense_streamed(n, first, batch, [(0, batch, stream)], procs, device, keep, slice_bytes)[0]


def heat1d(n=4096, batch=256):
    """Config 4 -- 1-D heat equation u_t = kappa u_xx, method of lines on n nodes, Dirichlet ends as algebraic equations;
    kappa_b = 1 + b/256; y_i(0) = sin(pi x_i) with exact zeros at both ends; y'(0) = the interior right-hand side."""
    dx = 1.0 / (n - 1)
    x = np.arange(n) * dx
    y = np.sin(np.pi * x)
    y[0] = 0.0
    y[-1] = 0.0
    kappa = 1.0 + np.arange(batch) / 256.0
    coef = kappa / (dx * dx)
    y0 = np.tile(y, (batch, 1))
    yp0 = np.zeros((batch, n))
    yp0[:, 1:-1] = coef[:, None] * ((y[None, :-2] - 2.0 * y[None, 1:-1]) + y[None, 2:])
    return {"kind": "heat1d", "n": n, "yy0": y0, "yp0": yp0, "params": coef.reshape(batch, 1), "rtol": 1.0e-5,
            "atol": np.array([1.0e-8]), "touts": 0.01 * np.arange(1, 11)}


def make_ctx(prob, device=0, stream=None):
    """Create a device context for a generated problem and load its data (import kept local: this module is also
    used by CPU-only tests)."""
    from . import Ctx
    batch = prob["yy0"].shape[0]
    ctx = Ctx(prob["kind"], prob["n"], batch, device=device, stream=stream)
    ctx.set_tolerances(prob["rtol"], prob["atol"])
    if prob["kind"] == "linear_dense":
        step = max(1, (1 << 28) // (8 * prob["n"] * prob["n"]))  # <= 256 MiB per matrix upload
        for f in range(0, batch, step):
            ctx.set_linear_dense(prob["A"][f:f + step], prob["B"][f:f + step], prob["c"][f:f + step], first=f)
    elif prob["kind"] == "host_callback":
        ctx.set_host_problem(prob["res"], prob["jac"])
    elif "params" in prob:
        ctx.set_problem_params(prob["params"])
    return ctx
