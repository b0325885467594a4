#!/usr/bin/env python3
"""bench.py -- Newton iterations / second (fp64) of the batched BDF/Newton hot path on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line (rank 0). For N > 1 it is normally launched
by torch.distributed.run, one rank per GPU; started by hand without a launcher (`python bench.py --gpus 2`, no WORLD_SIZE in
the environment) it starts the N ranks itself as child processes, before anything of torch or HIP is loaded.

Workload (BASELINE.json configs[2]): synthetic random linear dense index-1 DAE
F = A y' + B y - c, N = 512, batch B = 4096 systems per GPU (distinct matrices per system, all device resident),
rtol 1e-6, atol 1e-8, integrated from t = 0 to t = 1 with outputs every 0.1 (ten Ida::solve calls per system, handed over
as one schedule: systems do not wait for each other at the outputs). Other workloads (parity configs, not the headline):
`--workload lorenz63 --n 3 --batch 1024` (config 2), `--workload heat1d --n 4096 --batch 256` (config 4).

What is timed.
  value        Throughput mode (idaens_stream): a system that has reached t = 1 is created anew from its initial conditions
               and integrates again, so the batch never drains; a "step" is one lock-step step attempt of the whole batch
               (set_coeffs -> predict -> Newton solve with residual, Jacobian + batched LU when the reference's rule asks
               for it, 1..4 triangular solves + WRMS norms -> error test -> complete_step or restore, for every one of the B
               systems). value = Newton iterations of all systems on all ranks during the K timed steps / max over ranks of
               the wall time of those steps, inputs resident in HBM, NO event timers or extra synchronisation inside.
  whole_pass   SURVEY 8(d)'s protocol next to it (N = 1 only): the whole ensemble from fresh state, t = 0 -> 1, median of 10
               passes (--passes; Newton iterations of the pass / its wall time).
  kernel_classes_rank0, roofline, lu_plus_solve
               measured in further, untimed repetitions of the K steps with HIP-event timers on the ctx stream: once per
               kernel class, once per kernel of the LU (the dominant kernel's `roofline`).
  device_controller, newton_fusion  (N = 1 only) the whole pass again with the lock-step HOST stepper (idaens_set_device_controller(0))
               and, on top of that, with one host round trip per Newton iteration: the before/after of moving the controller to the
               device. Same work, same results.
Scheduling (results do not depend on either): the rank's systems run as --groups ensembles side by side on the device and a
lock-step round may postpone its linear setups until most stepping systems ask for one (--lu-period, idahip_set_lu_period).
Multi-GPU (config 5): the ensemble shards embarrassingly -- rank r integrates systems [4096 r, 4096 (r+1)); no data-path
collective; the ranks meet over gloo on CPU tensors for the barrier and the max-over-ranks of the time (no RCCL).
"""
import argparse
import hashlib
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "rust-ida_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)
VALU_FMA_TFLOPS = 78.6       # fp64 vector peak with FMA = 1/2 x 157.3 TFLOP/s fp32 vector (same guide), 2.4 GHz
VALU_UNFUSED_TFLOPS = 39.3   # a multiply and a subtract per update (the reference's arithmetic, dense.rs:151): half of it
PMC_SUMMARY = os.path.join("profiles", "r05_bench_summary.json")  # committed rocprofv3 --pmc passes of this command


def algorithmic_bytes(n, workload="linear_dense"):
    """Per-system algorithmic HBM bytes of each kernel class (fp64 = 8 B)."""
    if workload == "heat1d":  # three-point residual, dense Jacobian written (mostly zeros)
        return {"newton_iter": 8 * n * n + 40 * n + 8, "sys": 56 * n, "jac": 8 * n * n, "sys_jac": 8 * n * n + 56 * n,
                "lu": 16 * n * n + 8 * n}
    if workload == "lorenz63":
        return {"newton_iter": 8 * n * n + 40 * n + 8, "sys": 56 * n + 24, "jac": 8 * n * n + 8 * n + 24, "sys_jac": 0, "lu": 16 * n * n + 8 * n}
    return {
        "newton_iter": 8 * n * n + 40 * n + 8,   # getrs 8N^2+24N, + neg/scale/axpy/wrms vectors 16N+8
        "sys": 16 * n * n + 40 * n,              # residual of the linear dense DAE
        "jac": 24 * n * n,                       # J = B + cj A
        "sys_jac": 24 * n * n + 40 * n,          # residual and J = B + cj A in one pass over A and B
        "lu": 16 * n * n + 8 * n,                # getrf: read + write the matrix once, pivots
    }


def getrf_flops(n):
    return 2.0 * n ** 3 / 3.0 - n ** 2 / 2.0 - n / 6.0


def trailing_work(n):
    """Algorithmic work of lu_trail64w_kernel per matrix (all its launches of one factorisation): for every 64-column
    super-panel with mrem rows and ntrail columns right of it, U12 = L11^-1 A12 (64 x 63 / 2 updates per column) and
    A22 -= L21 U12 (mrem x ntrail x 64 updates); an update = 2 flops. Bytes: A22 read + written once per super-panel, L21 and
    the pivot rows read, U12 written."""
    flops, nbytes, launches = 0.0, 0.0, 0
    for k0 in range(0, n, 64):
        m = n - k0 - 64
        if m <= 0:
            break
        launches += 1
        flops += 2.0 * (m * m * 64 + m * (64 * 63 // 2))
        nbytes += 16.0 * m * m + 8.0 * 64 * (m + 2 * m)
    return flops, nbytes, launches


def kernel_sources_sha(root=None):
    """sha256 over the device sources (rust-ida_amd/csrc/*.hpp, idahip.hip): a profile summary is only valid for the kernels it
    was taken with."""
    import glob
    import hashlib
    d = os.path.join(root or ROOT, "rust-ida_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(d, "*.hpp")) + [os.path.join(d, "idahip.hip")]):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def profiled_traffic(kernel_prefix):
    """HBM bytes per matrix of one LU kernel from the committed rocprofv3 PMC summary of this command (separate --pmc
    FETCH_SIZE / WRITE_SIZE passes, gfx950-corrected; profiles/README.md). Counters cannot be read inside this process, so
    this is the profiled figure of the same command at the commit named in the summary, or None when it is absent."""
    try:
        s = json.load(open(os.path.join(ROOT, PMC_SUMMARY)))
    except Exception:
        return None
    if s.get("kernel_sources_sha") != kernel_sources_sha():
        return None  # the summary was taken with other kernels than the ones this process runs: not this build's traffic
    tot = sum(v for k, v in s.get("lu_hbm_bytes_per_matrix", {}).items() if k.startswith(kernel_prefix))
    if not tot:
        return None
    return {"hbm_bytes_per_matrix": int(tot), "source": PMC_SUMMARY, "commit": s.get("commit"), "kernel_sources_sha": s.get("kernel_sources_sha")}


def lu_plus_solve(tim, n, arithmetic, dense=True):
    """The kernel-level figure north_star states its target on: one batched getrf + one getrs per system, algorithmic bytes
    (24 N^2 + 32 N) over the device time per system of the lu class plus the newton_iter class (whose kernel
    is the getrs with the Newton vector updates fused in)."""
    lu, ni = tim["lu"], tim["newton_iter"]
    if lu["systems"] == 0 or ni["systems"] == 0 or lu["ms"] <= 0:
        return None
    us_lu, us_ni = 1e3 * lu["ms"] / lu["systems"], 1e3 * ni["ms"] / ni["systems"]
    gbs = (24 * n * n + 32 * n) / ((us_lu + us_ni) * 1e-6) / 1e9
    out = {"arithmetic": arithmetic, "getrf_us_per_matrix": round(us_lu, 3), "getrs_us_per_system": round(us_ni, 3),
           "GB/s": round(gbs, 1), "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4)}
    if dense:
        tf = getrf_flops(n) / (us_lu * 1e-6) / 1e12
        peak = VALU_FMA_TFLOPS if arithmetic == "fma" else VALU_UNFUSED_TFLOPS
        out.update({"getrf_TFLOP/s": round(tf, 2), "getrf_frac_of_valu_peak": round(tf / peak, 4), "valu_peak_TFLOP/s": peak})
    else:
        out["note"] = ("banded Jacobian in dense storage: the reference's a_kj == 0 rule (dense.rs:148) skips the updates of zero "
                       "pivot-row entries and so does the device (whole pivot rows and column blocks at a time), so a dense flop "
                       "count does not describe this factorisation; the byte figure (matrix read + written once) describes the "
                       "factorisation, while the triangular solves leave the factors' all-zero 64 x 64 blocks out (exactly) and move far "
                       "less than 8 N^2 bytes")
    return out


# IDAHIP_BENCH_TIME_ALL=1 (tools/profile_bench.sh): the per-kernel HIP-event timers of the LU run from the first launch of
# the process (spin-up, warm-up, every repetition), so that `lu_kernels_rank0` can be checked against a rocprofv3 kernel
# trace of the same process. `value` then carries the timers' synchronisations and is marked as such.
TIME_ALL = os.environ.get("IDAHIP_BENCH_TIME_ALL") == "1"


class Runner:
    """The rank's whole batch as one ensemble on one device context and stream, in throughput mode (idaens_stream): the ten
    Ida::solve calls of the workload are one output schedule per system, a system that has reached the end is created anew
    (Ida::new from its initial conditions) and starts over at once. The batch never drains: every lock-step round works on
    every system, each somewhere else in its integration."""

    LU_PERIOD = 1   # idahip_set_lu_period of every context (set from --lu-period)
    STAGGER = None  # rounds over which the systems' first starts are spread; None: one typical integration (calibrated)
    SPIN_UP = 200   # untimed rounds before the warm-up: the stagger plus at least one more integration
    CALIBRATION_SYSTEMS = 128

    @staticmethod
    def integration_length(prob, device, nsample):
        """Median number of lock-step rounds (step attempts) an integration of this workload takes, measured on the device
        on the first `nsample` systems. The first starts are spread over exactly this many rounds: with a shorter or longer
        ramp some phases of an integration are populated twice as densely as others and the load per round (the number of
        matrices to factorise above all) oscillates with the integration's period -- +-6 % between 20-round windows with the
        96 rounds of round 1, +-2 % with the calibrated 62 (tools/round_profile.py)."""
        import idahip
        from idahip import problems
        ns = min(nsample, prob["yy0"].shape[0], prob.get("matrices_on_host", 1 << 30))
        sub = {k: (v[:ns] if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] >= ns and k not in ("atol", "touts") else v) for k, v in prob.items()}
        ctx = problems.make_ctx(sub, device=device)
        ens = idahip.Ensemble(ctx, sub["yy0"], sub["yp0"])
        status, _, reached = ens.solve_schedule(sub["touts"])
        assert (status == 0).all() and (reached == len(sub["touts"])).all()
        att = ens.counter("n_attempts")
        ens.close()
        ctx.close()
        return max(1, int(round(float(np.median(att)))))

    def __init__(self, probs, device, ctxs=None, serial=False):
        """probs / ctxs: one problem (and device context) per group. The rank's batch is integrated as len(probs) ensembles side
        by side on the one device, each on its own HIP stream and host thread (idaens_stream_group): one group's
        stretches of a round that leave the chip idle are filled by the others' launches. One group = round 4's single ensemble.
        serial: the groups take turns instead (IDAHIP_BENCH_TIME_ALL / the timer passes: a kernel's duration is then its own)."""
        import idahip
        from idahip import problems
        self.probs = probs
        self.prob = probs[0]
        self.serial = serial
        self.stagger = self.STAGGER if self.STAGGER is not None else self.integration_length(probs[0], device, self.CALIBRATION_SYSTEMS)
        self.ctxs = ctxs if ctxs is not None else [problems.make_ctx(p, device=device) for p in probs]
        if self.LU_PERIOD > 1 and self.probs[0].get("kind") not in ("lorenz63", "roberts"):
            for c in self.ctxs:
                c.set_lu_period(self.LU_PERIOD)
        self.enss = [idahip.Ensemble(c, p["yy0"], p["yp0"]) for c, p in zip(self.ctxs, probs)]
        self.steppers = [e.device_controller_active() for e in self.enss]
        if TIME_ALL:
            for c in self.ctxs:
                c.timing(2)
                c.timing_reset()
        self._stream(max(self.SPIN_UP, 3 * self.stagger), stagger=self.stagger)

    def _stream(self, k, stagger=0, serial=None):
        import idahip
        if (self.serial if serial is None else serial) or len(self.enss) == 1:
            for e in self.enss:
                e.stream(self.prob["touts"], k, stagger_rounds=stagger)
        else:
            idahip.stream_group(self.enss, self.prob["touts"], k, stagger_rounds=stagger)

    def total_iters(self):
        return sum(e.total_newton_iters() for e in self.enss)

    def steps(self, k, serial=None):
        """Exactly k lock-step rounds of every group (one step attempt of every system of the batch each), in one call per group:
        the host stepper runs its k rounds back to back, the device-resident stepper of the small problems runs them inside one
        launch; the groups run concurrently."""
        if k <= 0:
            return
        before = [e.total_rounds() for e in self.enss]
        self._stream(k, serial=serial)
        assert [e.total_rounds() for e in self.enss] == [b + k for b in before]

    def sync(self):
        for c in self.ctxs:
            c._chk(c.H.idahip_sync(c.h), "sync")

    def _timing_sum(self):
        tims = [c.timing_get() for c in self.ctxs]
        return {k: {f: sum(t[k][f] for t in tims) for f in ("ms", "launches", "systems")} for k in tims[0]}

    def timed_steps(self, k, level):
        """k more rounds with the HIP-event timers at `level` (1: per kernel class, 2: per kernel of the LU). The groups take
        turns here: with two streams' kernels sharing the chip a kernel's event-to-event time is no longer its own."""
        if not TIME_ALL:
            for c in self.ctxs:
                c.timing(level)
                c.timing_reset()
        self.steps(k, serial=True)
        self.sync()
        tim = self._timing_sum()
        if not TIME_ALL:
            for c in self.ctxs:
                c.timing(0)
        return tim

    def whole_pass(self, variant=4, level=0, fused=1, device_ctl=1):
        """the whole ensemble from fresh state (Ida::new for every system) through its output schedule; the groups
        side by side as in the stream (idaens_solve_schedule_group)."""
        import idahip
        enss = []
        for c, p in zip(self.ctxs, self.probs):
            c.set_lu_variant(variant)
            e = idahip.Ensemble(c, p["yy0"], p["yp0"])
            e.set_fused_newton(fused)
            e.set_device_controller(device_ctl)
            c.timing(level)
            c.timing_reset()
            enss.append(e)
        self.sync()
        t0 = time.perf_counter()
        if self.serial or len(enss) == 1:
            res = [e.solve_schedule(self.prob["touts"]) for e in enss]
        else:
            res = idahip.solve_schedule_group(enss, self.prob["touts"])
        self.sync()
        dt = time.perf_counter() - t0
        for status, _, reached in res:
            assert (status == 0).all() and (reached == len(self.prob["touts"])).all()
        tim = self._timing_sum()
        cs = [e.counters() for e in enss]
        cat = lambda k: np.concatenate([c[k] for c in cs])
        out = {"seconds": dt, "iters": sum(e.total_newton_iters() for e in enss), "rounds": max(e.total_rounds() for e in enss), "tim": tim,
               "counts": np.stack([cat(k) for k in ("nst", "netf", "ncfn", "nni", "nsetups", "kused")]),
               "yy": np.concatenate([e.yy() for e in enss]), "yp": np.concatenate([e.yp() for e in enss]),
               "stepper": enss[0].device_controller_active(),
               "paths": {k: int(cat(k).sum()) for k in ("ncfn", "nls_nconvfails", "nlufail", "nconv_jcur", "nfail_first", "nge")}}
        for e, c in zip(enss, self.ctxs):
            e.close()
            c.timing(0)
            c.set_lu_variant(4)
        return out


def device_record(torch, local_rank):
    """Which device a rank really ran on (the N > 1 line carries one per rank: two ranks on one card would show here)."""
    pr = torch.cuda.get_device_properties(local_rank)
    rec = {"local_rank": int(local_rank), "name": pr.name, "total_memory_GiB": round(pr.total_memory / 2 ** 30, 1)}
    for k in ("uuid", "pci_bus_id", "pci_device_id", "gcnArchName"):
        v = getattr(pr, k, None)
        if v is not None:
            rec[k] = str(v)
    return rec


def cpu_baseline(prob_small, cores):
    """The reference's CPU path as restated in oracle/ (kind = "port"), timed on this host's cores on a bounded sample of
    the same workload: the first len(sample) systems, integrated over the whole horizon like the GPU run."""
    import oracle_lib as O
    n = prob_small["n"]
    iters, seconds, reps = 0, 0.0, 0
    while reps == 0 or (seconds < 5.0 and reps < 400):  # small systems: an integration of the sample is milliseconds; repeat it
        r = O.run_ensemble(prob_small["kind"], n, prob_small["yy0"], prob_small["yp0"], prob_small["rtol"], prob_small["atol"],
                           prob_small["touts"], params=prob_small.get("params"), A=prob_small.get("A"), B=prob_small.get("B"),
                           c=prob_small.get("c"), nthreads=cores)
        iters += int(r["counters"]["nni"].sum())
        seconds += r["seconds"]
        reps += 1
    return {"value": iters / seconds, "unit": "Newton iters/s", "cores": cores, "kind": "port",
            "sample": "%d systems of the same N=%d workload, full t=0..%g integration%s, %d Newton iterations in %.2f s, one std::thread per core"
                      % (prob_small["yy0"].shape[0], n, float(prob_small["touts"][-1]), (" repeated %d times" % reps) if reps > 1 else "",
                         iters, seconds)}


def inputs_only(args, rank, world, first, count, procs):
    """`--inputs-only`: what an N-rank start costs the HOST before the first barrier -- every rank generates its shard exactly as
    the bench does (slice by slice; the upload is left out), rank 0 collects every rank's record and prints one JSON line. No GPU, no torch."""
    import resource
    from idahip import problems
    t0 = time.time()
    nsl, checksum = 0, 0.0
    if args.workload == "linear_dense":
        for s0, A, Bm, cs, ys, yps in problems.linear_dense_slices(args.n, count, first, procs):
            nsl += 1
            checksum += float(A[0, 0, 0]) + float(Bm[-1, -1, -1])
            del A, Bm, cs, ys, yps
    t_gen = time.time() - t0
    rss = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1048576.0  # GiB (ru_maxrss is in KiB)
    rec = (rank, round(t_gen, 2), round(rss, 2), nsl)
    allr = [rec]
    if world > 1:
        # The ranks meet through files, not through torch.distributed: importing torch opens the GPU, and this mode must be
        # able to run eight ranks on a one-GPU box (at most six processes may hold the card there). The gloo meeting itself is
        # what tests/test_sharding_gloo.py and the 2- and 4-rank rehearsals of tests/test_gpu_bench_ranks.py cover.
        import tempfile
        d = os.path.join(tempfile.gettempdir(), "idahip_inputs_only_%s" % os.environ.get("MASTER_PORT", "0"))
        os.makedirs(d, exist_ok=True)
        tmp = os.path.join(d, "rank%d.tmp" % rank)
        with open(tmp, "w") as f:
            json.dump(rec, f)
        os.replace(tmp, os.path.join(d, "rank%d.json" % rank))
        if rank == 0:
            deadline = time.time() + 1800
            while time.time() < deadline and not all(os.path.exists(os.path.join(d, "rank%d.json" % r)) for r in range(world)):
                time.sleep(0.05)
            allr = [tuple(json.load(open(os.path.join(d, "rank%d.json" % r)))) for r in range(world)]
            for r in range(world):
                os.remove(os.path.join(d, "rank%d.json" % r))
            os.rmdir(d)
    if rank == 0:
        print(json.dumps({"inputs_only": True, "workload": args.workload, "n": args.n, "batch_per_rank": args.batch, "ranks": world,
                          "generator_processes_per_rank": procs, "seconds_until_every_rank_has_its_inputs_max": max(r[1] for r in allr),
                          "peak_rss_GiB_per_rank": [r[2] for r in allr], "slices_per_rank": allr[0][3],
                          "seconds_per_rank": [r[1] for r in allr]}))
    sys.exit(0)


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (this process has not
    loaded torch or touched HIP) and leave with the worst of their exit codes. Rank 0 prints the JSON line."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rcs = [p.wait() for p in procs]
    sys.exit(max(abs(rc) for rc in rcs))


# ensembles per GPU by default: same-box A/Bs in DESIGN.md section 4b (tools/half_streams.py); the one-thread-per-system stepper
# of the n = 3 problems runs whole schedules in one launch and has nothing to interleave
DEFAULT_GROUPS = {"linear_dense": 4, "heat1d": 1, "lorenz63": 1}
# linear setups only in every k-th lock-step round (idahip_set_lu_period; systems that need one wait, results unchanged): pays where a
# batched factorisation costs about the same for 50 matrices as for 250 (config 4), costs where its time is proportional to the batch
DEFAULT_LU_PERIOD = {"linear_dense": 1, "heat1d": 5, "lorenz63": 1}


WORKLOADS = {
    "linear_dense": ("Newton iters/sec (fp64), batched dense DAE N=%d B=%d",
                     "random linear dense index-1 DAE F=A y'+B y-c (SURVEY 8(d) config 3), N=%d, B=%d systems per GPU, rtol 1e-6 "
                     "atol 1e-8, t=0..1 with 10 outputs, every system restarts on its own when it reaches t=1 (endless stream "
                     "of integrations)"),
    "heat1d": ("Newton iters/sec (fp64), 1-D heat equation (dense Jacobian) N=%d B=%d",
               "1-D heat equation by the method of lines, Dirichlet ends algebraic, dense Jacobian (SURVEY 8(d) config 4), N=%d, "
               "B=%d systems per GPU, rtol 1e-5 atol 1e-8, t=0..0.1 with 10 outputs, every system restarts on its own at the end"),
    "lorenz63": ("Newton iters/sec (fp64), Lorenz63 as index-0 DAE N=%d B=%d",
                 "Lorenz63 as an index-0 DAE (SURVEY 8(d) config 2: tests/lorenz63.rs parameters), N=%d, B=%d systems per GPU, rtol "
                 "1e-6 atol 1e-9, t=0..5 with 50 outputs, every system restarts on its own at the end; launch-latency bound"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--n", type=int, default=None)
    ap.add_argument("--batch", type=int, default=None, help="systems per GPU")
    ap.add_argument("--workload", choices=tuple(WORKLOADS), default="linear_dense",
                    help="linear_dense = config 3 (the headline, N=512 B=4096); heat1d = config 4 (N=4096 B=256); lorenz63 = config 2 (N=3 B=1024)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the whole-pass figures (N = 1 extras)")
    ap.add_argument("--groups", type=int, default=None,
                    help="ensembles per GPU, each with 1/groups of the rank's systems on its own HIP stream and host thread "
                         "(idaens_stream_group; default: %s)" % ", ".join("%s %d" % kv for kv in DEFAULT_GROUPS.items()))
    ap.add_argument("--lu-period", type=int, default=None, help="linear setups only in every k-th round of the device lock-step stepper "
                    "(default: %s)" % ", ".join("%s %d" % kv for kv in DEFAULT_LU_PERIOD.items()))
    ap.add_argument("--passes", type=int, default=10, help="whole passes timed for `whole_pass` (SURVEY 8(d): >= 10 repetitions on fresh state, median)")
    ap.add_argument("--results-npz", default=None,
                    help="rank 0 writes the concatenated per-system results of the ensemble's verification pass (nst, nni, y(tout), y'(tout) in "
                         "global system order, SURVEY 8(e): 'host concatenates') to this file")
    ap.add_argument("--inputs-only", action="store_true",
                    help="host-side rehearsal of an N-rank start: every rank generates its shard (slice by slice, as for the upload), "
                         "rank 0 collects and prints seconds and peak resident memory per rank; no GPU is touched (torch is not imported)")
    args = ap.parse_args()
    dn, db = {"linear_dense": (512, 4096), "heat1d": (4096, 256), "lorenz63": (3, 1024)}[args.workload]
    args.n = dn if args.n is None else args.n
    args.batch = db if args.batch is None else args.batch
    args.groups = DEFAULT_GROUPS[args.workload] if args.groups is None else args.groups
    args.lu_period = DEFAULT_LU_PERIOD[args.workload] if args.lu_period is None else args.lu_period
    Runner.LU_PERIOD = args.lu_period
    args.groups = max(1, min(args.groups, args.batch))

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args.gpus)  # does not return

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    # ---- inputs (numpy worker processes of a fork server; nothing of the GPU is loaded yet)
    from idahip import problems, sharding
    # the cores this process may run on (a launcher or a container may have pinned it): the N ranks of a node start their
    # generator processes at the same time, so each takes its share of those cores and at most 16
    try:
        cores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cores = os.cpu_count() or 1
    procs = int(os.environ.get("IDAHIP_GEN_PROCS", max(1, min(16, cores // max(1, world)))))  # 1 = in-process (use under rocprofv3)
    if procs > 1:
        problems.ensure_fork_server()  # before anything of the GPU is loaded in this process
    t0 = time.time()
    first, count = sharding.shard_range(rank, world, args.batch)
    ncpu = max(1, min(cores, 64))
    nsmall = min(args.batch, 16 * ncpu if args.workload == "linear_dense" else (ncpu if args.workload == "heat1d" else args.batch))
    if args.inputs_only:
        inputs_only(args, rank, world, first, count, procs)  # does not return
    G = 1 if TIME_ALL else args.groups  # the profile runs (tools/profile_bench.sh) are about the kernels: one ensemble, one stream
    args.groups = G
    gsz = [count // G + (1 if g < count % G else 0) for g in range(G)]  # the rank's systems, dealt to its groups in order
    goff = [sum(gsz[:g]) for g in range(G)]
    nsmall = min(nsmall, gsz[0])  # the CPU baseline's sample comes from the first group's host copy
    stream_ctxs = None
    # one HIP stream per group, chosen so that the device runs them side by side (the runtime may put two streams on one hardware
    # queue: idahip_concurrent_streams finds that out with a probe kernel); one group: the context's own stream
    import idahip
    dev_early = 0 if os.environ.get("IDAHIP_BENCH_REHEARSE") == "1" else local_rank
    try:
        gstreams, nconc = (idahip.concurrent_streams(G, dev_early) if G > 1 else ([None], 1))
    except idahip.IdaHipError:  # the probe could not run: ordinary streams (every context creates its own), and the line says so
        gstreams, nconc = [None] * G, 0
    # (diagnostic in the line: the least even share of the chip between two of the groups' streams, 1 = interleaved dispatch)
    try:
        share_min = None if nconc == 0 else min([idahip.stream_pair_share(gstreams[i], gstreams[j], dev_early) for i in range(G) for j in range(G) if i != j], default=None)
    except idahip.IdaHipError:  # a diagnostic must not stop the run
        share_min = None
    if args.workload == "linear_dense":
        # the shard's matrices go from the generator to the device a slice (<= 2 GiB) at a time: the process never holds the
        # 17 GB host copy of its shard (eight ranks of a node would hold 137 GB), only the calibration / CPU-baseline sample
        keep = max(Runner.CALIBRATION_SYSTEMS, nsmall if (rank == 0 and world == 1 and not args.no_cpu_baseline) else 0)
        rehearse_early = os.environ.get("IDAHIP_BENCH_REHEARSE") == "1"
        # every group's context, and (more than one group) one context with ALL of the rank's systems: the per-kernel timers
        # run on that one, at the launch sizes the roofline figures of earlier rounds are quoted for
        parts = [(goff[g], gsz[g], gstreams[g]) for g in range(G)] + ([(0, count, None)] if G > 1 else [])
        made = problems.make_ctxs_linear_dense_streamed(args.n, first, count, parts, procs=procs, device=0 if rehearse_early else local_rank, keep=keep)
        stream_ctxs, probs = [m[0] for m in made[:G]], [m[1] for m in made[:G]]
        full_ctx, full_prob = made[G] if G > 1 else (None, None)
        prob = probs[0]
    else:
        full = problems.heat1d(n=args.n, batch=args.batch * world) if args.workload == "heat1d" else problems.lorenz63(batch=args.batch * world)
        cut = lambda lo, cnt: {k: (v[lo:lo + cnt] if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == args.batch * world else v)
                               for k, v in full.items()}  # kappa_b / the initial perturbation depend on the global system id
        probs = [cut(first + goff[g], gsz[g]) for g in range(G)]
        prob = cut(first, count)
        full_ctx, full_prob = None, (prob if G > 1 else None)
    t_gen = time.time() - t0

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        small = {k: (v[:nsmall] if isinstance(v, np.ndarray) and v.ndim >= 2 and v.shape[0] >= nsmall else v) for k, v in prob.items()}
        cpu = cpu_baseline(small, ncpu)

    import torch
    import torch.distributed as dist
    # Rehearsal knob for a one-GPU box: IDAHIP_BENCH_REHEARSE=1 puts every rank on cuda:0. Never set by the driver; the JSON
    # line says so.
    rehearse = os.environ.get("IDAHIP_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        # The ranks share nothing on the data path (independent IVPs; north_star: "no RCCL required"): the process
        # group only lines them up (barrier) and combines two scalars (max time, total iterations), on CPU tensors over gloo.
        # An 8-rank run therefore does not depend on RCCL bring-up. IDAHIP_BENCH_BACKEND=nccl switches to RCCL on device tensors.
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("IDAHIP_BENCH_BACKEND", "gloo")
        if backend == "nccl" and not rehearse:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            backend = "gloo"
            dist.init_process_group("gloo", rank=rank, world_size=world)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if stream_ctxs is None:
        stream_ctxs = [problems.make_ctx(p_, device=local_rank, stream=gstreams[g]) for g, p_ in enumerate(probs)]
    run = Runner(probs, local_rank, ctxs=stream_ctxs, serial=os.environ.get("IDAHIP_BENCH_SERIAL_GROUPS") == "1")
    # the kernels' own figures (kernel_classes_rank0, lu_kernels_rank0, roofline, lu_plus_solve) are measured on ONE ensemble of all
    # of the rank's systems, alone on the device, after the timed region: a kernel's event-to-event time is only its own when
    # no other stream shares the chip, and the launch sizes are then those of a whole batch
    Runner.STAGGER = run.stagger
    krun = run if G == 1 else Runner([full_prob], local_rank, ctxs=[full_ctx] if full_ctx is not None else None)
    for p_ in probs + [prob]:  # host copies of the matrices are no longer needed
        p_.pop("A", None)
        p_.pop("B", None)

    # ---- the number: W warm-up steps, then exactly K steps between barriers, no timers inside
    run.steps(args.warmup)
    barrier()
    it0 = run.total_iters()
    t0 = time.perf_counter()
    run.steps(args.steps)
    run.sync()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    iters = run.total_iters() - it0
    barrier()
    elapsed_max, iters_all = sharding.combine(elapsed, iters, dist if world > 1 else None, device="cuda" if (world > 1 and backend == "nccl") else "cpu")
    per_rank = [iters]
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, (int(iters), round(elapsed, 6)))

    # ---- untimed repetitions of the same K steps with event timers: per kernel class, then per kernel of the LU
    tim = krun.timed_steps(args.steps, 1)
    tim2 = krun.timed_steps(args.steps, 2)
    if TIME_ALL:
        tim = tim2  # the timers were never reset: both are totals over every launch of the process

    # ---- "each shard D2H-copies y(tout), y'(tout) and counters; host concatenates". Untimed: every rank
    # integrates its shard once from fresh state through the whole output schedule (a result that does not depend on how
    # the ensemble is sharded, unlike the state of a stream after K rounds; it reuses the stream's device context, so it comes
    # after everything that continues the stream), rank 0 gathers digests of every rank's block
    # and its counter sums -- a shard with wrong inputs or a wrong device shows here -- and, on request, the arrays themselves.
    # (not in the profile runs -- IDAHIP_BENCH_TIME_ALL=1 --: the kernel trace of such a run is compared with the timers' totals,
    # and the timers are off during a whole pass)
    want_vp = not TIME_ALL
    vp = run.whole_pass(4) if want_vp else None
    vcnt = vp["counts"] if want_vp else np.zeros((6, count), dtype=np.int64)  # rows: nst, netf, ncfn, nni, nsetups, kused
    shard = {"rank": rank, "first": first, "count": count, "device": device_record(torch, local_rank),
             "input_generation_s": round(t_gen, 1), "sum_nst": int(vcnt[0].sum()), "sum_nni": int(vcnt[3].sum()), "sum_nsetups": int(vcnt[4].sum()),
             "sha256_yy": hashlib.sha256(np.ascontiguousarray(vp["yy"]).tobytes()).hexdigest() if want_vp else None,
             "sha256_counters": hashlib.sha256(np.ascontiguousarray(vcnt).tobytes()).hexdigest() if want_vp else None}
    shards = [shard]
    blocks = [(vcnt, vp["yy"], vp["yp"])] if (args.results_npz and want_vp) else None
    if world > 1:
        shards = [None] * world
        dist.all_gather_object(shards, shard)
        if args.results_npz and want_vp:
            blocks = [None] * world
            dist.all_gather_object(blocks, (vcnt, vp["yy"], vp["yp"]))
    if rank == 0 and args.results_npz and want_vp:
        np.savez(args.results_npz, first=np.array([sh["first"] for sh in shards]), counts=np.concatenate([b[0] for b in blocks], axis=1),
                 yy=np.concatenate([b[1] for b in blocks], axis=0), yp=np.concatenate([b[2] for b in blocks], axis=0))


    extras = None
    if world == 1 and not args.no_extras:
        passes = ([vp] if want_vp else []) + [run.whole_pass(4) for _ in range(max(1, args.passes) - (1 if want_vp else 0))]
        rates = [p["iters"] / p["seconds"] for p in passes]
        # the same passes with one host round trip per Newton iteration (idaens_set_fused_newton(0)): the before/after of the
        # device-side convergence tests (first slice); identical work and results, only the pace changes
        unfused_rates = [(lambda p: p["iters"] / p["seconds"])(run.whole_pass(4, fused=0, device_ctl=0)) for _ in range(3)]
        pth = passes[0]["paths"]
        extras = {
            "reference_text_paths": {
                "sum_ncfn": pth["ncfn"], "sum_newton_internal_resetups": pth["nls_nconvfails"], "lu_failures_Q2": pth["nlufail"],
                "newton_gave_up_with_current_jacobian_Q3_Q4": pth["nconv_jcur"], "failed_attempts_before_first_step_Q5": pth["nfail_first"],
                "root_function_evaluations": pth["nge"], "root_returns": 0,
                "note": "sums over the batch of one whole pass: how often a system took a path on which oracle and product follow C IDA "
                        "where the reference's text does otherwise (SURVEY.md 9, DESIGN.md 2); zero = the deviation is not exercised by this "
                        "configuration (Newton-internal re-setups follow the reference's text: newton.rs:146-152); no root functions are set"},
            "whole_pass": {"value": statistics.median(rates), "unit": "Newton iters/s", "passes": [round(r, 1) for r in rates],
                           "newton_iters_per_pass": passes[0]["iters"], "rounds_per_pass": passes[0]["rounds"],
                           "seconds_median": statistics.median(p["seconds"] for p in passes),
                           "stepper": {0: "host lock-step", 1: "device, one thread per system", 2: "device lock-step rounds"}[passes[0]["stepper"]],
                           "protocol": "SURVEY 8(d): every system from fresh state through its whole output schedule, exact LU, "
                                       "median of %d passes, wall time of the pass with inputs resident" % len(passes)},
            "newton_fusion": {"whole_pass_value_with_host_ctest_every_iteration": statistics.median(unfused_rates), "unit": "Newton iters/s",
                              "note": "host stepper with one host round trip per Newton iteration (idaens_set_fused_newton(0) on top of "
                                      "idaens_set_device_controller(0)): round 1's control flow; same work, same results, median of 3 passes"},
            "device_controller": {
                "whole_pass_value_with_the_host_stepper": statistics.median(
                    [(lambda p: p["iters"] / p["seconds"])(run.whole_pass(4, device_ctl=0)) for _ in range(3)]),
                "unit": "Newton iters/s",
                "note": "`value` and `whole_pass` run with the step-size and order controller on the device (n <= 8: idahip_tiny_solve, "
                        "the whole of Ida::solve in one launch; larger n: idahip_round_solve, lock-step rounds enqueued without a host "
                        "round trip inside a round), pow with glibc's bits; this is the whole pass with the lock-step host stepper "
                        "instead (idaens_set_device_controller(0)): same work, same results"},
        }

    if rank == 0:
        ab = algorithmic_bytes(args.n, args.workload)
        cls_names = ("newton_iter", "sys", "sys_jac", "jac", "lu", "vector")
        total_ms = sum(tim[k]["ms"] for k in cls_names)
        classes = {}
        for k in cls_names:
            v = tim[k]
            ent = {"ms": round(v["ms"], 3), "calls": v["launches"], "systems": v["systems"],
                   "share_of_device_time": round(v["ms"] / total_ms, 4) if total_ms > 0 else None}
            if ab.get(k) and v["ms"] > 0:
                gbs = round(ab[k] * v["systems"] / (v["ms"] * 1e-3) / 1e9, 1)
                if args.workload == "heat1d" and k == "newton_iter":
                    # the factors of the banded Jacobian are zero almost everywhere and the solves leave all-zero 64 x 64 blocks out
                    # (exactly: solve_kernels.hpp): the dense byte count is not what moves, so no fraction of the peak is claimed
                    ent["GB/s_dense_equivalent"] = gbs
                else:
                    ent["GB/s"] = gbs
                    ent["frac_of_hbm_peak"] = round(gbs / HBM_PEAK_GBS, 4)
            classes[k] = ent
        lu_kernels = {}
        for k in ("lu_panel", "lu_trail", "lu_finalize"):
            v = tim2[k]
            lu_kernels[k] = {"ms": round(v["ms"], 3), "launches": v["launches"], "matrix_launches": v["systems"]}
        # the dominant kernel: by device time among the classes' kernels, with the LU split into its kernels
        lu_sub_ms = sum(tim2[k]["ms"] for k in ("lu_panel", "lu_trail", "lu_finalize"))
        scale = tim["lu"]["ms"] / lu_sub_ms if lu_sub_ms > 0 else 0.0  # level-2 pass -> level-1 pass (same work, fewer syncs)
        cand = {k: tim[k]["ms"] for k in ("newton_iter", "sys", "sys_jac", "jac")}
        cand.update({k: tim2[k]["ms"] * scale for k in ("lu_panel", "lu_trail", "lu_finalize")})
        dom = max(cand, key=cand.get)
        if dom == "lu_trail" and args.n > 8 and args.workload != "heat1d":
            flops, nbytes, nl = trailing_work(args.n)
            mats = tim2["lu"]["systems"]                    # matrices factorised in the level-2 pass
            v = tim2["lu_trail"]
            tfl = flops * mats / (v["ms"] * 1e-3) / 1e12
            tr = profiled_traffic("lu_trail64w_kernel") if (args.workload == "linear_dense" and args.n == 512) else None
            roof = {"bound": "valu", "kernel": "lu_trail64w_kernel<%d> (rank-64 trailing update + U12 solve of the batched getrf)" % (1024 if args.n <= 1024 else 4096),
                    "share_of_device_time": round(cand[dom] / total_ms, 4),
                    "achieved": round(tfl, 2), "peak": VALU_UNFUSED_TFLOPS, "unit": "TFLOP/s", "frac": round(tfl / VALU_UNFUSED_TFLOPS, 4),
                    "peak_note": "fp64 vector ceiling of the reference's arithmetic: a multiply and a subtract per update (dense.rs:151), "
                                 "half of the 78.6 TFLOP/s FMA peak at 2.4 GHz; the chip holds about 1.8 GHz in this kernel",
                    "frac_of_fma_peak": round(tfl / VALU_FMA_TFLOPS, 4),
                    "avg_launch_ms": round(v["ms"] / max(1, v["launches"]), 4), "launches": v["launches"],
                    "algorithmic_flops_per_matrix": flops, "algorithmic_bytes_per_matrix": nbytes,
                    "hbm_GB/s_on_algorithmic_bytes": round(nbytes * mats / (v["ms"] * 1e-3) / 1e9, 1),
                    "traffic": None if tr is None else int(tr["hbm_bytes_per_matrix"] * mats / max(1, v["launches"])),
                    "traffic_bytes_per_matrix": None if tr is None else tr["hbm_bytes_per_matrix"],
                    "traffic_source": None if tr is None else "%s (rocprofv3 --pmc passes of this command at commit %s)" % (tr["source"], tr["commit"])}
        elif args.n <= 8:
            # the device-resident stepper: one kernel does everything; timed under the "vector" class
            d = tim["vector"]
            nb = 24 * args.n * args.n + 80 * args.n  # algorithmic bytes of one Newton iteration
            ach = nb * iters_all / (elapsed_max * 1e9) if elapsed_max > 0 else 0.0
            roof = {"bound": "hbm", "kernel": "tiny_ida_kernel (the whole of Ida::solve, one thread per system, controller on the device)",
                    "share_of_device_time": 1.0, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 6),
                    "traffic": None, "avg_launch_ms": round(d["ms"] / max(1, d["launches"]), 4),
                    "note": "latency bound: 72-byte matrices, one thread per system (SURVEY 8(d): the roofline fraction is meaningless "
                            "for this configuration; iterations per second is the figure)"}
        else:
            cls = "lu" if dom.startswith("lu_") else dom  # (banded workload: the getrf as a whole on its algorithmic bytes)
            d = tim[cls]
            ach = ab.get(cls, 0) * d["systems"] / (d["ms"] * 1e-3) / 1e9 if d["ms"] > 0 else 0.0
            roof = {"bound": "hbm", "kernel": {"sys": "linear_sys_kernel<2, false>", "sys_jac": "linear_sys_kernel<2, true>",
                                               "newton_iter": "newton_iter_kernel<2>", "jac": "linear_jac_kernel",
                                               "lu": "batched getrf (all its kernels: panel, trailing update, row scatter)"}.get(cls, cls),
                    "share_of_device_time": round((d["ms"] if cls == "lu" else cand[dom]) / total_ms, 4) if total_ms > 0 else None,
                    "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                    "avg_launch_ms": round(d["ms"] / max(1, d["launches"]), 4)}
        name, wl = WORKLOADS[args.workload]
        out = {
            "metric": name % (args.n, args.batch),
            "value": iters_all / elapsed_max,
            "unit": "Newton iters/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed_max / max(1, args.steps),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": wl % (args.n, args.batch), "n": args.n, "batch_per_gpu": args.batch, "total_batch": args.batch * world,
                       "sharding": "independent systems, contiguous block per rank, no collective",
                       "lu_period": args.lu_period,
                       "groups_per_gpu": args.groups, "group_sizes": gsz, "mutually_concurrent_streams": nconc, "least_pairwise_stream_share": None if share_min is None else round(share_min, 3),
                       "groups": "the rank's systems run as %d ensemble(s) side by side on the one device, each on its own HIP stream and host "
                                 "thread (idaens_stream_group): every system is integrated exactly as alone; the groups only fill each other's idle "
                                 "stretches of a lock-step round%s" % (args.groups, " -- here they take turns (serial)" if run.serial else "")},
            "newton_iters_timed": iters_all,
            "per_rank": None if world == 1 else {"newton_iters": [p[0] for p in per_rank], "seconds": [p[1] for p in per_rank],
                                                 "input_generation_s": [sh["input_generation_s"] for sh in shards],
                                                 "devices": [sh["device"] for sh in shards],
                                                 "generator_processes": procs, "cores_available_to_rank0": cores,
                                                 "process_group": backend + " (barrier and two scalars only; no data-path collective)"},
            "ensemble_result": None if not want_vp else {
                "systems": sum(sh["count"] for sh in shards), "shards": [{k: sh[k] for k in ("rank", "first", "count", "sum_nst", "sum_nni", "sum_nsetups", "sha256_yy", "sha256_counters")} for sh in shards],
                "sum_nst": sum(sh["sum_nst"] for sh in shards), "sum_nni": sum(sh["sum_nni"] for sh in shards),
                "note": "SURVEY 8(e): every rank integrates its shard once from fresh state through the whole schedule (untimed) and rank 0 "
                        "concatenates: digests of y(tout) and of the counters (nst, netf, ncfn, nni, nsetups, kused) per shard, in rank = global "
                        "system order; --results-npz writes the concatenated arrays (tests/test_gpu_bench_ranks.py compares them with a "
                        "single-process run)"},
            "roofline": roof,
            "kernel_classes_rank0": classes,
            "lu_kernels_rank0": lu_kernels,
            "kernel_figures_measured_on": ("the %d groups' one ensemble" % G if G == 1 else
                                           "ONE ensemble of all %d systems of the rank, alone on the device, after the timed region (K rounds with per-class "
                                           "event timers, K more with per-kernel timers for the LU): kernel_classes_rank0, lu_kernels_rank0, roofline, "
                                           "lu_plus_solve and hbm_roofline's iterations per setup are the kernels' own figures at whole-batch launch sizes; "
                                           "`value` runs the same kernels as %d groups side by side" % (count, G)),
            "lu_plus_solve": lu_plus_solve(tim, args.n, "unfused", dense=args.workload != "heat1d") if args.n > 8 else None,
            "cpu_baseline": cpu,
            "input_generation_s": round(t_gen, 1),
            "stream": {"stagger_rounds": run.stagger, "spin_up_rounds": max(Runner.SPIN_UP, 3 * run.stagger),
                       "note": "first starts spread over one median integration length (rounds), measured on the device on the "
                               "first %d systems before the stream starts; then the untimed spin-up, W warm-up and K timed rounds" % Runner.CALIBRATION_SYSTEMS},
        }
        if args.workload == "linear_dense":
            # north_star: the rate as a fraction of the HBM roofline. BASELINE.md section 3: one Newton iteration moves 24 N^2 + 80 N
            # bytes (solve + residual), a linear-solver setup (40 N^2 + 8 N bytes: Jacobian + factor) is shared by the r iterations
            # measured per setup in this run
            N = args.n
            it_bytes, ls_bytes = 24 * N * N + 80 * N, 40 * N * N + 8 * N
            n_it = tim["newton_iter"]["systems"]
            n_ls = tim["lu"]["systems"]
            r = n_it / n_ls if n_ls else None
            per_gpu = 8.0e12 / it_bytes
            out["hbm_roofline"] = {
                "iters_per_s_per_gpu": round(per_gpu), "frac": out["value"] / (world * per_gpu),
                "newton_iters_per_lsetup": None if r is None else round(r, 2),
                "iters_per_s_per_gpu_with_lsetup": None if r is None else round(8.0e12 / (it_bytes + ls_bytes / r)),
                "frac_with_lsetup": None if r is None else out["value"] / (world * 8.0e12 / (it_bytes + ls_bytes / r)),
                "note": "8 TB/s over the algorithmic bytes of BASELINE.md section 3; the factorisation itself is bound by the fp64 vector "
                        "pipe at this N, not by HBM (`roofline`, `lu_plus_solve`)"}
        if extras:
            out.update(extras)
        if TIME_ALL:
            out["timers"] = "IDAHIP_BENCH_TIME_ALL=1: per-kernel LU timers on in every launch of the process, `value` includes their synchronisations"
        if rehearse:
            out["rehearsal"] = "all ranks on cuda:0 -- not a scaling measurement"
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
