#!/usr/bin/env python3
"""bench.py -- Newton iterations / second (fp64) of the batched BDF/Newton hot path on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched by torch.distributed.run, one
rank per GPU. Prints ONE JSON line on rank 0.

Workload (BASELINE.json configs[2]): synthetic random linear dense index-1 DAE
F = A y' + B y - c, N = 512, batch B = 4096 systems per GPU (distinct matrices per system, all device resident),
rtol 1e-6, atol 1e-8, integrated from t = 0 to t = 1 with outputs every 0.1 (ten Ida::solve calls per system, handed over
as one schedule: systems do not wait for each other at the outputs). Throughput mode: a system that has reached t = 1 is
created anew from its initial conditions and integrates again, so the batch never drains and the measured rate is that of
an endless stream of such integrations -- it does not depend on which rounds K and W select (the systems' first starts are
staggered over 96 rounds and 200 untimed rounds precede the warm-up, so the batch is spread evenly over the phases of an
integration).
A "step" is one lock-step step attempt of the whole batch: set_coeffs -> predict -> Newton solve (residual, Jacobian +
batched LU when the reference's rule asks for it, 1..4 triangular solves + WRMS norms) -> error test -> complete_step
or restore, for every one of the B systems. Nothing is skipped: every accepted step is bit-identical to the CPU
oracle's (tests/test_gpu_ensemble.py, incl. test_streaming_restarts_reproduce_fresh_integrations).
value = (Newton iterations of all systems on all ranks during the K timed steps) / (max over ranks of the wall time
of those steps), inputs already resident in HBM.
Multi-GPU (config 5): the ensemble shards embarrassingly -- rank r integrates systems [4096 r, 4096 (r+1)); no data-path
collective; torch.distributed (RCCL) is used only for the barrier and the max-over-ranks of the time.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "rust-ida_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)


def algorithmic_bytes(n, workload="linear_dense"):
    """Per-system algorithmic HBM bytes of each kernel class (fp64 = 8 B)."""
    if workload == "heat1d":  # three-point residual, dense Jacobian written (mostly zeros)
        return {"newton_iter": 8 * n * n + 40 * n + 8, "sys": 56 * n, "jac": 8 * n * n, "sys_jac": 8 * n * n + 56 * n,
                "lu": 16 * n * n + 8 * n}
    return {
        "newton_iter": 8 * n * n + 40 * n + 8,   # getrs 8N^2+24N, + neg/scale/axpy/wrms vectors 16N+8
        "sys": 16 * n * n + 40 * n,              # residual of the linear dense DAE
        "jac": 24 * n * n,                       # J = B + cj A
        "sys_jac": 24 * n * n + 40 * n,          # residual and J = B + cj A in one pass over A and B
        "lu": 16 * n * n + 8 * n,                # getrf: read + write the matrix once, pivots
    }


KERNEL_OF_CLASS = {
    "lu": "batched getrf = the lu_panel2 / lu_trail / lu_trail64 / lu_finalize kernel launches of one idahip_nls_lsetup call",
    "sys": "linear_sys_kernel<2, false>", "newton_iter": "newton_iter_kernel", "jac": "linear_jac_kernel",
    "sys_jac": "linear_sys_kernel<2, true>",
}


def profiled_traffic(cls):
    """HBM bytes per system of the kernel class from the committed rocprofv3 PMC summary (separate --pmc FETCH_SIZE /
    WRITE_SIZE passes, gfx950-corrected; profiles/README.md). PMC counters cannot be read inside this process, so the
    figure is the profiled one for the same command, or None when no summary is present."""
    path = os.path.join(ROOT, "profiles", "r01_bench_w0_summary.json")
    try:
        s = json.load(open(path))
    except Exception:
        return None
    names = {"lu": ("lu_",), "sys": ("linear_sys_kernel<2, false>", "linear_sys_kernel<1, false>"),
             "sys_jac": ("linear_sys_kernel<2, true>", "linear_sys_kernel<1, true>"),
             "newton_iter": ("newton_iter_kernel",), "jac": ("linear_jac_kernel",)}[cls]
    tot = 0.0
    for k, v in s["kernels"].items():
        if k.startswith(names) and "hbm_read_GB_total" in v:
            tot += v["hbm_read_GB_total"] + v["hbm_write_GB_total"]
    systems = s["bench"]["kernel_classes_rank0"].get(cls, {}).get("systems", 0)
    return {"hbm_bytes_per_system": int(tot * 1e9 / max(1, systems)), "source": "profiles/r01_bench_w0_summary.json"} if tot else None


def lu_plus_solve(tim, n):
    """The kernel-level figure north_star states its target on: one batched getrf + one getrs per system, algorithmic
    bytes (24 N^2 + 32 N) over the device time per system of the lu class plus the newton_iter class
    (whose kernel is the getrs with the Newton vector updates fused in)."""
    lu, ni = tim["lu"], tim["newton_iter"]
    if lu["systems"] == 0 or ni["systems"] == 0:
        return None
    us = 1e3 * (lu["ms"] / lu["systems"] + ni["ms"] / ni["systems"])
    gbs = (24 * n * n + 32 * n) / (us * 1e-6) / 1e9
    return {"us_per_system": round(us, 3), "GB/s": round(gbs, 1), "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4),
            "note": "reference-exact arithmetic (unfused mul, sub) caps the getrf at 23 % of the HBM roofline (DESIGN.md section 4)"}


def roofline(dom, d, alg_bytes, achieved, n):
    """The `roofline` object of the JSON line for the kernel class with the most device time. achieved = algorithmic bytes
    per launch / average launch duration (HIP events on the ctx stream, timed region); traffic = measured HBM bytes per
    launch, i.e. the committed PMC figure per system x the systems one launch of this run processed."""
    spl = d["systems"] / max(1, d["launches"])
    tr = profiled_traffic(dom) if n == 512 else None  # the committed PMC passes are of the N = 512 headline workload
    extra = {}
    if n < 0:  # not the headline workload: no committed traffic profile
        n = -n
        tr = None
    if dom == "lu" and d["ms"] > 0:
        # the reference-exact elimination is an unfused mul + sub per update: 2 fp64 VALU ops, ~n^3/3 updates per matrix;
        # ceiling = 256 CUs x 64 lanes x 2.4 GHz / 2 ops
        upd = (n ** 3 / 3.0) * d["systems"] / (d["ms"] * 1e-3)
        peak = 256 * 64 * 2.4e9 / 2.0
        extra = {"valu": {"achieved": round(upd / 1e12, 3), "peak": round(peak / 1e12, 2), "unit": "T updates/s (mul + sub, fp64)",
                          "frac": round(upd / peak, 4)}}
    return {"bound": "hbm", "kernel": KERNEL_OF_CLASS[dom], **extra, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": None if tr is None else int(tr["hbm_bytes_per_system"] * spl),
            "traffic_bytes_per_system": None if tr is None else tr["hbm_bytes_per_system"],
            "traffic_source": None if tr is None else tr["source"],
            "algorithmic_bytes_per_launch": int(alg_bytes * spl), "algorithmic_bytes_per_system": alg_bytes,
            "avg_launch_ms": round(d["ms"] / max(1, d["launches"]), 4), "systems_per_launch": round(spl, 1),
            "note": "a launch of the lu class is one batched getrf (all its kernel launches); for the LU the fp64 VALU ceiling "
                    "binds before HBM (DESIGN.md section 4)"}


# IDAHIP_BENCH_TIME_ALL=1 (tools/profile_bench.sh): the HIP-event kernel-class timers run from the first launch of the process
# (spin-up and warm-up included), so that kernel_classes_rank0 can be checked against a rocprofv3 kernel trace of the same
# process; `value` is unaffected.
TIME_ALL = os.environ.get("IDAHIP_BENCH_TIME_ALL") == "1"


class Lane:
    """One contiguous slice of the rank's systems in throughput mode (idaens_stream): the ten Ida::solve calls of the
    workload are one output schedule per system, a system that has reached t = 1 is created anew (Ida::new from its
    initial conditions) and starts over at once. The batch never drains: every lock-step round works on every system,
    each somewhere else in its integration, so the rate does not depend on which rounds are timed."""

    STAGGER = 96   # the systems' first starts are spread over this many rounds (about one integration)
    SPIN_UP = 200  # untimed rounds before the warm-up: the stagger plus one more integration

    def __init__(self, prob, device, stream=None):
        import idahip
        from idahip import problems
        self.prob = prob
        self.ctx = problems.make_ctx(prob, device=device, stream=stream)
        self.ens = idahip.Ensemble(self.ctx, prob["yy0"], prob["yp0"])
        if TIME_ALL:
            self.ctx.timing(True)
            self.ctx.timing_reset()
        self.passes = self.ens.stream(prob["touts"], self.SPIN_UP, stagger_rounds=self.STAGGER)

    def total_iters(self):
        return self.ens.total_newton_iters()

    def step(self):
        """Exactly one lock-step round."""
        before = self.ens.total_rounds()
        self.passes = self.ens.stream(self.prob["touts"], 1)
        assert self.ens.total_rounds() == before + 1


class Runner:
    """The rank's whole batch as one ensemble on one device context and stream. (Two or four slices stepped concurrently
    from their own host threads were tried: +1.5 % at best -- the kernels of the slices slow each other down.)"""

    def __init__(self, prob, device):
        self.lane = Lane(prob, device)

    def total_iters(self):
        return self.lane.total_iters()

    def step(self):
        self.lane.step()

    def sync(self):
        c = self.lane.ctx
        c._chk(c.H.idahip_sync(c.h), "sync")

    def timing(self, on):
        self.lane.ctx.timing(on)
        self.lane.ctx.timing_reset()

    def timing_get(self):
        return self.lane.ctx.timing_get()


def cpu_baseline(prob_small, cores):
    """The reference's CPU path as restated in oracle/ (kind = "port"), timed on this host's cores on a bounded sample of
    the same workload: the first len(sample) systems of config 3, integrated t = 0 -> 1 like the GPU run."""
    import oracle_lib as O
    n = prob_small["n"]
    r = O.run_ensemble(prob_small["kind"], n, prob_small["yy0"], prob_small["yp0"], prob_small["rtol"], prob_small["atol"],
                       prob_small["touts"], params=prob_small.get("params"), A=prob_small.get("A"), B=prob_small.get("B"),
                       c=prob_small.get("c"), nthreads=cores)
    iters = int(r["counters"]["nni"].sum())
    return {"value": iters / r["seconds"], "unit": "Newton iters/s", "cores": cores, "kind": "port",
            "sample": "%d systems of the same N=%d workload, full t=0..%g integration, %d Newton iterations in %.2f s, one std::thread per core"
                      % (prob_small["yy0"].shape[0], n, float(prob_small["touts"][-1]), iters, r["seconds"])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--batch", type=int, default=4096, help="systems per GPU")
    ap.add_argument("--workload", choices=("linear_dense", "heat1d"), default="linear_dense",
                    help="linear_dense = config 3 (the headline, default N=512 B=4096); heat1d = config 4 (use --n 4096 --batch 256)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world

    # ---- inputs: generated before this process touches the GPU (forked worker processes)
    from idahip import problems
    cores = os.cpu_count() or 1
    procs = int(os.environ.get("IDAHIP_GEN_PROCS", max(1, min(16, cores // max(1, world)))))  # 1 = no fork (use under rocprofv3)
    t0 = time.time()
    from idahip import sharding
    first, count = sharding.shard_range(rank, world, args.batch)
    if args.workload == "heat1d":
        full = problems.heat1d(n=args.n, batch=args.batch * world)  # kappa_b depends on the global system id
        prob = {k: (v[first:first + count] if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == args.batch * world else v)
                for k, v in full.items()}
    else:
        prob = problems.linear_dense(n=args.n, batch=count, first=first, procs=procs)
    t_gen = time.time() - t0

    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        ncpu = max(1, min(cores, 64))
        nsmall = min(args.batch, 16 * ncpu)  # ~10 s of wall time on 64 threads
        small = {k: (v[:nsmall] if isinstance(v, np.ndarray) and v.ndim >= 2 and v.shape[0] == args.batch else v) for k, v in prob.items()}
        cpu = cpu_baseline(small, ncpu)

    import torch
    import torch.distributed as dist
    # Rehearsal knob for a one-GPU box: IDAHIP_BENCH_REHEARSE=1 puts every rank on cuda:0 and uses gloo for the barrier
    # and the max-time (RCCL refuses two ranks on one device). Never set by the driver; the JSON line says so.
    rehearse = os.environ.get("IDAHIP_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run = Runner(prob, local_rank)
    prob.pop("A", None)  # host copies no longer needed
    prob.pop("B", None)

    for _ in range(args.warmup):
        run.step()
    if not TIME_ALL:
        run.timing(True)
    barrier()
    it0 = run.total_iters()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run.step()
    run.sync()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    iters = run.total_iters() - it0
    barrier()
    tim = run.timing_get()

    elapsed_max, iters_all = sharding.combine(elapsed, iters, dist if world > 1 else None, device="cpu" if rehearse else "cuda")

    if rank == 0:
        ab = algorithmic_bytes(args.n, args.workload)
        dom = max(("newton_iter", "sys", "sys_jac", "jac", "lu"), key=lambda k: tim[k]["ms"])
        d = tim[dom]
        achieved = (ab[dom] * d["systems"]) / (d["ms"] * 1e-3) / 1e9 if d["ms"] > 0 else 0.0
        classes = {}
        for k in ("newton_iter", "sys", "sys_jac", "jac", "lu", "vector"):
            v = tim[k]
            ent = {"ms": round(v["ms"], 3), "calls": v["launches"], "systems": v["systems"]}
            if k in ab and v["ms"] > 0:
                ent["GB/s"] = round(ab[k] * v["systems"] / (v["ms"] * 1e-3) / 1e9, 1)
            classes[k] = ent
        out = {
            "metric": ("Newton iters/sec (fp64), batched dense DAE N=%d B=%d" if args.workload == "linear_dense" else
                       "Newton iters/sec (fp64), 1-D heat equation (dense Jacobian) N=%d B=%d") % (args.n, args.batch),
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
            "config": {"workload": ("random linear dense index-1 DAE F=A y'+B y-c (SURVEY 8(d) config 3), N=%d, B=%d systems per GPU, "
                                    "rtol 1e-6 atol 1e-8, t=0..1 with 10 outputs, every system restarts on its own when it reaches "
                                    "t=1 (endless stream of integrations)" if args.workload == "linear_dense" else
                                    "1-D heat equation by the method of lines, Dirichlet ends algebraic, dense Jacobian (SURVEY 8(d) "
                                    "config 4), N=%d, B=%d systems per GPU, rtol 1e-5 atol 1e-8, t=0..0.1 with 10 outputs, every "
                                    "system restarts on its own at the end") % (args.n, args.batch),
                       "n": args.n, "batch_per_gpu": args.batch, "total_batch": args.batch * world,
                       "sharding": "independent systems, contiguous block per rank, no collective"},
            "newton_iters_timed": iters_all,
            "roofline": roofline(dom, d, ab[dom], achieved, args.n if args.workload == "linear_dense" else -args.n),
            "kernel_classes_rank0": classes,
            "lu_plus_solve": lu_plus_solve(tim, args.n),
            "cpu_baseline": cpu,
            "input_generation_s": round(t_gen, 1),
        }
        if TIME_ALL:
            out["kernel_classes_cover"] = "every launch of the process (spin-up, warm-up, timed steps)"
        if rehearse:
            out["rehearsal"] = "all ranks on cuda:0 over gloo -- not a scaling measurement"
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
