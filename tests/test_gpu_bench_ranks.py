"""bench.py under world_size 2 and 4 on the HIP path: the ranks the driver starts for `--gpus N` share nothing
but a barrier and a few scalars -- and their concatenated per-system results equal a single process's."""
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
@pytest.mark.parametrize("ranks", [2, 4])
def test_bench_ranks_rehearsal(ranks, tmp_path):
    """`python bench.py --gpus N` as the driver starts it for N > 1 (fresh rank processes, gloo for the barrier and the two
    scalars, no RCCL), rehearsed with every rank on cuda:0: the HIP path under world_size N -- rank r integrates its own shard
    (uploaded slice by slice from the generator) and the line carries the sum over ranks. Four ranks is what a one-GPU box
    admits next to this process (at most six processes on the card); the eight-rank start is rehearsed on the host side only
    (tests/test_bench_contract.py, bench.py --inputs-only)."""
    import json
    import subprocess
    import sys
    import numpy as np
    env = dict(os.environ, IDAHIP_BENCH_REHEARSE="1", IDAHIP_GEN_PROCS="1")
    npz = str(tmp_path / "results.npz")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--n", "64", "--batch", "64", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline", "--no-extras", "--results-npz", npz], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == ranks and line["steps"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    pr = line["per_rank"]
    assert len(pr["newton_iters"]) == ranks and all(v > 0 for v in pr["newton_iters"]) and sum(pr["newton_iters"]) == line["newton_iters_timed"]
    assert pr["process_group"].startswith("gloo")
    # run hygiene of an N-rank line: every rank says how long its inputs took and which device it ran on
    assert len(pr["input_generation_s"]) == ranks and len(pr["devices"]) == ranks
    assert all(d["local_rank"] == 0 and "name" in d and d["total_memory_GiB"] > 0 for d in pr["devices"]), pr["devices"]  # (the rehearsal puts every rank on cuda:0, and says so)
    assert "rehearsal" in line and pr["generator_processes"] == 1

    # ---- results, not contract fields ("host concatenates"): the ranks' blocks, concatenated in rank order, are
    # bit for bit what ONE process computes on the HIP path for the same 64 * ranks systems
    er = line["ensemble_result"]
    assert er["systems"] == 64 * ranks and [sh["first"] for sh in er["shards"]] == [64 * r for r in range(ranks)]
    assert len({sh["sha256_yy"] for sh in er["shards"]}) == ranks  # distinct shards: nobody integrated someone else's block
    got = np.load(npz)
    import idahip
    from idahip import problems
    p = problems.linear_dense(n=64, batch=64 * ranks, procs=1)
    ctx = problems.make_ctx(p)
    ens = idahip.Ensemble(ctx, p["yy0"], p["yp0"])
    status, _, reached = ens.solve_schedule(p["touts"])
    assert (status == 0).all() and (reached == len(p["touts"])).all()
    c = ens.counters()
    for row, k in enumerate(("nst", "netf", "ncfn", "nni", "nsetups", "kused")):
        assert np.array_equal(got["counts"][row], c[k]), k
    assert np.array_equal(got["yy"], ens.yy()) and np.array_equal(got["yp"], ens.yp())
    assert er["sum_nst"] == int(c["nst"].sum()) and er["sum_nni"] == int(c["nni"].sum())
    import hashlib
    for r, sh in enumerate(er["shards"]):
        assert sh["sha256_yy"] == hashlib.sha256(np.ascontiguousarray(ens.yy()[64 * r:64 * (r + 1)]).tobytes()).hexdigest()
    ens.close()
    ctx.close()
