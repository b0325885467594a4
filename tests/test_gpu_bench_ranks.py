"""bench.py under world_size 2 on the HIP path: the ranks the driver starts for `--gpus N` share nothing but a
barrier and two scalars."""
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
@pytest.mark.parametrize("ranks", [2, 4])
def test_bench_ranks_rehearsal(ranks):
    """`python bench.py --gpus N` as the driver starts it for N > 1 (fresh rank processes, gloo for the barrier and the two
    scalars, no RCCL), rehearsed with every rank on cuda:0: the HIP path under world_size N -- rank r integrates its own shard
    (uploaded slice by slice from the generator) and the line carries the sum over ranks. Four ranks is what a one-GPU box
    admits next to this process (at most six processes on the card); the eight-rank start is rehearsed on the host side only
    (tests/test_bench_contract.py, bench.py --inputs-only)."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, IDAHIP_BENCH_REHEARSE="1", IDAHIP_GEN_PROCS="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--n", "64", "--batch", "64", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline", "--no-extras"], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == ranks and line["steps"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    pr = line["per_rank"]
    assert len(pr["newton_iters"]) == ranks and all(v > 0 for v in pr["newton_iters"]) and sum(pr["newton_iters"]) == line["newton_iters_timed"]
    assert pr["process_group"].startswith("gloo")
