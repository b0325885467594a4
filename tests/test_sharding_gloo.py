"""The N > 1 path on CPU: two gloo ranks shard an ensemble exactly as bench.py does (contiguous blocks of independent
systems, no data-path collective), integrate their shards with the CPU oracle, and combine time / iteration counts.
The union of the shards must be bit-identical to the single-process run over the whole batch."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, PER_RANK, WORLD = 24, 3, 2
TOUTS = [0.1, 0.2]


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "rust-ida_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from idahip import problems, sharding
    import oracle_lib as O
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    first, count = sharding.shard_range(rank, world, PER_RANK)
    p = problems.linear_dense(n=N, batch=count, first=first)
    r = O.run_ensemble(p["kind"], N, p["yy0"], p["yp0"], p["rtol"], p["atol"], TOUTS, A=p["A"], B=p["B"], c=p["c"], nthreads=1)
    dist.barrier()
    tmax, total = sharding.combine(1.0 + rank, int(r["counters"]["nni"].sum()), dist)
    q.put((rank, first, r["yy"][-1], r["counters"]["nni"], tmax, total))
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process():
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "rust-ida_amd"))
    from idahip import problems, sharding
    import oracle_lib as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, WORLD, port, q)) for r in range(WORLD)]
    for p_ in procs:
        p_.start()
    res = sorted([q.get(timeout=180) for _ in range(WORLD)], key=lambda x: x[0])
    for p_ in procs:
        p_.join(timeout=60)
        assert p_.exitcode == 0
    full = problems.linear_dense(n=N, batch=PER_RANK * WORLD, first=0)
    ref = O.run_ensemble(full["kind"], N, full["yy0"], full["yp0"], full["rtol"], full["atol"], TOUTS, A=full["A"], B=full["B"],
                         c=full["c"], nthreads=1)
    yy = np.concatenate([r[2] for r in res])
    nni = np.concatenate([r[3] for r in res])
    assert [r[1] for r in res] == [0, PER_RANK]
    assert np.array_equal(yy, ref["yy"][-1]) and np.array_equal(nni, ref["counters"]["nni"])
    for r in res:  # every rank sees max(time) and sum(iterations)
        assert r[4] == 2.0 and r[5] == int(ref["counters"]["nni"].sum())


def test_shard_range_and_single_process_combine():
    sys.path.insert(0, os.path.join(ROOT, "rust-ida_amd"))
    from idahip import sharding
    assert sharding.shard_range(3, 8, 4096) == (3 * 4096, 4096)
    with pytest.raises(ValueError):
        sharding.shard_range(8, 8, 4096)
    assert sharding.combine(0.5, 7) == (0.5, 7)
