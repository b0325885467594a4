"""Config 3/5 inputs generated a slice at a time (problems.linear_dense_slices, what bench.py uploads from): the same systems,
bit for bit, as the whole-shard generator, in-process and with worker processes."""
import numpy as np
import pytest


@pytest.mark.parametrize("procs", [1, 3])
def test_slices_equal_the_whole_shard(procs):
    from idahip import problems
    n, batch, first = 12, 25, 7
    whole = problems.linear_dense(n=n, batch=batch, first=first, procs=1)
    parts = [tuple(np.array(x) if isinstance(x, np.ndarray) else x for x in sl)
             for sl in problems.linear_dense_slices(n=n, batch=batch, first=first, procs=procs, slice_bytes=16 * n * n * 6)]
    assert len(parts) > 1 and [p[0] for p in parts] == sorted(p[0] for p in parts)
    for i, key in ((1, "A"), (2, "B"), (3, "c"), (4, "yy0"), (5, "yp0")):
        assert np.array_equal(np.concatenate([p[i] for p in parts]), whole[key]), key
