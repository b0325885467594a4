"""Arguments for the pow tests: the domain of the reference's step-size / order controller and Newton rate estimate
(/root/reference/src/lib.rs:1163-1169, src/impl_complete_step.rs:128-132, src/ida_nls.rs:249-253): bases 2*err + 1e-4 and
norm ratios, exponents +-1/m with m = 1..6. Deterministic: a splitmix64 sequence in integer arithmetic."""
import numpy as np


def _splitmix64(n, seed):
    with np.errstate(over="ignore"):
        z = (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) + np.uint64(seed)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def pow_cases(n, seed=20261004):
    u = _splitmix64(n, seed)
    v = _splitmix64(n, seed + 1)
    # base: a random mantissa with a binary exponent in [-40, 40) -- norm ratios down to 1e-12, error estimates up to 1e12
    mant = (u & np.uint64((1 << 52) - 1)) | np.uint64(0x3FF0000000000000)
    x = mant.view(np.float64) * np.exp2(((u >> np.uint64(52)) % np.uint64(80)).astype(np.float64) - 40.0)
    m = ((v >> np.uint64(8)) % np.uint64(6)).astype(np.float64) + 1.0
    sign = np.where((v & np.uint64(1)) == 0, 1.0, -1.0)
    y = sign * (1.0 / m)
    return np.ascontiguousarray(x), np.ascontiguousarray(y)
