"""The parity yardstick for float sums: a float32 result that is a sum of n signed terms is held
to  |got - f64| <= REL * sum|terms|  -- the condition-number form -- with REL = 1e-5, the tolerance BASELINE.json's
north_star states for the dequantised floats and the quantization-loss term.  A bound relative to |sum| would be the
wrong yardstick wherever the terms cancel (Difference-penalty scale gradients alternate in sign)."""
import numpy as np

REL = 1e-5
# float32 cannot hold a relative bound next to its denormal range: an intermediate below 2^-126 is stored with an absolute
# error of up to 2^-150, and a later division by a small s^2 scales that error up (seen once in a 640-seed soak: a Difference
# scale gradient of -2.2e-38 off by 1.6e-42 = 7e-5 relative).  Results are therefore held to max(REL * sum|terms|, 2^-136).
DENORMAL_FLOOR = 2.0 ** -136


def assert_within_terms(got, ref64, abs_terms64, what="", rel=REL, extra_abs=0.0):
    """got: float32 result (array or scalar); ref64: float64 oracle value; abs_terms64: float64 sum of |terms| (same
    shape as ref64; pass None for single-term quantities: |ref64| is then the yardstick)."""
    got = np.asarray(got, np.float64).reshape(-1)
    ref = np.asarray(ref64, np.float64).reshape(-1)
    yard = np.abs(ref) if abs_terms64 is None else np.asarray(abs_terms64, np.float64).reshape(-1)
    assert got.shape == ref.shape == yard.shape, f"{what}: shapes {got.shape} {ref.shape} {yard.shape}"
    err = np.abs(got - ref)
    bound = np.maximum(rel * yard, DENORMAL_FLOOR) + extra_abs
    bad = ~(err <= bound)
    if bad.any():
        i = int(np.argmax(np.where(bad, err / np.maximum(bound, 1e-300), 0)))
        raise AssertionError(f"{what}: {int(bad.sum())} of {got.size} outside {rel:g} * sum|terms|; worst at {i}: got {got[i]!r} "
                             f"f64 {ref[i]!r} err {err[i]:.3e} bound {bound[i]:.3e} (sum|terms| {yard[i]:.3e})")


def stable_seed(*parts) -> int:
    """Seed from the test parameters that is the same in every process (hash() of a str is salted per interpreter run:
    seeds drawn from it made a different random case fail once in a few dozen runs)."""
    import zlib
    return zlib.crc32(repr(parts).encode())
