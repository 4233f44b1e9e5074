"""The allocation-free brute-force entry point against orbgpu_hamming_bruteforce_device: bit-exact (integer
work), on two calls in a row, the second with a smaller input so that the shared fixed scratch still holds the
first call's partial results."""
import ctypes as C

import numpy as np
import pytest

from orb_slam2_amd import api
from tests.hiputil import Hip

pytestmark = pytest.mark.gpu


def _run(fn, hip, Q, T):
    nq, nt = len(Q), len(T)
    dQ, dT = hip.upload(Q), hip.upload(T if nt else np.zeros((1, 32), np.uint8))
    out = [hip.alloc(max(nq, 1) * 4) for _ in range(3)]
    fn(dQ.value, nq, dT.value, nt, out[0].value, out[1].value, out[2].value)
    return [hip.download(o, nq, np.int32) for o in out]


def test_noalloc_bruteforce_matches_allocating_entry_point():
    rng = np.random.default_rng(77)
    m = api.ORBmatcher(0.9, True)
    hip = Hip()
    try:
        # (nq, nt): several train chunks; then smaller (stale partial results of the first call stay in the scratch); a large query set; no train set
        for nq, nt in ((5000, 30000), (777, 1300), (300000, 4096), (64, 0)):
            Q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
            T = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
            if nt:
                T[rng.integers(0, nt, nq // 3)] = Q[rng.integers(0, nq, nq // 3)]  # exact hits and ties
            ref = _run(m.bruteforce_device, hip, Q, T)
            got = _run(m.bruteforce_device_noalloc, hip, Q, T)
            for r, g, name in zip(ref, got, ("best_idx", "best_d", "second_d")):
                assert np.array_equal(r, g), (nq, nt, name)
            if nt:
                assert (ref[0] >= 0).all() and (ref[1] <= ref[2]).all()
            else:
                assert (ref[0] == -1).all() and (ref[1] == 256).all()
            hip.free()
        # beyond the fixed scratch: refused with a clear error, nothing launched
        big = 3 << 20
        with pytest.raises(api.OrbGpuError, match="scratch"):
            m.bruteforce_device_noalloc(1, big, 1, 16, 1, 1, 1)
    finally:
        hip.free()
        m.close()
