#!/usr/bin/env python3
"""Brute-force Hamming scan: the vector-unit kernel against the matrix-core kernel at several sizes (device-resident)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orb_slam2_amd import api, synth  # noqa: E402

# --lib <path>: measure another build of the library (tools/build_variant.sh)
argv = sys.argv[1:]
if argv and argv[0] == "--lib":
    api.LIB_PATH = os.path.abspath(argv[1])
    argv = argv[2:]
dev = torch.device("cuda", 0)
m = api.ORBmatcher()
for n in [int(a) for a in argv] or [1000, 4000, 20000, 100000]:
    Q = torch.from_numpy(synth.descriptors(n, 5)).to(dev); T = torch.from_numpy(synth.descriptors(n, 6)).to(dev)
    out = [torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(6)]
    res = {"n": n}
    for which, name in ((1, "vector_unit"), (2, "matrix_cores")):
        m.set_bruteforce_kernel(which)
        o = out[:3] if which == 1 else out[3:]

        def step():
            m.bruteforce_device(Q.data_ptr(), n, T.data_ptr(), n, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr())
        for _ in range(3):
            step()
        m.sync()
        reps = 20 if n <= 20000 else 5
        t0 = time.perf_counter()
        for _ in range(reps):
            step()
        m.sync()
        dt = (time.perf_counter() - t0) / reps
        res[name + "_ms"] = dt * 1e3
        res[name + "_pairs_per_s"] = float(n) * n / dt
    res["equal"] = all(bool(torch.equal(a, b)) for a, b in zip(out[:3], out[3:]))
    print(json.dumps(res), flush=True)
