#!/usr/bin/env python3
"""Single-frame latency through the host-pointer ABI (PCIe both ways): extract, then match against the previous frame."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orb_slam2_amd import api, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
if os.environ.get("ORBGPU_LIB_OVERRIDE"):  # a variant build (tools/build_variant.sh)
    api.LIB_PATH = os.path.abspath(os.environ["ORBGPU_LIB_OVERRIDE"])
seq = synth.sequence(640, 480, 4, seed=1)
ex = api.ORBextractor(1000, 1.2, 8, 20, 7, max_width=640, max_height=480)
m = api.ORBmatcher(0.9, True)
b = np.array([0, 0, 640, 480], np.float32)
res = [ex(f) for f in seq]
t0 = time.perf_counter()
for i in range(n):
    ex(seq[i & 3])
t1 = time.perf_counter()
q = api.shift_queries(res[0][0], -2.0, -1.0, 15.0, ex.GetScaleFactors(), b)
for i in range(n):
    m.SearchByProjection(api.MODE_LAST_FRAME, q, res[0][1], res[1][0], res[1][1], b)
t2 = time.perf_counter()
print("extract %.1f us/frame, match %.1f us/frame" % ((t1 - t0) / n * 1e6, (t2 - t1) / n * 1e6))
