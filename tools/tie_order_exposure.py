#!/usr/bin/env python3
"""How much of the extraction depends on the order in which DistributeOctTree takes quadtree nodes of EQUAL size (the
reference sorts (size, node pointer) pairs, src/ORBextractor.cc:906, i.e. by heap address; this library and its oracle
take the latest-created node first).  Runs the CPU oracle with both extreme orders on the benchmark's frames and
prints, per level: how often the selected keypoint SET differs, and how often only the OUTPUT ORDER differs.
CPU only (test infrastructure)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402
from orb_slam2_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
O.build()
for name, (w, h, nfeat, seed) in {"TUM 640x480/1000": (640, 480, 1000, 1), "EuRoC 752x480/1200": (752, 480, 1200, 3),
                                  "KITTI 1241x376/2000": (1241, 376, 2000, 2)}.items():
    seq = synth.sequence(w, h, n, seed=seed, per_frame_density=True)
    oe = O.Extractor(nfeat)
    lv_total = lv_set = lv_order = kp_total = kp_dep = frames_straddle = 0
    for img in seq:
        O.set_tie_rule(0)
        k0, _ = oe.extract(img)
        frames_straddle += 1 if oe.tie_straddles() else 0
        O.set_tie_rule(1)
        try:
            k1, _ = oe.extract(img)
        finally:
            O.set_tie_rule(0)
        a_all = set(zip(k0["x"].tolist(), k0["y"].tolist(), k0["octave"].tolist()))
        b_all = set(zip(k1["x"].tolist(), k1["y"].tolist(), k1["octave"].tolist()))
        kp_total += len(a_all); kp_dep += len(a_all - b_all)
        for lv in range(8):
            a = list(zip(k0["x"][k0["octave"] == lv].tolist(), k0["y"][k0["octave"] == lv].tolist()))
            b = list(zip(k1["x"][k1["octave"] == lv].tolist(), k1["y"][k1["octave"] == lv].tolist()))
            lv_total += 1
            if set(a) != set(b):
                lv_set += 1
            elif a != b:
                lv_order += 1
    print("%s, %d frames: a size tie straddles the cut on %d frames; keypoints that change with the tie order %d of %d (%.2f %%); "
          "levels with another keypoint set %d of %d, with the same set in another output order %d"
          % (name, n, frames_straddle, kp_dep, kp_total, 100.0 * kp_dep / max(kp_total, 1), lv_set, lv_total, lv_order), flush=True)
