#!/usr/bin/env python3
"""KITTI-sized stereo frames through the device-resident frame handle, for a rocprofv3 kernel trace:
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stereo -- python3 tools/stereo_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orb_slam2_amd import api, synth  # noqa: E402

w, h, nfeat, bf, fx = 1241, 376, 2000, 386.1448, 718.856
L, R, _ = synth.stereo_pair(w, h, seed=2)
exL = api.ORBextractor(nfeat, 1.2, 8, 20, 7, max_width=w, max_height=h)
exR = api.ORBextractor(nfeat, 1.2, 8, 20, 7, max_width=w, max_height=h)
fr = api.Frame(exL.max_keypoints)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    n, nv = fr.build_stereo(exL, exR, L, R, bf, bf / fx, [fx, fx, w / 2, h / 2], [0, 0, 0, 0, 0], [0, 0, w, h])
print(n, nv)
