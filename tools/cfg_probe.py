#!/usr/bin/env python3
"""One configuration of bench.py's `configs` block alone (for profiling runs):
usage: python3 tools/cfg_probe.py kitti_stereo|euroc_stereo|euroc_search|4k|bf [batch]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
which = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
if which == "kitti_stereo":
    r = bench.cfg_stereo(torch, dev, "KITTI", 1241, 376, 2000, 386.1448, 718.856, 2, B, False)
elif which == "euroc_stereo":
    r = bench.cfg_stereo(torch, dev, "EuRoC", 752, 480, 1200, 47.90639384423901, 435.2046959714599, 3, B, False)
elif which == "euroc_search":
    r = bench.cfg_euroc_search(torch, dev, B)
elif which == "4k":
    r = bench.cfg_4k(torch, dev, max(B, 32), [(B, 1), (32, 1), (32, 2), (32, 4)])
else:
    r = bench.cfg_bruteforce(torch, dev, 100000)
print(json.dumps(r))
