import sys, os
sys.path.insert(0, os.getcwd())
from orb_slam2_amd import api
api.LIB_PATH = os.path.join(os.getcwd(), "stamps", "octtrace.so")
from orb_slam2_amd import synth
import numpy as np
kind = sys.argv[1] if len(sys.argv) > 1 else "tum"
if kind == "tum":
    w, h, nf = 640, 480, 1000
elif kind == "kitti":
    w, h, nf = 1241, 376, 2000
else:
    w, h, nf = 3840, 2160, 8000
seq = synth.sequence(w, h, 2, seed=1)
ex = api.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h)
for i in range(2):
    k, d = ex(seq[i])
    print("frame", i, len(k), flush=True)
