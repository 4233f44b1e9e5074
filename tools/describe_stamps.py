"""Debug aid: mean cycle-counter ticks per phase of k_describe, per wave (a wave = 8 keypoint slots).
Needs a library built with the phase stamps: make -C orb_slam2_amd/csrc CXXFLAGS+=-DORBGPU_STAMPS (and a plain
rebuild afterwards).  The stamp buffer is shared with k_fast_cells, which uses its lower half."""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
import torch  # noqa: E402
from orb_slam2_amd import api, synth  # noqa: E402
if len(sys.argv) > 1:
    api.LIB_PATH = sys.argv[1]  # a stamped build kept beside the product library

dev = torch.device("cuda", 0)
B = 512
seq = synth.sequence(640, 480, 8, seed=1, per_frame_density=True)
d = torch.from_numpy(np.ascontiguousarray(np.tile(seq, (B // 8, 1, 1)))).to(dev)
ex = api.ORBextractor(1000, 1.2, 8, 20, 7, max_width=640, max_height=480, max_batch=B)
cap = ex.max_keypoints
k = torch.zeros((B, cap, 28), dtype=torch.uint8, device=dev); ds = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
c = torch.zeros(B, dtype=torch.int32, device=dev)
L = api.load_library()


def run():
    ex.extract_batch_device(d.data_ptr(), B, 640, 480, 640, 640 * 480, k.data_ptr(), ds.data_ptr(), cap, c.data_ptr()); ex.sync()


run(); run()
n = 1 << 19
out = np.zeros((n, 8), np.uint32)
L.orbgpu_debug_fast_stamps(None, 0, 1)
run()
L.orbgpu_debug_fast_stamps(C.c_void_p(out.ctypes.data), n, 0)
out = out[n // 2:]  # the describe kernel stamps into the upper half
live = out[out[:, 4] > 0].astype(np.float64)
names = ["", "setup", "stage A moments", "stage B angle", "stage C brief"]
print("waves with work:", len(live))
print("per-wave mean ticks:", {names[i]: round(live[:, i].mean()) for i in range(1, 5)}, "total", round(live[:, 1:5].sum(axis=1).mean()))
