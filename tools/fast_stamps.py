"""Debug aid: mean cycle-counter ticks per phase of k_fast_cells, per cell.
Needs a library built with the phase stamps: touch the .hip files, then make -C orb_slam2_amd/csrc with the
Makefile's CXXFLAGS plus -DORBGPU_STAMPS (and a plain rebuild afterwards: the stamped kernel is slower).  Run on the GPU box: python tools/fast_stamps.py"""
import sys, ctypes as C, numpy as np
sys.path.insert(0, '.')
import torch
from orb_slam2_amd import api, synth
if len(sys.argv) > 1:
    api.LIB_PATH = sys.argv[1]  # a stamped build kept beside the product library
dev = torch.device('cuda', 0)
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
nwg = ((815 + 3) // 4) * B
out = np.zeros((nwg, 8), np.uint32)
L.orbgpu_debug_fast_stamps(None, 0, 1)
run()
L.orbgpu_debug_fast_stamps(C.c_void_p(out.ctypes.data), nwg, 0)
live = out[out[:, 6] > 0].astype(np.float64)
names = ['', 'staging', 'stage1 reject', 'expand', 'arcs', 'nms', 'emit']
print('workgroups with work:', len(live)); live = live / 4.0
print('per-WG mean ticks:', {names[i]: round(live[:, i].mean()) for i in range(1, 7)}, 'total', round(live[:, 1:7].sum(axis=1).mean()))
print('per-WG median ticks:', {names[i]: round(float(np.median(live[:, i]))) for i in range(1, 7)})
