#!/usr/bin/env python3
"""Side measurements for the other workload configurations (not the bench.py contract line):
KITTI / EuRoC stereo frames through the host API, 4K batches device-resident, and the
100k x 100k brute-force Hamming scan.  Prints one JSON object per config."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orb_slam2_amd import api, synth  # noqa: E402

dev = torch.device("cuda", 0)


def timed(fn, n, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def stereo(name, w, h, nfeat, bf, fx, seed):
    L, R, _ = synth.stereo_pair(w, h, seed=seed)
    exL = api.ORBextractor(nfeat, 1.2, 8, 20, 7, max_width=w, max_height=h)
    exR = api.ORBextractor(nfeat, 1.2, 8, 20, 7, max_width=w, max_height=h)
    m = api.ORBmatcher()
    res = {}

    def frame():
        kL, dL = exL(L); kR, dR = exR(R)
        res["n"] = m.ComputeStereoMatches(exL, exR, kL, dL, kR, dR, bf, bf / fx)[0]
        res["kp"] = (len(kL), len(kR))
    dt = timed(frame, 20)
    # the same frame through the device-resident frame handle: both extractions side by side, row table, matching,
    # refinement and outlier cut on the device, one wait (Frame::Frame stereo, src/Frame.cc:58-116)
    fr = api.Frame(exL.max_keypoints)
    K4, b4 = [fx, fx, w / 2, h / 2], [0, 0, w, h]

    def frame2():
        res["n2"] = fr.build_stereo(exL, exR, L, R, bf, bf / fx, K4, [0, 0, 0, 0, 0], b4)[1]
    dt2 = timed(frame2, 20)
    print(json.dumps({"config": name, "ms_per_stereo_frame_host_api_incl_pcie": dt * 1e3, "ms_per_stereo_frame_frame_handle": dt2 * 1e3,
                      "keypoints": res["kp"], "stereo_matches": res["n"], "stereo_matches_frame_handle": res["n2"]}), flush=True)


def batch_4k(B=16):
    w, h = 3840, 2160
    one = synth.sequence(w, h, 2, seed=4, n_shapes=3000)
    d = torch.from_numpy(np.ascontiguousarray(np.repeat(one, B // 2, axis=0))).to(dev)
    ex = api.ORBextractor(8000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=B)
    cap = ex.max_keypoints
    k = torch.zeros((B, cap, 28), dtype=torch.uint8, device=dev); ds = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    c = torch.zeros(B, dtype=torch.int32, device=dev)

    def step():
        ex.extract_batch_device(d.data_ptr(), B, w, h, w, w * h, k.data_ptr(), ds.data_ptr(), cap, c.data_ptr())
        ex.sync()
    dt = timed(step, 5)
    ex.profile(True); step(); ms, _ = ex.stage_times()
    alg = 103190808
    print(json.dumps({"config": "4K 3840x2160, 8000 features, batch %d device-resident" % B, "frames_per_s": B / dt,
                      "algorithmic_GBps": alg * B / dt / 1e9, "keypoints_per_frame": float(c.float().mean().item()),
                      "stages_ms": [float(x) for x in ms]}), flush=True)


def brute(n=100000):
    Q = torch.from_numpy(synth.descriptors(n, 5)).to(dev); T = torch.from_numpy(synth.descriptors(n, 6)).to(dev)
    m = api.ORBmatcher()
    bi = torch.zeros(n, dtype=torch.int32, device=dev); b1 = torch.zeros_like(bi); b2 = torch.zeros_like(bi)

    def step():
        m.bruteforce_device(Q.data_ptr(), n, T.data_ptr(), n, bi.data_ptr(), b1.data_ptr(), b2.data_ptr())
        m.sync()
    dt = timed(step, 5)
    pairs = float(n) * n
    lanes_clk = 256 * 4 * 16 * 2.4e9  # CUs x SIMDs x lanes/clk x max clock: a wave64 integer instruction issues over 4 cycles
    print(json.dumps({"config": "brute force %dk x %dk x 256 bit" % (n // 1000, n // 1000), "ms": dt * 1e3,
                      "pairs_per_s": pairs / dt, "valu_util_at_16_ops_per_pair": 16 * pairs / dt / lanes_clk,
                      "hbm_GBps": (64.0 * n + 12.0 * n) / dt / 1e9}), flush=True)


def bow(B=512, nfeat=1000):
    """Frame::ComputeBoW for B frames of TUM-sized features against a 10^6-word tree (shape of the stock vocabulary)."""
    voc = synth.vocabulary_full(10, 6, seed=1)
    v = api.ORBVocabulary.from_arrays(10, 6, voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"])
    seq = synth.sequence(640, 480, 8, seed=1)
    ex = api.ORBextractor(nfeat, 1.2, 8, 20, 7, max_width=640, max_height=480, max_batch=B)
    cap = ex.max_keypoints
    d = torch.from_numpy(np.ascontiguousarray(np.tile(seq, (B // 8, 1, 1)))).to(dev)
    k = torch.zeros((B, cap, 28), dtype=torch.uint8, device=dev); ds = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    c = torch.zeros(B, dtype=torch.int32, device=dev)
    ex.extract_batch_device(d.data_ptr(), B, 640, 480, 640, 640 * 480, k.data_ptr(), ds.data_ptr(), cap, c.data_ptr())
    ex.sync()
    i32 = lambda *shape: torch.zeros(shape, dtype=torch.int32, device=dev)
    outs = [i32(B, cap), i32(B, cap), i32(B, cap), torch.zeros((B, cap), dtype=torch.float64, device=dev), i32(B), i32(B, cap),
            i32(B, cap + 1), i32(B, cap), i32(B)]

    def step():
        v.transform_batch_device(ds.data_ptr(), c.data_ptr(), B, cap, 4, [t.data_ptr() for t in outs])
        v.sync()
    dt = timed(step, 10)
    nf = float(c.float().sum().item())
    print(json.dumps({"config": "ComputeBoW, %d frames x %d features, vocabulary k=10 L=6 (%d nodes)" % (B, nfeat, len(voc["parent"])),
                      "ms": dt * 1e3, "us_per_frame": dt * 1e6 / B, "features_per_s": nf / dt,
                      "words_per_frame": float(outs[4].float().mean().item()), "nodes_per_frame": float(outs[8].float().mean().item())}),
          flush=True)


def pcie(B=512):
    """The host-pointer batch entry point: images from host memory, keypoints and descriptors back to host memory."""
    frames = synth.sequence(640, 480, B, seed=1, step=(2, 1), per_frame_density=True)
    ex = api.ORBextractor(1000, 1.2, 8, 20, 7, max_width=640, max_height=480, max_batch=B)
    res = {}

    def step():
        res["r"] = ex.extract_batch(frames)
    dt = timed(step, 5)
    print(json.dumps({"config": "TUM 640x480, 1000 features, %d frames per call through orbgpu_extract_batch (host images in, host "
                                "keypoints + descriptors out, pageable memory, incl. the Python wrapper's output copies)" % B,
                      "frames_per_s": B / dt, "ms_per_call": dt * 1e3,
                      "keypoints_per_frame": float(np.mean([len(r[0]) for r in res["r"]]))}), flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["kitti", "euroc", "4k", "bf", "bow"]
    if "kitti" in which:
        stereo("KITTI 1241x376 stereo, 2000 features/side + ComputeStereoMatches", 1241, 376, 2000, 386.1448, 718.856, 2)
    if "euroc" in which:
        stereo("EuRoC 752x480 stereo, 1200 features/side + ComputeStereoMatches", 752, 480, 1200, 47.90639384423901, 435.2046959714599, 3)
    for a in which:
        if a.startswith("4k"):  # "4k" or "4k:256" (frames per launch)
            batch_4k(int(a.split(":")[1]) if ":" in a else 16)
    if "bf" in which:
        brute()
    if "bow" in which:
        bow()
    if "pcie" in which:
        pcie()
