#!/usr/bin/env python3
"""Benchmark of the ORB front-end path (extract + windowed match) on MI355X.

One "step" = one batch of B (default 2048) synthetic TUM-like frames (640x480, 1000 features,
8 levels, scale 1.2, FAST 20/7), already resident in HBM: extract all B frames,
then SearchByProjection(Cur, Last)-style windowed matching of every frame
against its predecessor (the predecessor of frame 0 is the last frame of the
previous step).  Metric: frames/s, whole job (all ranks).

Multi-GPU: one process per GPU, every rank runs the same per-GPU workload on its
own frames (weak scaling); there is no data-path collective, torch.distributed
is used only for the barrier and the max-over-ranks of the elapsed time.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# multi-process GPU work on this driver needs dmabuf IPC (RCCL fails with hipIpcGetMemHandle errors otherwise)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

W, H, NFEAT, NLEVELS, SCALE, INI_TH, MIN_TH = 640, 480, 1000, 8, 1.2, 20, 7
STEP = (2, 1)   # per-frame image motion of the synthetic sequence
TH_PROJ = 15.0  # Tracking.cc:1060-1066 (monocular window)
HBM_PEAK_GBS = 8000.0


def level_pixels(w, h):
    sc, tot, sizes = 1.0, 0, []
    f = np.float32(1.0)
    for l in range(NLEVELS):
        inv = np.float32(1.0) / f
        lw, lh = int(np.rint(np.float32(w) * inv)), int(np.rint(np.float32(h) * inv))
        sizes.append((lw, lh))
        tot += lw * lh
        f = np.float32(f * np.float32(SCALE))
    return tot, sizes


def cpu_baseline(frames, n_frames):
    """Oracle (scalar CPU port) timed on one host core over a bounded sample."""
    from oracle import oracle as O
    O.build()
    oe = O.Extractor(NFEAT, SCALE, NLEVELS, INI_TH, MIN_TH)
    sc = O.tables(oe.p)["scale"]
    from orb_slam2_amd import api
    bounds = np.array([0, 0, W, H], np.float32)
    prev = oe.extract(frames[0])
    t0 = time.perf_counter()
    for k in range(1, n_frames + 1):
        f = frames[k % len(frames)]
        cur = oe.extract(f)
        q = api.shift_queries(prev[0], -STEP[0], -STEP[1], TH_PROJ, sc, bounds)
        O.search_by_projection(1, q.view(O.WQUERY_DTYPE), prev[1], cur[0], cur[1], bounds, th_dist=100, nnratio=0.9,
                               checkOri=True)
        prev = cur
    dt = time.perf_counter() - t0
    return n_frames / dt, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=2048, help="frames per step and GPU")
    ap.add_argument("--streams", type=int, default=4, help="independent sub-pipelines (HIP streams) the batch is split over")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-frame latency probe (profiling runs)")
    ap.add_argument("--cpu-frames", type=int, default=800, help="frames of the CPU baseline sample (0 = skip)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from orb_slam2_amd import api, synth

    from orb_slam2_amd.dist import env_world, max_over_ranks
    rank, local_rank, world = env_world()
    # rehearsal of the multi-rank path on a one-GPU box: ORBGPU_BENCH_REHEARSAL=1 puts every rank on device 0 and
    # lines the ranks up over gloo (RCCL refuses two ranks on one device); never set in a measured run
    rehearsal = os.environ.get("ORBGPU_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    B = args.batch
    frames = synth.sequence(W, H, B, seed=1 + rank, step=STEP, per_frame_density=True)
    d_imgs = torch.from_numpy(frames).to(dev)

    # S independent sub-pipelines (own extractor + matcher handle, own HIP stream), each taking B/S
    # consecutive frames of the batch: the latency-bound kernels of one overlap the ALU-bound ones of another
    S = args.streams
    assert B % S == 0, "--batch must be a multiple of --streams"
    Bs = B // S
    exs = [api.ORBextractor(NFEAT, SCALE, NLEVELS, INI_TH, MIN_TH, device=local_rank, max_width=W, max_height=H, max_batch=Bs)
           for _ in range(S)]
    mts = [api.ORBmatcher(0.9, True, device=local_rank) for _ in range(S)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    for e, m, st in zip(exs, mts, streams):
        e.set_stream(st.cuda_stream)
        m.set_stream(st.cuda_stream)
    ex, mt, stream = exs[0], mts[0], streams[0]
    ev_extracted = [torch.cuda.Event() for _ in range(S)]
    ev_matched = [torch.cuda.Event() for _ in range(S)]
    ev_carried = torch.cuda.Event()
    cap = ex.max_keypoints
    scale = ex.GetScaleFactors()
    bounds = np.array([0, 0, W, H], np.float32)

    # slot 0 = last frame of the previous step, slots 1..B = this step's frames
    d_kps = torch.zeros((B + 1, cap, 28), dtype=torch.uint8, device=dev)
    d_desc = torch.zeros((B + 1, cap, 32), dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(B + 1, dtype=torch.int32, device=dev)
    d_q = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_assign = torch.zeros((B, cap), dtype=torch.int32, device=dev)
    d_nm = torch.zeros(B, dtype=torch.int32, device=dev)

    def step():
        for k in range(S):
            lo = k * Bs  # frames lo .. lo+Bs-1 of the batch live in slots lo+1 .. lo+Bs
            with torch.cuda.stream(streams[k]):
                # slot lo+1.. are still read by the previous step's matching on the next stream (its slot "lo+Bs")
                streams[k].wait_event(ev_matched[k + 1] if k + 1 < S else ev_carried)
                exs[k].extract_batch_device(d_imgs[lo].data_ptr(), Bs, W, H, W, W * H, d_kps[lo + 1].data_ptr(),
                                            d_desc[lo + 1].data_ptr(), cap, d_cnt[lo + 1:].data_ptr())
                ev_extracted[k].record(streams[k])
                # the predecessor of this sub-batch's first frame comes from the previous sub-batch (slot lo)
                if k > 0:
                    streams[k].wait_event(ev_extracted[k - 1])
                mts[k].make_shift_queries_device(Bs, d_kps[lo].data_ptr(), d_cnt[lo:].data_ptr(), cap, -float(STEP[0]),
                                                 -float(STEP[1]), TH_PROJ, scale, bounds, d_q[lo].data_ptr())
                mts[k].search_by_projection_batch_device(api.MODE_LAST_FRAME, Bs, d_q[lo].data_ptr(), d_desc[lo].data_ptr(),
                                                         d_cnt[lo:].data_ptr(), cap, d_kps[lo + 1].data_ptr(),
                                                         d_desc[lo + 1].data_ptr(), d_cnt[lo + 1:].data_ptr(), cap, bounds,
                                                         d_assign[lo].data_ptr(), d_nm[lo:].data_ptr(), th_dist=api.TH_HIGH)
                ev_matched[k].record(streams[k])
        with torch.cuda.stream(streams[0]):
            # carry the last frame into slot 0 for the next step (after sub-pipeline 0 has matched against the old slot 0)
            streams[0].wait_event(ev_extracted[S - 1])
            d_kps[0].copy_(d_kps[B], non_blocking=True)
            d_desc[0].copy_(d_desc[B], non_blocking=True)
            d_cnt[0:1].copy_(d_cnt[B:B + 1], non_blocking=True)
            ev_carried.record(streams[0])

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    ex.profile(True)
    ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        step()
    for k in range(1, S):
        stream.wait_event(ev_matched[k])
    ev1.record(stream)
    barrier()
    wall = time.perf_counter() - t0
    gpu_ms = ev0.elapsed_time(ev1)
    stage_ms, ncalls = ex.stage_times()
    ex.profile(False)

    # the same launches of sub-pipeline 0 with nothing else on the chip (after the timed region): in the timed region
    # the event times include the other sub-pipelines' kernels and the side-stream blur, this pass tells the stages apart
    ex.profile(2)  # blur kept on the main stream: every stage alone
    for _ in range(3):
        with torch.cuda.stream(stream):
            ex.extract_batch_device(d_imgs[0].data_ptr(), Bs, W, H, W, W * H, d_kps[1].data_ptr(), d_desc[1].data_ptr(), cap,
                                    d_cnt[1:].data_ptr())
    torch.cuda.synchronize(dev)
    iso_ms, _ = ex.stage_times()
    ex.profile(False)

    wall_max = max_over_ranks(wall, None if rehearsal else dev)

    wstats = [mt.window_stats(p) for p in range(0, Bs, max(1, Bs // 8))]
    kp_mean = float(d_cnt[1:].float().mean().item())
    nm_mean = float(d_nm.float().mean().item())

    latency = None
    if rank == 0 and not args.no_latency:
        # single-frame latency through the host-pointer API (H2D image, D2H features, sync): not part of `value`
        ex1 = api.ORBextractor(NFEAT, SCALE, NLEVELS, INI_TH, MIN_TH, device=local_rank, max_width=W, max_height=H)
        mt1 = api.ORBmatcher(0.9, True, device=local_rank)
        k0, d0 = ex1(frames[0]); k1, d1 = ex1(frames[1])
        q1 = api.shift_queries(k0, -STEP[0], -STEP[1], TH_PROJ, scale, bounds)
        for _ in range(5):
            ex1(frames[1]); mt1.SearchByProjection(api.MODE_LAST_FRAME, q1, d0, k1, d1, bounds)
        t1 = time.perf_counter()
        for _ in range(50):
            ex1(frames[1])
        t2 = time.perf_counter()
        for _ in range(50):
            mt1.SearchByProjection(api.MODE_LAST_FRAME, q1, d0, k1, d1, bounds)
        t3 = time.perf_counter()
        latency = {"extract_ms": (t2 - t1) / 50 * 1e3, "match_ms": (t3 - t2) / 50 * 1e3,
                   "note": "one frame through the host-pointer C ABI incl. PCIe both ways and the Python call"}

    if rank == 0:
        P, sizes = level_pixels(W, H)
        # algorithmic HBM bytes per frame of each extractor stage
        alg = {
            "pyramid": W * H + P,            # read the input, write every level once
            "fast_cells": P,                 # read every level once
            "octree": 0,
            "blur": 2 * P,                   # read + write every level
            "describe": int(kp_mean * 60),   # 28 B keypoint + 32 B descriptor per feature
        }
        names = ["pyramid", "fast_cells", "octree", "blur", "describe"]
        # the blur runs on a side stream beside the quadtree: main-stream time = everything but the blur
        match_ms = gpu_ms / args.steps - float(stage_ms.sum() - stage_ms[3]) if S == 1 else None
        # dominant kernel = the single-launch stage with the longest isolated time (the pyramid stage is a chain of nine launches)
        dom = 1 + int(np.argmax(iso_ms[1:]))
        dom_name = names[dom]
        # the quadtree moves O(N) bytes; price it (if dominant) by the candidate bytes it gathers
        dom_bytes = alg[dom_name] * Bs if alg[dom_name] else 4 * 4000 * Bs
        achieved = dom_bytes / (float(stage_ms[dom]) * 1e-3) / 1e9
        # HBM-side bytes per launch of the dominant kernel from the committed PMC passes (same batch size only)
        traffic = None
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            if pm.get("batch") == Bs:
                kk = pm["kernels"][{"pyramid": "k_pyr_resize", "fast_cells": "k_fast_cells", "octree": "k_octree_batch", "blur": "k_blur",
                                    "describe": "k_describe"}[dom_name]]
                traffic = (kk["FETCH_SIZE_KB_per_launch"] * kk.get("fetch_corr", 1.0) + kk["WRITE_SIZE_KB_per_launch"]) * 1024
        except Exception:
            traffic = None
        # vector-issue view of the same kernel: SQ_INSTS_VALU per launch from the committed PMC pass (same launch size
        # only) over the isolated launch time, against 1024 SIMDs x 2.4 GHz / 4 cycles per wave64 integer instruction
        valu = None
        try:
            if Bs == 512:
                kname = {"pyramid": "k_pyr_resize", "fast_cells": "k_fast_cells", "octree": "k_octree_batch", "blur": "k_blur",
                         "describe": "k_describe"}[dom_name]
                for line in open(os.path.join(ROOT, "profiles", "r01_sq_counters.txt")):
                    f = line.split()
                    if f and f[0] == kname:
                        insts = float(f[2])
                        peak = 1024 * 2.4e9 / 4
                        valu = {"wave_instructions_per_launch": insts, "achieved_Ginstr_per_s": insts / (float(iso_ms[dom]) * 1e-3) / 1e9,
                                "peak_Ginstr_per_s": peak / 1e9, "frac": insts / (float(iso_ms[dom]) * 1e-3) / peak}
        except Exception:
            valu = None
        frame_alg = W * H + (P - W * H) + P + 2 * P + int(kp_mean * 60)
        out = {
            "metric": "frames/sec ORB extract+match at 640x480/1000 kp",
            "value": world * B * args.steps / wall_max,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": wall_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": "TUM mono 640x480, 1000 features, 8 levels, scale 1.2, FAST 20/7: extract + "
                                   "SearchByProjection(Cur,Last) th=15 against the previous frame",
                       "batch_per_gpu": B, "streams": S, "keypoints_per_frame": kp_mean, "matches_per_frame": nm_mean,
                       "parallelism": "frames sharded across %d GPU(s), no collective" % world + (" (REHEARSAL: all ranks on one device)" if rehearsal else "")},
            "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": dom_bytes, "launch_ms": float(stage_ms[dom]),
                         "launch_ms_isolated": float(iso_ms[dom]), "achieved_isolated": dom_bytes / (float(iso_ms[dom]) * 1e-3) / 1e9,
                         "valu_issue": valu,
                         "note": "integer byte kernels: the binding limit is vector-ALU issue (1 wave-instr / 4 cycles / SIMD), "
                                 "see profiles/r01_sq_counters.txt; traffic = committed rocprofv3 PMC passes of this workload"},
            "stages_ms_per_step": dict({n: float(v) for n, v in zip(names, stage_ms)}, match=match_ms),
            "stages_ms_isolated": {n: float(v) for n, v in zip(names, iso_ms)},
            "extract_algorithmic_GBps": frame_alg * Bs / (float(stage_ms.sum()) * 1e-3) / 1e9,
            "gpu_ms_per_step": gpu_ms / args.steps,
            "window_resolver_iterations_rescans": wstats,
            "single_frame_latency": latency,
        }
        if args.cpu_frames > 0 and world == 1:
            fps, dt = cpu_baseline(frames, args.cpu_frames)
            out["cpu_baseline"] = {"value": fps, "unit": "frames/s", "cores": 1, "kind": "port",
                                   "sample": "%d frames of the same sequence, extract + windowed match, %.1f s" % (args.cpu_frames, dt)}
        elif world == 1:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
