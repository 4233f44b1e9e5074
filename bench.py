#!/usr/bin/env python3
"""Benchmark of the ORB front-end path (extract + windowed match) on MI355X.

Default workload (`--workload tum`, the configuration the metric is quoted on): one "step" = one batch of B
(default 2048) synthetic TUM-like frames (640x480, 1000 features, 8 levels, scale 1.2, FAST 20/7) per GPU, already
resident in HBM: extract all B frames, then SearchByProjection(Cur, Last)-style windowed matching of every frame
against its predecessor (the predecessor of frame 0 is the last frame of the previous step).  Metric: frames/s of the
whole job.  Every rank runs the same per-GPU workload on its own sequence (weak scaling).

`--workload 4k`: 256 frames of 3840x2160 with 8000 features, extraction only, device-resident, the 256 frames sharded
across the ranks in contiguous chunks (strong scaling).

Multi-GPU: one process per GPU, no data-path collective; torch.distributed (RCCL) only lines the ranks up (barrier)
and takes the slowest rank's time.  Under torchrun the ranks come from the environment; `--gpus N` without a torchrun
environment makes this process a launcher that starts N rank processes itself (it never touches the GPU) and prints
rank 0's line.
"""
import argparse
import json
import os
import socket
import subprocess
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
KERNEL_OF_STAGE = {"pyramid": "k_pyr_resize", "fast_cells": "k_fast_cells", "octree": "k_octree_batch", "blur": "k_blur",
                   "describe": "k_describe"}


# ---------------------------------------------------------------------------------------------------------------
# ranks
# ---------------------------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(n, argv, extra_env=None, timeout=None):
    """Start `n` rank processes of `argv` on this node (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set as torchrun
    would), wait for all, return (exit code, stdout of rank 0).  The caller must not have initialised the GPU."""
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen(argv, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, rc = "", 0
    deadline = None if timeout is None else time.time() + timeout
    try:
        out0 = procs[0].communicate(timeout=timeout)[0]
        for p in procs:
            p.wait(timeout=None if deadline is None else max(1.0, deadline - time.time()))
            rc = rc or p.returncode
    except subprocess.TimeoutExpired:
        rc = 124
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc, out0


def init_ranks(backend, device=None):
    """(rank, local_rank, world) from the environment; joins the process group when world > 1."""
    from orb_slam2_amd.dist import env_world
    rank, local_rank, world = env_world()
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    return rank, local_rank, world


def timed_steps(step, steps, warmup, sync, barrier):
    """`warmup` untimed steps, then exactly `steps` steps between barrier + sync on both sides.  Seconds on this rank."""
    for _ in range(warmup):
        step()
    sync(); barrier(); sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync(); barrier(); sync()
    return time.perf_counter() - t0


def bench_line(metric, unit, value, world, steps, warmup, wall_max, scaling, dtype, data, config, **more):
    out = {"metric": metric, "value": value, "unit": unit, "n_gpus": world, "steps": steps, "warmup": warmup,
           "ms_per_step": wall_max / steps * 1e3, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
           "dtype": dtype, "data": data, "config": config}
    out.update(more)
    return out


# ---------------------------------------------------------------------------------------------------------------
# CPU baseline: the oracle (a scalar port of the reference's algorithm) on the GPU box's host cores
# ---------------------------------------------------------------------------------------------------------------
def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(frames, n_frames=400, n_stereo=60, pool_threads=None, per_thread=100):
    """Timed like the reference's drivers (Examples/Monocular/mono_tum.cc:79-96: steady clock around the per-frame
    call, median over the frames) after 5 warm-up frames.
    (a) one thread: extract + windowed match per frame -- what the reference does for mono / RGB-D (src/Frame.cc:136,212);
    (b) two threads: left || right extraction (src/Frame.cc:78-81) + ComputeStereoMatches on KITTI-sized pairs;
    (c) for context, the mono workload frame-parallel on the host cores this job may use."""
    import threading
    from oracle import oracle as O
    from orb_slam2_amd import api, synth
    O.build()
    oe = O.Extractor(NFEAT, SCALE, NLEVELS, INI_TH, MIN_TH)
    sc = O.tables(oe.p)["scale"]
    bounds = np.array([0, 0, W, H], np.float32)

    def one(ex, prev, f):
        cur = ex.extract(f)
        q = api.shift_queries(prev[0], -STEP[0], -STEP[1], TH_PROJ, sc, bounds)
        O.search_by_projection(1, q.view(O.WQUERY_DTYPE), prev[1], cur[0], cur[1], bounds, th_dist=100, nnratio=0.9, checkOri=True)
        return cur

    t_start = time.perf_counter()
    prev = oe.extract(frames[0])
    per = []
    for k in range(1, n_frames + 6):
        t0 = time.perf_counter()
        prev = one(oe, prev, frames[k % len(frames)])
        if k > 5:
            per.append(time.perf_counter() - t0)
    per = np.asarray(per)
    res = {"value": 1.0 / float(np.median(per)), "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": "%d frames of the same sequence after 5 warm-ups, extract + windowed match per frame on one thread, "
                     "median per-frame time %.2f ms (mean %.2f ms)" % (len(per), np.median(per) * 1e3, per.mean() * 1e3),
           "cpu_model": cpu_model(), "host_cores_visible": os.cpu_count()}

    # (b) stereo, two threads
    L, R, _ = synth.stereo_pair(1241, 376, seed=2)
    exL, exR = O.Extractor(2000, SCALE, NLEVELS, INI_TH, MIN_TH), O.Extractor(2000, SCALE, NLEVELS, INI_TH, MIN_TH)
    tabs = O.tables(exL.p)
    per = []
    for k in range(n_stereo + 3):
        t0 = time.perf_counter()
        box = {}
        th = threading.Thread(target=lambda: box.__setitem__("R", exR.extract(R)))
        th.start()
        kL, dL = exL.extract(L)
        th.join()
        kR, dR = box["R"]
        O.stereo_matches(kL, dL, kR, dR, [exL.level(l) for l in range(NLEVELS)], [exR.level(l) for l in range(NLEVELS)],
                         tabs["scale"], tabs["inv_scale"], 386.1448, 386.1448 / 718.856)
        if k >= 3:
            per.append(time.perf_counter() - t0)
    per = np.asarray(per)
    res["stereo_2_threads"] = {"value": 1.0 / float(np.median(per)), "unit": "stereo frames/s", "cores": 2,
                               "sample": "%d KITTI-sized pairs (1241x376, 2000 features per eye): left || right extraction on two "
                                         "threads + ComputeStereoMatches, median %.2f ms" % (len(per), np.median(per) * 1e3)}

    # (c) all cores this job may use (the box gives one GPU's share of the host: OMP_NUM_THREADS)
    nth = pool_threads or max(1, min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16") or 16)))
    exs = [O.Extractor(NFEAT, SCALE, NLEVELS, INI_TH, MIN_TH) for _ in range(nth)]

    def worker(t):
        p = exs[t].extract(frames[t % len(frames)])
        for k in range(per_thread):
            p = one(exs[t], p, frames[(t * per_thread + k + 1) % len(frames)])
    ths = [threading.Thread(target=worker, args=(t,)) for t in range(nth)]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    dt = time.perf_counter() - t0
    res["all_cores"] = {"value": nth * per_thread / dt, "unit": "frames/s", "cores": nth,
                        "sample": "%d threads x %d frames, frame-parallel, %.1f s" % (nth, per_thread, dt)}
    res["seconds_total"] = time.perf_counter() - t_start
    return res


# ---------------------------------------------------------------------------------------------------------------
# helpers for the roofline block
# ---------------------------------------------------------------------------------------------------------------
def level_pixels(w, h):
    tot, sizes = 0, []
    f = np.float32(1.0)
    for _ in range(NLEVELS):
        inv = np.float32(1.0) / f
        lw, lh = int(np.rint(np.float32(w) * inv)), int(np.rint(np.float32(h) * inv))
        sizes.append((lw, lh))
        tot += lw * lh
        f = np.float32(f * np.float32(SCALE))
    return tot, sizes


def committed_counters(kernel, launch_frames):
    """HBM traffic and vector-instruction counts of `kernel` from the committed rocprofv3 PMC passes (builder-run on a
    one-stream schedule with `launch_frames` frames per launch; counters cannot be collected inside a timed run)."""
    traffic = valu = None
    src = []
    for tag in ("r02", "r01"):
        p = os.path.join(ROOT, "profiles", "%s_pmc_traffic.json" % tag)
        if traffic is None and os.path.exists(p):
            try:
                pm = json.load(open(p))
                if pm.get("batch") == launch_frames and kernel in pm["kernels"]:
                    kk = pm["kernels"][kernel]
                    traffic = (kk["FETCH_SIZE_KB_per_launch"] * kk.get("fetch_corr", 1.0) + kk["WRITE_SIZE_KB_per_launch"]) * 1024
                    src.append("profiles/%s_pmc_traffic.json" % tag)
            except Exception:
                pass
        p = os.path.join(ROOT, "profiles", "%s_sq_counters.txt" % tag)
        if valu is None and os.path.exists(p) and launch_frames == 512:
            for line in open(p):
                f = line.split()
                if f and f[0] == kernel:
                    valu = float(f[2])
                    src.append("profiles/%s_sq_counters.txt" % tag)
    return traffic, valu, src


def measured_issue_peak():
    """Chip-wide wave-instructions per second from the committed issue-rate microbenchmark (tools/valu_issue.hip):
    the mean over the packed-integer opcodes the FAST kernel is made of, at 8 waves per SIMD."""
    p = os.path.join(ROOT, "profiles", "r02_valu_issue.json")
    if not os.path.exists(p):
        return None, None
    try:
        j = json.load(open(p))
        ops = ["v_pk_min_i16", "v_pk_max_u16", "v_perm_b32", "v_alignbyte_b32", "v_xor_b32", "v_min_u32"]
        v = [j["ops"][o]["waves_per_simd_8"]["chip_Gwave_instr_per_s"] for o in ops if o in j["ops"]]
        return (float(np.mean(v)) * 1e9, "profiles/r02_valu_issue.json") if v else (None, None)
    except Exception:
        return None, None


def cpp_latency(frames):
    """operator() and a stack ORBmatcher through the C++ classes (examples/mono_frontend, a child process)."""
    exe = os.path.join(ROOT, "examples", "mono_frontend")
    if not os.path.exists(exe):
        return None
    import tempfile
    with tempfile.NamedTemporaryFile(suffix=".raw", delete=False) as f:
        f.write(np.array([W, H], np.int32).tobytes())
        f.write(np.ascontiguousarray(frames[:2]).tobytes())
        path = f.name
    try:
        r = subprocess.run([exe, path], capture_output=True, text=True, timeout=120)
        out = None
        for line in r.stdout.splitlines():
            t = line.split()
            if t and t[0] == "latency":
                out = {"extract_ms": float(t[2]), "stack_matcher_search_ms": float(t[4]), "matcher_ctor_us": float(t[6]),
                       "note": "C++ ORBextractor::operator() and an ORBmatcher built on the stack + SearchForInitialization per "
                               "frame (include/ORBextractor.h, include/ORBmatcher.h), host pointers in and out, median of 50; "
                               "device_frame_*: the same two steps with DeviceFrame (include/FrameDevice.h: extraction + "
                               "undistortion + grid kept on the GPU, mvKeysUn fetched; the search reads both frames there)"}
            if t and t[0] == "device_frames" and out is not None:
                out["device_frame_build_ms"] = float(t[2])
                out["device_frame_matcher_search_ms"] = float(t[4])
        return out
    except Exception:
        return None
    finally:
        os.unlink(path)
    return None


# ---------------------------------------------------------------------------------------------------------------
# workloads
# ---------------------------------------------------------------------------------------------------------------
def run_tum(args, rank, local_rank, world, torch, dev, barrier):
    from orb_slam2_amd import api, synth
    from orb_slam2_amd.dist import max_over_ranks
    B = args.batch
    frames = synth.sequence(W, H, B, seed=1 + rank, step=STEP, per_frame_density=True)
    d_imgs = torch.from_numpy(frames).to(dev)

    # S independent sub-pipelines (own extractor + matcher handle, own HIP stream), each taking B/S consecutive
    # frames of the batch: the latency-bound kernels of one overlap the ALU-bound ones of another
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

    def sync():
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    sync()
    ex.profile(True)  # HIP events between the stages on the stream the kernels are launched on
    wall = timed_steps(step, args.steps, 0, sync, barrier)
    stage_ms, _ = ex.stage_times()
    ex.profile(False)
    wall_max = max_over_ranks(wall, dev if world > 1 else None)

    # the same launches of sub-pipeline 0 with nothing else on the chip (after the timed region): in the timed region
    # the event intervals include the other sub-pipelines' kernels and the side-stream blur, this pass tells the stages apart
    ex.profile(2)  # blur kept on the main stream: every stage alone
    for _ in range(3):
        with torch.cuda.stream(stream):
            ex.extract_batch_device(d_imgs[0].data_ptr(), Bs, W, H, W, W * H, d_kps[1].data_ptr(), d_desc[1].data_ptr(), cap,
                                    d_cnt[1:].data_ptr())
    sync()
    iso_ms, _ = ex.stage_times()
    ex.profile(False)

    if rank != 0:
        return None
    wstats = [mt.window_stats(p) for p in range(0, Bs, max(1, Bs // 8))]
    kp_mean = float(d_cnt[1:].float().mean().item())
    nm_mean = float(d_nm.float().mean().item())
    P, _ = level_pixels(W, H)
    names = ["pyramid", "fast_cells", "octree", "blur", "describe"]
    alg = {"pyramid": W * H + P, "fast_cells": P, "octree": 0, "blur": 2 * P, "describe": int(kp_mean * 60)}
    # dominant kernel = the single-launch stage with the longest isolated time (the pyramid stage is a chain of launches)
    dom = 1 + int(np.argmax(iso_ms[1:]))
    dom_name = names[dom]
    dom_bytes = alg[dom_name] * Bs if alg[dom_name] else 4 * 4000 * Bs
    achieved = dom_bytes / (float(stage_ms[dom]) * 1e-3) / 1e9
    traffic, insts, src = committed_counters(KERNEL_OF_STAGE[dom_name], Bs)
    valu = None
    if insts:
        peak, peak_src = measured_issue_peak()
        rate = insts / (float(iso_ms[dom]) * 1e-3)
        valu = {"wave_instructions_per_launch": insts, "achieved_Ginstr_per_s": rate / 1e9,
                "peak_Ginstr_per_s": None if peak is None else peak / 1e9, "frac": None if peak is None else rate / peak,
                "peak_source": peak_src}
    frame_alg = W * H + (P - W * H) + P + 2 * P + int(kp_mean * 60)
    out = bench_line(
        "frames/sec ORB extract+match at 640x480/1000 kp", "frames/s", world * B * args.steps / wall_max, world, args.steps,
        args.warmup, wall_max, "weak", "u8", "synthetic",
        {"workload": "TUM mono 640x480, 1000 features, 8 levels, scale 1.2, FAST 20/7: extract + SearchByProjection(Cur,Last) "
                     "th=15 against the previous frame",
         "batch_per_gpu": B, "streams": S, "keypoints_per_frame": kp_mean, "matches_per_frame": nm_mean,
         "parallelism": "every rank its own sequence on its own GPU (%d), no collective" % world},
        roofline={"bound": "hbm", "kernel": KERNEL_OF_STAGE[dom_name], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                  "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "algorithmic_bytes_per_launch": dom_bytes,
                  "launch_ms": float(stage_ms[dom]), "launch_ms_isolated": float(iso_ms[dom]),
                  "achieved_isolated": dom_bytes / (float(iso_ms[dom]) * 1e-3) / 1e9, "valu_issue": valu,
                  "source": {"achieved": "HIP events in this run", "traffic_and_instruction_counts": src + ["builder-run rocprofv3 --pmc passes"]},
                  "note": "integer byte kernel: its L2 fetches stay within about twice the algorithmic volume, the time goes into vector "
                          "instructions (valu_issue: instructions per launch over the isolated launch time, against the measured issue rate)"},
        stages_ms_per_step={n: float(v) for n, v in zip(names, stage_ms)},
        stages_ms_isolated={n: float(v) for n, v in zip(names, iso_ms)},
        extract_algorithmic_GBps=frame_alg * Bs / (float(iso_ms.sum()) * 1e-3) / 1e9,
        step_algorithmic_GBps=frame_alg * world * B * args.steps / wall_max / 1e9,
        window_resolver_iterations_rescans=wstats)

    if world == 1 and not args.no_latency:
        # what a host caller sees: (1) one frame through the host-pointer C ABI; (2) a batch of host images in, host
        # keypoints + descriptors out (PCIe both ways) -- never the reported `value`
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
        out["single_frame_latency"] = {"extract_ms": (t2 - t1) / 50 * 1e3, "match_ms": (t3 - t2) / 50 * 1e3,
                                       "note": "one frame through the host-pointer C ABI incl. PCIe both ways and the Python call",
                                       "cpp": cpp_latency(frames)}
        nb = min(512, B)
        exb = api.ORBextractor(NFEAT, SCALE, NLEVELS, INI_TH, MIN_TH, device=local_rank, max_width=W, max_height=H, max_batch=nb)
        capb = exb.max_keypoints
        cnt = np.zeros(nb, np.int32)
        pcie = {"frames_per_call": nb,
                "note": "orbgpu_extract_batch, extraction only: host images in, keypoints + descriptors out to host memory, chunks "
                        "of the batch overlapped (upload || kernels || download); pinned = page-locked caller memory used in place, "
                        "pageable = staged through the handle's pinned buffers by host threads"}
        for kind in ("pinned", "pageable"):
            if kind == "pinned":
                img = api.pinned_empty((nb, H, W), np.uint8); img[...] = frames[:nb]
                k = api.pinned_empty((nb, capb), api.KEYPOINT_DTYPE); d = api.pinned_empty((nb, capb, 32), np.uint8)
            else:
                img = np.ascontiguousarray(frames[:nb]); k = np.zeros((nb, capb), api.KEYPOINT_DTYPE); d = np.zeros((nb, capb, 32), np.uint8)
            exb.extract_batch_into(img, k, d, cnt)
            t4 = time.perf_counter()
            for _ in range(5):
                exb.extract_batch_into(img, k, d, cnt)
            pcie[kind] = 5 * nb / (time.perf_counter() - t4)
        pcie["value"] = pcie["pinned"]
        out["pcie_inclusive_frames_per_s"] = pcie
    return out


def run_4k(args, rank, local_rank, world, torch, dev, barrier):
    """256 frames 3840x2160 / 8000 features, extraction only, sharded over the ranks in contiguous chunks."""
    from orb_slam2_amd import api, synth
    from orb_slam2_amd.dist import max_over_ranks, shard_frames
    w, h, nfeat, total = 3840, 2160, 8000, args.frames_4k
    mine = shard_frames(total, world, rank)
    n = len(mine)
    distinct = synth.sequence(w, h, 4, seed=4, n_shapes=3000)  # four distinct frames, cycled by global frame index
    idx = torch.tensor([g % 4 for g in mine], dtype=torch.long)
    d_imgs = torch.from_numpy(distinct).to(dev)[idx.to(dev)].contiguous()
    per_launch = min(n, args.launch_4k)
    ex = api.ORBextractor(nfeat, SCALE, NLEVELS, INI_TH, MIN_TH, device=local_rank, max_width=w, max_height=h, max_batch=per_launch)
    cap = ex.max_keypoints
    d_k = torch.zeros((n, cap, 28), dtype=torch.uint8, device=dev)
    d_d = torch.zeros((n, cap, 32), dtype=torch.uint8, device=dev)
    d_c = torch.zeros(n, dtype=torch.int32, device=dev)

    def step():
        for lo in range(0, n, per_launch):
            m = min(per_launch, n - lo)
            ex.extract_batch_device(d_imgs[lo].data_ptr(), m, w, h, w, w * h, d_k[lo].data_ptr(), d_d[lo].data_ptr(), cap,
                                    d_c[lo:].data_ptr())

    def sync():
        ex.sync()
        torch.cuda.synchronize(dev)

    wall = timed_steps(step, args.steps, args.warmup, sync, barrier)
    wall_max = max_over_ranks(wall, dev if world > 1 else None)
    ex.profile(2)
    step(); sync()
    iso_ms, _ = ex.stage_times()
    ex.profile(False)
    if rank != 0:
        return None
    kp_mean = float(d_c.float().mean().item())
    P, _ = level_pixels(w, h)
    frame_alg = w * h + (P - w * h) + P + 2 * P + int(kp_mean * 60)
    names = ["pyramid", "fast_cells", "octree", "blur", "describe"]
    dom = 1 + int(np.argmax(iso_ms[1:]))
    alg = {"fast_cells": P, "octree": 4 * 32000, "blur": 2 * P, "describe": int(kp_mean * 60)}[names[dom]]
    m_last = n - (n - 1) // per_launch * per_launch  # frames of the launch the isolated pass timed last
    achieved = alg * m_last / (float(iso_ms[dom]) * 1e-3) / 1e9
    return bench_line(
        "frames/sec ORB extract at 3840x2160/8000 kp", "frames/s", total * args.steps / wall_max, world, args.steps, args.warmup,
        wall_max, "strong", "u8", "synthetic (4 distinct frames cycled)",
        {"workload": "256-frame batch 3840x2160, 8000 features, 8 levels: extraction, device-resident, frames sharded in contiguous "
                     "chunks across the ranks",
         "frames_total": total, "frames_this_rank": n, "frames_per_launch": per_launch, "keypoints_per_frame": kp_mean,
         "parallelism": "%d rank(s), no collective" % world},
        roofline={"bound": "hbm", "kernel": KERNEL_OF_STAGE[names[dom]], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                  "frac": achieved / HBM_PEAK_GBS, "traffic": None, "launch_ms_isolated": float(iso_ms[dom]),
                  "source": {"achieved": "HIP events in this run (launches timed alone after the timed region)"}},
        stages_ms_isolated={nm: float(v) for nm, v in zip(names, iso_ms)},
        step_algorithmic_GBps=frame_alg * total * args.steps / wall_max / 1e9)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["tum", "4k"], default="tum")
    ap.add_argument("--batch", type=int, default=2048, help="tum: frames per step and GPU")
    ap.add_argument("--streams", type=int, default=4, help="tum: independent sub-pipelines (HIP streams) the batch is split over")
    ap.add_argument("--frames-4k", type=int, default=256, help="4k: frames of the whole job per step")
    ap.add_argument("--launch-4k", type=int, default=256, help="4k: frames per launch (at most the rank's share)")
    ap.add_argument("--no-latency", action="store_true", help="skip the host-side latency / PCIe probes (profiling runs)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # launcher: this process starts the ranks and relays rank 0's line; it must not touch the GPU itself
        rc, out0 = spawn_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:])
        sys.stdout.write(out0)
        sys.stdout.flush()
        sys.exit(rc)

    import torch
    import torch.distributed as dist
    from orb_slam2_amd.dist import env_world
    _, local_rank, _ = env_world()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rank, local_rank, world = init_ranks("nccl", dev)

    def barrier():
        if world > 1:
            dist.barrier()

    run = run_tum if args.workload == "tum" else run_4k
    out = run(args, rank, local_rank, world, torch, dev, barrier)
    if rank == 0:
        if world == 1 and args.workload == "tum":
            if args.no_cpu:
                out["cpu_baseline"] = None
            else:
                from orb_slam2_amd import synth
                out["cpu_baseline"] = cpu_baseline(synth.sequence(W, H, 64, seed=1, step=STEP, per_frame_density=True))
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
