#!/usr/bin/env python3
"""Benchmark of the ORB front-end path (extract + windowed match) on MI355X.

Default workload (`--workload tum`, the configuration the metric is quoted on): one "step" = `--sub-batches` (8)
sub-batches of B (default 2048) synthetic TUM-like frames (640x480, 1000 features, 8 levels, scale 1.2, FAST 20/7)
per GPU, all resident in HBM and all distinct: extract every frame, then match it against its predecessor with
SearchByProjection(Cur, Last) (th = 15, src/Tracking.cc:1060-1066) and SearchForInitialization (window 100,
src/Tracking.cc:738-748).  Metric: frames/s of the whole job.  Every rank runs the same per-GPU workload on its own
sequence (weak scaling).

On one GPU the line also carries
  * `parity_in_run`: frames of the timed batch compared with the CPU oracle's results on the same images;
  * `cpu_baseline`: the oracle timed on the host cores on the same workload;
  * `configs`: the other benchmark configurations measured in the same run, each with its own roofline and parity
    check -- KITTI and EuRoC stereo frames (two extractions + ComputeStereoMatches), the three Tracking-side
    SearchByProjection variants between consecutive EuRoC frames, 3840x2160 batches and the 100k x 100k brute force.

`--workload 4k`: 256 frames of 3840x2160 with 8000 features, extraction only, device-resident, the 256 frames sharded
across the ranks in contiguous chunks (strong scaling).

Multi-GPU: one process per GPU, no data-path collective; torch.distributed only lines the ranks up (barrier) and
takes the slowest rank's time (RCCL, or gloo on CPU tensors if RCCL cannot start).  Under torchrun the ranks come from
the environment; `--gpus N` without a torchrun environment makes this process a launcher that starts N rank processes
itself (it never touches the GPU) and prints rank 0's line.
"""
import argparse
import datetime
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
STEP = (2, 1)    # per-frame image motion of the synthetic sequence
TH_PROJ = 15.0   # Tracking.cc:1060-1066 (monocular window)
INIT_WINDOW = 100  # Tracking.cc:738-748
HBM_PEAK_GBS = 8000.0
I8_MFMA_PEAK_TOPS = 5000.0  # dense int8 matrix rate (twice the 2.5 PFLOP/s bf16 figure)
PARITY_FRAMES = 8
KERNEL_OF_STAGE = {"pyramid": "k_pyr_resize", "fast_cells": "k_fast_cells", "octree": "k_octree_batch", "blur": "k_blur",
                   "describe": "k_describe"}
BACKEND = {"name": None}  # process-group backend in use (set by init_ranks)


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
    """(rank, local_rank, world) from the environment; joins the process group when world > 1.  `nccl` (RCCL) is tried
    first and checked with one all-reduce; if it cannot start -- the job needs it only for barriers and one max --
    the ranks fall back to gloo on CPU tensors (rendezvous on the next port).  BACKEND["name"] says which is in use."""
    from orb_slam2_amd.dist import env_world
    rank, local_rank, world = env_world()
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        limit = datetime.timedelta(seconds=180)
        if backend == "nccl":
            try:
                dist.init_process_group("nccl", device_id=device, timeout=limit)
                t = torch.ones(1, device=device)
                dist.all_reduce(t)
                torch.cuda.synchronize(device)
                if int(t.item()) != world:
                    raise RuntimeError("all-reduce over %d ranks returned %r" % (world, t.item()))
            except Exception as e:  # noqa: BLE001 -- any failure of the optional backend
                print("[rank %d] nccl process group unavailable (%s: %s); falling back to gloo" % (rank, type(e).__name__, e),
                      file=sys.stderr, flush=True)
                try:
                    if dist.is_initialized():
                        dist.destroy_process_group()
                except Exception:  # noqa: BLE001
                    pass
                os.environ["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + 1)
                backend = "gloo"
                dist.init_process_group("gloo", timeout=limit)
        else:
            dist.init_process_group(backend, timeout=limit)
    BACKEND["name"] = backend if world > 1 else None
    return rank, local_rank, world


def reduce_device(dev):
    """Where the rank-to-rank scalars live: the GPU under RCCL, the host under gloo."""
    return dev if BACKEND["name"] == "nccl" else None


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


def rank_report(rank, device, frames, wall):
    """One line per rank on stderr, so that a rank that never arrives is visible in the job's log."""
    print(json.dumps({"rank": rank, "device": str(device), "frames": int(frames), "wall_s": wall, "backend": BACKEND["name"]}),
          file=sys.stderr, flush=True)


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


def oracle_frame_step(O, api, ex, sc, bounds, prev, f):
    """The benchmark's per-frame work on the CPU: extract, SearchByProjection(Cur, Last) against the previous frame with
    the true image motion as the prediction, SearchForInitialization (window 100) between the two frames."""
    cur = ex.extract(f)
    q = api.shift_queries(prev[0], -STEP[0], -STEP[1], TH_PROJ, sc, bounds)
    nproj, assign = O.search_by_projection(1, q.view(O.WQUERY_DTYPE), prev[1], cur[0], cur[1], bounds, th_dist=100, nnratio=0.9,
                                           checkOri=True)
    pxy = np.stack([prev[0]["x"], prev[0]["y"]], axis=1).astype(np.float32).reshape(-1)
    ninit, m12, _ = O.search_for_initialization(prev[0], prev[1], cur[0], cur[1], bounds, pxy, windowSize=INIT_WINDOW, nnratio=0.9,
                                                checkOri=True)
    return cur, (nproj, assign, ninit, m12)


def cpu_stereo(O, synth, w, h, nfeat, bf, fx, seed, n_stereo):
    """Frame::Frame(stereo) as the reference runs it: left || right extraction on two threads (src/Frame.cc:78-81), then
    ComputeStereoMatches.  Median over n_stereo pairs after 3 warm-ups."""
    import threading
    L, R, _ = synth.stereo_pair(w, h, seed=seed)
    exL, exR = O.Extractor(nfeat, SCALE, NLEVELS, INI_TH, MIN_TH), O.Extractor(nfeat, SCALE, NLEVELS, INI_TH, MIN_TH)
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
                         tabs["scale"], tabs["inv_scale"], bf, bf / fx)
        if k >= 3:
            per.append(time.perf_counter() - t0)
    per = np.asarray(per)
    return {"value": 1.0 / float(np.median(per)), "unit": "stereo frames/s", "cores": 2, "kind": "port",
            "sample": "%d pairs of %dx%d, %d features per eye: left || right extraction on two threads + ComputeStereoMatches, "
                      "median %.2f ms" % (len(per), w, h, nfeat, np.median(per) * 1e3)}


def cpu_baseline(frames, n_frames=300, n_stereo=40, pool_threads=None, per_thread=60):
    """Timed like the reference's drivers (Examples/Monocular/mono_tum.cc:79-96: steady clock around the per-frame
    call, median over the frames) after 5 warm-up frames, on the first frames of the sequence the GPU processed.
    (a) one thread: extract + the two windowed matchers per frame -- what the reference does for mono / RGB-D
        (src/Frame.cc:136,212);
    (b) two threads: left || right extraction (src/Frame.cc:78-81) + ComputeStereoMatches on KITTI-sized pairs;
    (c) for context, the mono workload frame-parallel on the host cores this job may use.
    Also returns the oracle's outputs for frames 1..PARITY_FRAMES (the in-run parity check compares the GPU's with them)."""
    import threading
    from oracle import oracle as O
    from orb_slam2_amd import api, synth
    O.build()
    oe = O.Extractor(NFEAT, SCALE, NLEVELS, INI_TH, MIN_TH)
    sc = O.tables(oe.p)["scale"]
    bounds = np.array([0, 0, W, H], np.float32)

    t_start = time.perf_counter()
    prev = oe.extract(frames[0])
    per, golden = [], {}
    for k in range(1, n_frames + 6):
        t0 = time.perf_counter()
        prev, matches = oracle_frame_step(O, api, oe, sc, bounds, prev, frames[k % len(frames)])
        if k > 5:
            per.append(time.perf_counter() - t0)
        if k <= PARITY_FRAMES:
            golden[k] = (prev[0].copy(), prev[1].copy()) + matches
    per = np.asarray(per)
    res = {"value": 1.0 / float(np.median(per)), "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": "%d frames of the GPU's own sequence after 5 warm-ups, extract + SearchByProjection + SearchForInitialization per "
                     "frame on one thread, median per-frame time %.2f ms (mean %.2f ms)" % (len(per), np.median(per) * 1e3, per.mean() * 1e3),
           "cpu_model": cpu_model(), "host_cores_visible": os.cpu_count()}

    # (b) stereo, two threads
    res["stereo_2_threads"] = cpu_stereo(O, synth, 1241, 376, 2000, 386.1448, 718.856, 2, n_stereo)

    # (c) all cores this job may use (the box gives one GPU's share of the host: OMP_NUM_THREADS)
    nth = pool_threads or max(1, min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16") or 16)))
    exs = [O.Extractor(NFEAT, SCALE, NLEVELS, INI_TH, MIN_TH) for _ in range(nth)]

    def worker(t):
        p = exs[t].extract(frames[t % len(frames)])
        for k in range(per_thread):
            p, _ = oracle_frame_step(O, api, exs[t], sc, bounds, p, frames[(t * per_thread + k + 1) % len(frames)])
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
    return res, golden


def compare_extraction(k, d, ok, od):
    """Mismatch counts of one frame's keypoints / descriptors against the oracle's, to the bar of the parity tests:
    x, y, size, response, octave bit-exact, angle within 1e-4 rad, descriptors counted."""
    if len(k) != len(ok):
        return {"keypoint_count": 1}
    bad = {}
    for name in ("x", "y", "size", "response", "octave"):
        n = int((k[name] != ok[name]).sum())
        if n:
            bad[name] = n
    n = int((np.abs(k["angle"] - ok["angle"]) > np.degrees(1e-4)).sum())
    if n:
        bad["angle"] = n
    n = int((d != od).any(axis=1).sum())
    if n:
        bad["descriptors_differing"] = n
    return bad


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


def frame_algorithmic_bytes(w, h, kp_mean):
    """Bytes one frame's extraction has to move: read the image, write levels 1.., FAST reads every level once, the blur
    reads and writes every level, 60 bytes per keypoint out."""
    P, _ = level_pixels(w, h)
    return w * h + (P - w * h) + P + 2 * P + int(kp_mean * 60)


def committed_counters(kernel, launch_frames):
    """HBM traffic and vector-instruction counts of `kernel` from the committed rocprofv3 PMC passes (builder-run on a
    one-stream schedule with `launch_frames` frames per launch; counters cannot be collected inside a timed run)."""
    traffic = valu = None
    src = []
    for tag in ("r03", "r02", "r01"):
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
    """Chip-wide wave-instructions per second from the committed issue-rate microbenchmark (tools/valu_issue.hip), at 8
    waves per SIMD: the packed-integer / permute / dot-product class the FAST kernel is made of (580-590 G) -- not the
    plain two-operand logic class, which issues faster and which the kernel hardly uses."""
    for tag in ("r03", "r02"):
        p = os.path.join(ROOT, "profiles", "%s_valu_issue.json" % tag)
        if not os.path.exists(p):
            continue
        try:
            j = json.load(open(p))
            ops = ["v_pk_min_i16", "v_pk_max_u16", "v_perm_b32", "v_alignbyte_b32", "v_min_u32"]
            v = [j["ops"][o]["waves_per_simd_8"]["chip_Gwave_instr_per_s"] for o in ops if o in j["ops"]]
            if v:
                return float(np.mean(v)) * 1e9, "profiles/%s_valu_issue.json (packed / permute / min class)" % tag
        except Exception:
            pass
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
            if t and t[0] == "device_frames_per_frame" and out is not None:
                out["device_frame_make_build_destroy_ms"] = float(t[2])
        return out
    except Exception:
        return None
    finally:
        os.unlink(path)
    return None


def gpu_timed(torch, dev, fn, reps, warm=2):
    """Seconds per call of `fn` (which enqueues GPU work), wall clock between device synchronisations."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / reps


# ---------------------------------------------------------------------------------------------------------------
# workloads
# ---------------------------------------------------------------------------------------------------------------
def run_tum(args, rank, local_rank, world, torch, dev, barrier):
    from orb_slam2_amd import api, synth
    from orb_slam2_amd.dist import max_over_ranks
    B, NS = args.batch, args.sub_batches
    frames = synth.sequence(W, H, B, seed=1 + rank, step=STEP, per_frame_density=True)
    # sub-batch 0 = the generated sequence; sub-batches 1.. = the same camera path with their own sensor noise (made on
    # the device: every frame of the step is a different image)
    d_imgs = [torch.from_numpy(frames).to(dev)]
    gen = torch.Generator(device=dev)
    gen.manual_seed(1000 + rank)
    for j in range(1, NS):
        noise = torch.randint(-3, 4, (B, H, W), dtype=torch.int16, device=dev, generator=gen)
        d_imgs.append((d_imgs[0].to(torch.int16) + noise).clamp_(0, 255).to(torch.uint8))
        del noise

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

    # slot 0 = last frame of the previous sub-batch, slots 1..B = this sub-batch's frames
    d_kps = torch.zeros((B + 1, cap, 28), dtype=torch.uint8, device=dev)
    d_desc = torch.zeros((B + 1, cap, 32), dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(B + 1, dtype=torch.int32, device=dev)
    d_q = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_assign = torch.zeros((B, cap), dtype=torch.int32, device=dev)
    d_nm = torch.zeros(B, dtype=torch.int32, device=dev)
    d_m12 = torch.zeros((B, cap), dtype=torch.int32, device=dev)
    d_ni = torch.zeros(B, dtype=torch.int32, device=dev)

    def substep(imgs):
        for k in range(S):
            lo = k * Bs  # frames lo .. lo+Bs-1 of the batch live in slots lo+1 .. lo+Bs
            with torch.cuda.stream(streams[k]):
                # slot lo+1.. are still read by the previous sub-batch's matching on the next stream (its slot "lo+Bs")
                streams[k].wait_event(ev_matched[k + 1] if k + 1 < S else ev_carried)
                exs[k].extract_batch_device(imgs[lo].data_ptr(), Bs, W, H, W, W * H, d_kps[lo + 1].data_ptr(),
                                            d_desc[lo + 1].data_ptr(), cap, d_cnt[lo + 1:].data_ptr())
                ev_extracted[k].record(streams[k])
                # the predecessor of this sub-pipeline's first frame comes from the one before it (slot lo)
                if k > 0:
                    streams[k].wait_event(ev_extracted[k - 1])
                mts[k].make_shift_queries_device(Bs, d_kps[lo].data_ptr(), d_cnt[lo:].data_ptr(), cap, -float(STEP[0]),
                                                 -float(STEP[1]), TH_PROJ, scale, bounds, d_q[lo].data_ptr())
                mts[k].search_by_projection_batch_device(api.MODE_LAST_FRAME, Bs, d_q[lo].data_ptr(), d_desc[lo].data_ptr(),
                                                         d_cnt[lo:].data_ptr(), cap, d_kps[lo + 1].data_ptr(),
                                                         d_desc[lo + 1].data_ptr(), d_cnt[lo + 1:].data_ptr(), cap, bounds,
                                                         d_assign[lo].data_ptr(), d_nm[lo:].data_ptr(), th_dist=api.TH_HIGH)
                mts[k].search_for_initialization_batch_device(Bs, d_kps[lo].data_ptr(), d_desc[lo].data_ptr(), d_cnt[lo:].data_ptr(), cap,
                                                              d_kps[lo + 1].data_ptr(), d_desc[lo + 1].data_ptr(),
                                                              d_cnt[lo + 1:].data_ptr(), cap, bounds, d_m12[lo].data_ptr(),
                                                              d_ni[lo:].data_ptr(), windowSize=INIT_WINDOW)
                ev_matched[k].record(streams[k])
        with torch.cuda.stream(streams[0]):
            # carry the last frame into slot 0 for the next sub-batch (after sub-pipeline 0 has matched against the old slot 0)
            streams[0].wait_event(ev_extracted[S - 1])
            d_kps[0].copy_(d_kps[B], non_blocking=True)
            d_desc[0].copy_(d_desc[B], non_blocking=True)
            d_cnt[0:1].copy_(d_cnt[B:B + 1], non_blocking=True)
            ev_carried.record(streams[0])

    def step():
        for imgs in d_imgs:
            substep(imgs)

    def sync():
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    sync()
    ex.profile(True)  # HIP events between the stages on the stream the kernels are launched on
    wall = timed_steps(step, args.steps, 0, sync, barrier)
    stage_ms, _ = ex.stage_times()
    ex.profile(False)
    rank_report(rank, dev, B * NS * args.steps, wall)
    wall_max = max_over_ranks(wall, reduce_device(dev) if world > 1 else None)
    kp_mean = float(d_cnt[1:].float().mean().item())
    nm_mean = float(d_nm.float().mean().item())
    ni_mean = float(d_ni.float().mean().item())

    # one more pass over sub-batch 0 (the host-generated frames): its first frames are what the parity check compares
    substep(d_imgs[0])
    sync()
    parity_gpu = None
    if rank == 0:
        n = PARITY_FRAMES + 1
        parity_gpu = {"kps": d_kps[1:n + 1].cpu().numpy().view(api.KEYPOINT_DTYPE).reshape(n, cap), "desc": d_desc[1:n + 1].cpu().numpy(),
                      "cnt": d_cnt[1:n + 1].cpu().numpy(), "assign": d_assign[:n].cpu().numpy(), "nm": d_nm[:n].cpu().numpy(),
                      "m12": d_m12[:n].cpu().numpy(), "ni": d_ni[:n].cpu().numpy()}

    # the same launches of sub-pipeline 0 with nothing else on the chip (after the timed region): in the timed region
    # the event intervals include the other sub-pipelines' kernels and the side-stream blur, this pass tells the stages apart
    ex.profile(2)  # blur kept on the main stream: every stage alone
    for _ in range(3):
        with torch.cuda.stream(stream):
            ex.extract_batch_device(d_imgs[0][0].data_ptr(), Bs, W, H, W, W * H, d_kps[1].data_ptr(), d_desc[1].data_ptr(), cap,
                                    d_cnt[1:].data_ptr())
    sync()
    iso_ms, _ = ex.stage_times()
    ex.profile(False)

    if rank != 0:
        return None
    # the matching kernels alone, one sub-pipeline's launches
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    with torch.cuda.stream(stream):
        ev[0].record(stream)
        for _ in range(3):
            mt.make_shift_queries_device(Bs, d_kps[0].data_ptr(), d_cnt.data_ptr(), cap, -float(STEP[0]), -float(STEP[1]), TH_PROJ, scale,
                                         bounds, d_q[0].data_ptr())
            mt.search_by_projection_batch_device(api.MODE_LAST_FRAME, Bs, d_q[0].data_ptr(), d_desc[0].data_ptr(), d_cnt.data_ptr(), cap,
                                                 d_kps[1].data_ptr(), d_desc[1].data_ptr(), d_cnt[1:].data_ptr(), cap, bounds,
                                                 d_assign[0].data_ptr(), d_nm.data_ptr(), th_dist=api.TH_HIGH)
        ev[1].record(stream)
        for _ in range(3):
            mt.search_for_initialization_batch_device(Bs, d_kps[0].data_ptr(), d_desc[0].data_ptr(), d_cnt.data_ptr(), cap, d_kps[1].data_ptr(),
                                                      d_desc[1].data_ptr(), d_cnt[1:].data_ptr(), cap, bounds, d_m12[0].data_ptr(),
                                                      d_ni.data_ptr(), windowSize=INIT_WINDOW)
        ev[2].record(stream)
    sync()
    match_iso = {"search_by_projection": ev[0].elapsed_time(ev[1]) / 3, "search_for_initialization": ev[1].elapsed_time(ev[2]) / 3}

    wstats = [mt.window_stats(p) for p in range(0, Bs, max(1, Bs // 8))]
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
    frame_alg = frame_algorithmic_bytes(W, H, kp_mean)
    total_frames = world * B * NS * args.steps
    out = bench_line(
        "frames/sec ORB extract+match at 640x480/1000 kp", "frames/s", total_frames / wall_max, world, args.steps,
        args.warmup, wall_max, "weak", "u8", "synthetic",
        {"workload": "TUM mono 640x480, 1000 features, 8 levels, scale 1.2, FAST 20/7: extract + SearchByProjection(Cur,Last) "
                     "th=15 + SearchForInitialization window 100 against the previous frame",
         "frames_per_step_per_gpu": B * NS, "sub_batches_per_step": NS, "batch_per_gpu": B, "streams": S,
         "keypoints_per_frame": kp_mean, "projection_matches_per_frame": nm_mean, "initialization_matches_per_frame": ni_mean,
         "parallelism": "every rank its own sequence on its own GPU (%d), no collective; ranks lined up over %s"
                        % (world, BACKEND["name"] or "nothing (one process)")},
        roofline={"bound": "hbm", "kernel": KERNEL_OF_STAGE[dom_name], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                  "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "algorithmic_bytes_per_launch": dom_bytes,
                  "launch_ms": float(stage_ms[dom]), "launch_ms_isolated": float(iso_ms[dom]),
                  "achieved_isolated": dom_bytes / (float(iso_ms[dom]) * 1e-3) / 1e9, "valu_issue": valu,
                  "source": {"achieved": "HIP events in this run", "traffic_and_instruction_counts": src + ["builder-run rocprofv3 --pmc passes"]},
                  "note": "integer byte kernel: its L2 fetches stay within about twice the algorithmic volume, the time goes into vector "
                          "instructions (valu_issue: instructions per launch over the isolated launch time, against the measured issue rate)"},
        stages_ms_per_step={n: float(v) for n, v in zip(names, stage_ms)},
        stages_ms_isolated={n: float(v) for n, v in zip(names, iso_ms)},
        matching_ms_isolated=match_iso,
        extract_algorithmic_GBps=frame_alg * Bs / (float(iso_ms.sum()) * 1e-3) / 1e9,
        step_algorithmic_GBps=frame_alg * total_frames / wall_max / 1e9,
        window_resolver_iterations_rescans=wstats)
    out["_parity_gpu"] = parity_gpu
    out["_frames"] = frames

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


def parity_in_run(gpu, golden):
    """The timed batch's first frames against the oracle's results on the same images (computed by the cpu_baseline leg):
    keypoints, descriptors, the SearchByProjection assignment and the SearchForInitialization matches of every frame
    against its predecessor."""
    from orb_slam2_amd import api
    rep = {"frames": 0, "mismatches": 0, "keypoints_compared": 0, "descriptors_differing": 0, "detail": {}}
    for k in sorted(golden):
        ok, od, nproj, assign, ninit, m12 = golden[k]
        n = int(gpu["cnt"][k])
        kg, dg = gpu["kps"][k, :n], gpu["desc"][k, :n]
        bad = compare_extraction(kg, dg, ok.view(api.KEYPOINT_DTYPE), od)
        rep["descriptors_differing"] += bad.pop("descriptors_differing", 0)
        # frame k is the train side of pair k (queries from frame k-1)
        if int(gpu["nm"][k]) != nproj or not np.array_equal(gpu["assign"][k, :n], assign):
            bad["search_by_projection"] = 1
        n1 = int(gpu["cnt"][k - 1])
        if int(gpu["ni"][k]) != ninit or not np.array_equal(gpu["m12"][k, :n1], m12):
            bad["search_for_initialization"] = 1
        rep["frames"] += 1
        rep["keypoints_compared"] += n
        if bad:
            rep["mismatches"] += 1
            rep["detail"][str(k)] = bad
    rep["bar"] = ("x, y, size, response, octave, match assignments and counts bit-exact; angle within 1e-4 rad; descriptors differing "
                  "are counted (allowed < 0.1 %)")
    if rep["keypoints_compared"] and rep["descriptors_differing"] > 0.001 * rep["keypoints_compared"]:
        rep["mismatches"] += 1
        rep["detail"]["descriptors"] = "more than 0.1 % differ"
    return rep


# ---------------------------------------------------------------------------------------------------------------
# the other benchmark configurations (one GPU, after the timed region of the main workload)
# ---------------------------------------------------------------------------------------------------------------
def cfg_stereo(torch, dev, name, w, h, nfeat, bf, fx, seed, B, with_cpu):
    """B stereo pairs per step, device-resident: extract the left and the right batch (two handles, two streams), then
    ComputeStereoMatches for all pairs (orbgpu_stereo_match_batch_device)."""
    from oracle import oracle as O
    from orb_slam2_amd import api, synth
    ndist = 8
    pairs = [synth.stereo_pair(w, h, seed=seed + 10 * i)[:2] for i in range(ndist)]
    idx = torch.arange(B, device=dev) % ndist
    dL = torch.from_numpy(np.stack([p[0] for p in pairs])).to(dev)[idx].contiguous()
    dR = torch.from_numpy(np.stack([p[1] for p in pairs])).to(dev)[idx].contiguous()
    exL = api.ORBextractor(nfeat, SCALE, NLEVELS, INI_TH, MIN_TH, max_width=w, max_height=h, max_batch=B)
    exR = api.ORBextractor(nfeat, SCALE, NLEVELS, INI_TH, MIN_TH, max_width=w, max_height=h, max_batch=B)
    m = api.ORBmatcher()
    sL, sR = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    exL.set_stream(sL.cuda_stream); m.set_stream(sL.cuda_stream); exR.set_stream(sR.cuda_stream)
    evR, evDone = torch.cuda.Event(), torch.cuda.Event()
    cap = exL.max_keypoints
    z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
    kk = [z((B, cap, 28), torch.uint8) for _ in range(2)]; dd = [z((B, cap, 32), torch.uint8) for _ in range(2)]
    cc = [z(B, torch.int32) for _ in range(2)]
    uR, dep, nv = z((B, cap), torch.float32), z((B, cap), torch.float32), z(B, torch.int32)

    def step():
        with torch.cuda.stream(sR):
            sR.wait_event(evDone)  # the previous step's matching still reads the right side's buffers
            exR.extract_batch_device(dR.data_ptr(), B, w, h, w, w * h, kk[1].data_ptr(), dd[1].data_ptr(), cap, cc[1].data_ptr())
            evR.record(sR)
        with torch.cuda.stream(sL):
            exL.extract_batch_device(dL.data_ptr(), B, w, h, w, w * h, kk[0].data_ptr(), dd[0].data_ptr(), cap, cc[0].data_ptr())
            sL.wait_event(evR)
            m.stereo_match_batch_device(exL, exR, B, kk[0].data_ptr(), dd[0].data_ptr(), cc[0].data_ptr(), cap, kk[1].data_ptr(),
                                        dd[1].data_ptr(), cc[1].data_ptr(), cap, bf, bf / fx, uR.data_ptr(), dep.data_ptr(), nv.data_ptr())
            evDone.record(sL)
    dt = gpu_timed(torch, dev, step, 5)
    # the matching alone
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(sL):
        e0.record(sL)
        for _ in range(3):
            m.stereo_match_batch_device(exL, exR, B, kk[0].data_ptr(), dd[0].data_ptr(), cc[0].data_ptr(), cap, kk[1].data_ptr(),
                                        dd[1].data_ptr(), cc[1].data_ptr(), cap, bf, bf / fx, uR.data_ptr(), dep.data_ptr(), nv.data_ptr())
        e1.record(sL)
    torch.cuda.synchronize(dev)
    match_ms = e0.elapsed_time(e1) / 3
    nL, nR = cc[0].cpu().numpy(), cc[1].cpu().numpy()
    kp_mean = float(nL.mean() + nR.mean()) / 2
    # parity: pair 0 against the oracle's ComputeStereoMatches on the GPU's own keypoints, descriptors and pyramids
    kL = kk[0][0].cpu().numpy().view(api.KEYPOINT_DTYPE).reshape(-1)[:nL[0]]; kR = kk[1][0].cpu().numpy().view(api.KEYPOINT_DTYPE).reshape(-1)[:nR[0]]
    desL, desR = dd[0][0].cpu().numpy()[:nL[0]], dd[1][0].cpu().numpy()[:nR[0]]
    t = O.tables(O.make_params(nfeat))
    rn, ruR, rdep = O.stereo_matches(kL.view(O.KEYPOINT_DTYPE), desL, kR.view(O.KEYPOINT_DTYPE), desR,
                                     [exL.pyramid_level(l, frame=0) for l in range(NLEVELS)],
                                     [exR.pyramid_level(l, frame=0) for l in range(NLEVELS)], t["scale"], t["inv_scale"], bf, bf / fx)
    ok = int(nv[0].item()) == rn and np.array_equal(uR[0].cpu().numpy()[:nL[0]], ruR) and np.array_equal(dep[0].cpu().numpy()[:nL[0]], rdep)
    # candidates the row-band scan looks at: right keypoints whose band covers the left keypoint's row (src/Frame.cc:617-627, 638-650)
    rr = 2.0 * t["scale"][np.clip(kR["octave"], 0, NLEVELS - 1)]
    lo, hi = np.floor(kR["y"] - rr), np.ceil(kR["y"] + rr)
    rowL = kL["y"].astype(np.int32)
    cand = int(((rowL[:, None] >= lo[None, :]) & (rowL[:, None] <= hi[None, :])).sum())
    alg = 2 * frame_algorithmic_bytes(w, h, kp_mean)
    res = {"workload": name, "pairs_per_step": B, "distinct_pairs": ndist, "stereo_frames_per_s": B / dt, "ms_per_stereo_frame": dt / B * 1e3,
           "keypoints_per_eye": kp_mean, "depths_per_frame": float(nv.float().mean().item()),
           "stereo_match_ms_per_batch": match_ms, "row_band_candidates_per_frame": cand,
           "row_band_candidates_per_s_matching_alone": cand * B / (match_ms * 1e-3),
           "roofline": {"bound": "hbm", "achieved": alg * B / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg * B / dt / 1e9 / HBM_PEAK_GBS,
                        "algorithmic_bytes_per_stereo_frame": alg, "note": "both extractions + matching, wall clock of this run"},
           "parity_in_run": {"pairs": 1, "mismatches": 0 if ok else 1, "against": "oracle ComputeStereoMatches on the GPU's keypoints and pyramids of pair 0"}}
    if with_cpu:
        res["cpu_baseline"] = cpu_stereo(O, synth, w, h, nfeat, bf, fx, seed, 12)
    return res


def cfg_euroc_search(torch, dev, B):
    """The three Tracking-side SearchByProjection variants (src/ORBmatcher.cc:45-129, 1379-1521, 1523-1650) between
    consecutive EuRoC-sized frames (752x480, 1200 features): B frame pairs per launch, queries = the previous frame's
    keypoints displaced by the true image motion, with each variant's radius, level window and thresholds."""
    from oracle import oracle as O
    from orb_slam2_amd import api, synth
    w, h, nfeat = 752, 480, 1200
    ndist = 33
    seq = synth.sequence(w, h, ndist, seed=3, step=STEP, per_frame_density=True)
    ex = api.ORBextractor(nfeat, SCALE, NLEVELS, INI_TH, MIN_TH, max_width=w, max_height=h, max_batch=ndist)
    cap = ex.max_keypoints
    res = ex.extract_batch(seq)
    sc = ex.GetScaleFactors()
    bounds = np.array([0, 0, w, h], np.float32)
    # variants: (mode, radius factor, (level below, level above), distance threshold, nnratio)
    variants = {
        "local_map (SearchByProjection(F, vpMapPoints, th=1): r = 4 * scale, levels [l-1, l], TH_HIGH, ratio 0.8)": (api.MODE_LOCAL_MAP, 4.0, (1, 0), api.TH_HIGH, 0.8),
        "last_frame (SearchByProjection(Cur, Last, th=7, stereo): r = 7 * scale, levels [l-1, l+1], TH_HIGH, ratio 0.9)": (api.MODE_LAST_FRAME, 7.0, (1, 1), api.TH_HIGH, 0.9),
        "keyframe (SearchByProjection(Cur, KF, found, th=10, ORBdist=100): r = 10 * scale, levels [l-1, l+1])": (api.MODE_KEYFRAME, 10.0, (1, 1), 100, 0.9),
    }
    npair = ndist - 1
    K2 = np.zeros((B, cap), api.KEYPOINT_DTYPE); D2 = np.zeros((B, cap, 32), np.uint8); N2 = np.zeros(B, np.int32)
    QD = np.zeros((B, cap, 32), np.uint8); NQ = np.zeros(B, np.int32)
    for p in range(B):
        (ka, da), (kb, db) = res[p % npair], res[p % npair + 1]
        K2[p, :len(kb)] = kb; D2[p, :len(kb)] = db; N2[p] = len(kb); QD[p, :len(ka)] = da; NQ[p] = len(ka)
    up = lambda a: torch.from_numpy(a.view(np.uint8) if a.dtype.fields else a).to(dev)
    dK2, dD2, dN2, dQD, dNQ = up(K2), up(D2), up(N2), up(QD), up(NQ)
    dA = torch.zeros((B, cap), dtype=torch.int32, device=dev); dNM = torch.zeros(B, dtype=torch.int32, device=dev)
    out = {"workload": "EuRoC 752x480, 1200 features: SearchByProjection between consecutive frames, %d pairs per launch (%d distinct)" % (B, npair),
           "variants": {}}
    for vname, (mode, rf, (below, above), th_dist, ratio) in variants.items():
        m = api.ORBmatcher(ratio, True)
        Q = np.zeros((B, cap), api.WQUERY_DTYPE)
        qs = []
        for p in range(npair):
            ka = res[p][0]
            q = api.shift_queries(ka, -STEP[0], -STEP[1], rf, sc, bounds)
            q["minLevel"] = ka["octave"] - below; q["maxLevel"] = ka["octave"] + above
            qs.append(q)
        for p in range(B):
            Q[p, :len(qs[p % npair])] = qs[p % npair]
        dQ = up(Q)

        def step():
            m.search_by_projection_batch_device(mode, B, dQ.data_ptr(), dQD.data_ptr(), dNQ.data_ptr(), cap, dK2.data_ptr(), dD2.data_ptr(),
                                                dN2.data_ptr(), cap, bounds, dA.data_ptr(), dNM.data_ptr(), th_dist=th_dist)
        m.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        dt = gpu_timed(torch, dev, step, 10)
        A = dA.cpu().numpy(); NM = dNM.cpu().numpy()
        bad, cand, t_cpu = 0, 0, []
        for p in range(4):
            (ka, da), (kb, db) = res[p], res[p + 1]
            t0 = time.perf_counter()
            rn, rassign = O.search_by_projection(mode, qs[p].view(O.WQUERY_DTYPE), da, kb.view(O.KEYPOINT_DTYPE), db, bounds, th_dist=th_dist,
                                                 nnratio=ratio, checkOri=True)
            t_cpu.append(time.perf_counter() - t0)
            bad += int(NM[p] != rn or not np.array_equal(A[p, :len(kb)], rassign))
            q = qs[p][qs[p]["flags"] != 0]
            inwin = (np.abs(kb["x"][None, :] - q["u"][:, None]) < q["r"][:, None]) & (np.abs(kb["y"][None, :] - q["v"][:, None]) < q["r"][:, None]) & \
                    (kb["octave"][None, :] >= q["minLevel"][:, None]) & (kb["octave"][None, :] <= q["maxLevel"][:, None])
            cand += int(inwin.sum())
        cand_pair = cand / 4.0
        out["variants"][vname] = {
            "pairs_per_s": B / dt, "ms_per_launch": dt * 1e3, "matches_per_pair": float(NM.mean()), "candidates_per_pair": cand_pair,
            "candidates_per_s": cand_pair * B / dt,
            "roofline": {"bound": "hbm", "achieved": cand_pair * 64 * B / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": cand_pair * 64 * B / dt / 1e9 / HBM_PEAK_GBS,
                         "note": "64 bytes per evaluated candidate (two descriptors); the kernels are latency-bound (a few dozen candidates per "
                                 "query, sequential claim resolution), far from any throughput roof"},
            "parity_in_run": {"pairs": 4, "mismatches": bad},
            "cpu_baseline": {"value": 1.0 / float(np.median(t_cpu)), "unit": "pairs/s", "cores": 1, "kind": "port", "sample": "4 pairs, one thread"}}
    return out


class SplitExtractor:
    """n frames per call, split over `nsplit` extractor handles on their own streams (contiguous shares): for small
    batches of large images the latency-bound quadtree of one share then overlaps the FAST / blur kernels of another."""

    def __init__(self, torch, dev, api, nfeat, w, h, n, nsplit, device=0):
        self.torch, self.dev, self.w, self.h = torch, dev, w, h
        self.shares = [(i * n // nsplit, (i + 1) * n // nsplit) for i in range(nsplit)]
        self.shares = [(a, b) for a, b in self.shares if b > a]
        self.exs = [api.ORBextractor(nfeat, SCALE, NLEVELS, INI_TH, MIN_TH, device=device, max_width=w, max_height=h, max_batch=b - a)
                    for a, b in self.shares]
        self.streams = [torch.cuda.Stream(device=dev) for _ in self.shares]
        for e, st in zip(self.exs, self.streams):
            e.set_stream(st.cuda_stream)
        self.cap = self.exs[0].max_keypoints

    def __call__(self, d_imgs, d_k, d_d, d_c):
        for (a, b), e in zip(self.shares, self.exs):
            e.extract_batch_device(d_imgs[a].data_ptr(), b - a, self.w, self.h, self.w, self.w * self.h, d_k[a].data_ptr(), d_d[a].data_ptr(),
                                   self.cap, d_c[a:].data_ptr())


def cfg_4k(torch, dev, total, launches):
    """`total` frames of 3840x2160 with 8000 features, extraction only, device-resident (4 distinct frames cycled), at
    several (frames per call, handles the call is split over): a rank's share at 8 ranks is 32 frames."""
    from oracle import oracle as O
    from orb_slam2_amd import api, synth
    w, h, nfeat = 3840, 2160, 8000
    distinct = synth.sequence(w, h, 4, seed=4, n_shapes=(w * h) // 768, smax=80)
    d_imgs = torch.from_numpy(distinct).to(dev)[torch.arange(total, device=dev) % 4].contiguous()
    out = {"workload": "%d frames 3840x2160, 8000 features, 8 levels: extraction, device-resident" % total, "frames_per_call": {}}
    best = None
    for per_launch, nsplit in launches:
        n = min(per_launch, total)
        sx = SplitExtractor(torch, dev, api, nfeat, w, h, n, nsplit)
        cap = sx.cap
        d_k = torch.zeros((n, cap, 28), dtype=torch.uint8, device=dev); d_d = torch.zeros((n, cap, 32), dtype=torch.uint8, device=dev)
        d_c = torch.zeros(n, dtype=torch.int32, device=dev)
        dt = gpu_timed(torch, dev, lambda: sx(d_imgs, d_k, d_d, d_c), max(3, min(8, 256 // n)))
        kp_mean = float(d_c.float().mean().item())
        alg = frame_algorithmic_bytes(w, h, kp_mean)
        r = {"frames_per_s": n / dt, "ms_per_call": dt * 1e3, "keypoints_per_frame": kp_mean,
             "roofline": {"bound": "hbm", "achieved": alg * n / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg * n / dt / 1e9 / HBM_PEAK_GBS,
                          "algorithmic_bytes_per_frame": alg}}
        if nsplit == 1:
            ex = sx.exs[0]
            ex.profile(2); sx(d_imgs, d_k, d_d, d_c); iso_ms, _ = ex.stage_times(); ex.profile(False)
            r["stages_ms_isolated"] = {nm: float(v) for nm, v in zip(["pyramid", "fast_cells", "octree", "blur", "describe"], iso_ms)}
        out["frames_per_call"]["%d frames, %d handle%s" % (n, len(sx.shares), "" if len(sx.shares) == 1 else "s on their own streams")] = r
        if best is None or r["frames_per_s"] > best["frames_per_s"]:
            best = r
        if (per_launch, nsplit) == launches[-1]:
            # parity + CPU leg: frame 0 through the oracle (one 4K extraction on one host core)
            t0 = time.perf_counter()
            ok_k, ok_d = O.Extractor(nfeat, SCALE, NLEVELS, INI_TH, MIN_TH).extract(distinct[0])
            t_cpu = time.perf_counter() - t0
            torch.cuda.synchronize(dev)
            c0 = int(d_c[0].item())
            kg = d_k[0].cpu().numpy().view(api.KEYPOINT_DTYPE).reshape(-1)[:c0]; dg = d_d[0].cpu().numpy()[:c0]
            bad = compare_extraction(kg, dg, ok_k.view(api.KEYPOINT_DTYPE), ok_d)
            ndiff = bad.pop("descriptors_differing", 0)
            out["parity_in_run"] = {"frames": 1, "mismatches": int(bool(bad) or ndiff > 0.001 * max(c0, 1)), "keypoints_compared": c0,
                                    "descriptors_differing": ndiff, "detail": bad}
            out["cpu_baseline"] = {"value": 1.0 / t_cpu, "unit": "frames/s", "cores": 1, "kind": "port", "sample": "one frame, one thread, %.2f s" % t_cpu}
        del sx, d_k, d_d, d_c
    out["frames_per_s"] = best["frames_per_s"]
    out["roofline"] = best["roofline"]
    return out


def cfg_bruteforce(torch, dev, n):
    """n x n brute-force Hamming scan (loop-closure scale): the matrix-core kernel (the default at this size) and the
    vector-unit kernel beside it."""
    from oracle import oracle as O
    from orb_slam2_amd import api, synth
    Qh, Th = synth.descriptors(n, 5), synth.descriptors(n, 6)
    Q, T = torch.from_numpy(Qh).to(dev), torch.from_numpy(Th).to(dev)
    m = api.ORBmatcher()
    m.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    o = [torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(3)]
    res = {"workload": "brute force %dk x %dk descriptors of 256 bits: best index, best and second-best distance per query" % (n // 1000, n // 1000)}
    pairs = float(n) * n
    for which, name in ((2, "matrix_cores"), (1, "vector_unit")):
        m.set_bruteforce_kernel(which)
        dt = gpu_timed(torch, dev, lambda: m.bruteforce_device(Q.data_ptr(), n, T.data_ptr(), n, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr()), 5)
        res[name] = {"ms": dt * 1e3, "pairs_per_s": pairs / dt}
        if which == 2:
            tops = 2 * 256 * pairs / dt / 1e12
            res["pairs_per_s"] = pairs / dt
            res["roofline"] = {"bound": "mfma", "achieved": tops, "peak": I8_MFMA_PEAK_TOPS, "unit": "TOP/s", "frac": tops / I8_MFMA_PEAK_TOPS,
                               "note": "v_mfma_i32_32x32x32_i8 on byte-per-bit images: 256 multiply-adds per pair; HBM traffic is 64 B per "
                                       "descriptor in, 12 B per query out (%.1f MB per call)" % ((64.0 * n + 12.0 * n) / 1e6)}
            got = [t.cpu().numpy() for t in o]
        else:
            lanes_clk = 256 * 4 * 16 * 2.4e9  # CUs x SIMDs x lanes per clock x clock
            res[name]["valu_util_at_16_ops_per_pair"] = 16 * pairs / dt / lanes_clk
            res["kernels_agree"] = all(np.array_equal(a, t.cpu().numpy()) for a, t in zip(got, o))
    m.set_bruteforce_kernel(0)
    ns = 1000
    t0 = time.perf_counter()
    ref = O.bruteforce(Qh[:ns], Th)
    t_cpu = time.perf_counter() - t0
    res["parity_in_run"] = {"queries": ns, "mismatches": int(sum(not np.array_equal(g[:ns], r) for g, r in zip(got, ref)))}
    res["cpu_baseline"] = {"value": ns * float(n) / t_cpu, "unit": "pairs/s", "cores": 1, "kind": "port", "sample": "%d queries x %d, one thread" % (ns, n)}
    return res


def run_configs(args, torch, dev, with_cpu):
    out = {}
    todo = [("kitti_stereo", lambda: cfg_stereo(torch, dev, "KITTI 1241x376 stereo, 2000 features per eye: two extractions + ComputeStereoMatches",
                                                1241, 376, 2000, 386.1448, 718.856, 2, args.stereo_batch, False)),
            ("euroc_stereo", lambda: cfg_stereo(torch, dev, "EuRoC 752x480 stereo, 1200 features per eye: two extractions + ComputeStereoMatches",
                                                752, 480, 1200, 47.90639384423901, 435.2046959714599, 3, args.stereo_batch, with_cpu)),
            ("euroc_search_by_projection", lambda: cfg_euroc_search(torch, dev, args.stereo_batch)),
            ("batch_4k", lambda: cfg_4k(torch, dev, args.frames_4k, [(args.frames_4k, 1), (32, 1), (32, 2), (32, 4)])),
            ("bruteforce_100k", lambda: cfg_bruteforce(torch, dev, 100000))]
    for name, fn in todo:
        t0 = time.perf_counter()
        try:
            out[name] = fn()
        except Exception as e:  # noqa: BLE001 -- a side configuration must not lose the main line
            out[name] = {"error": "%s: %s" % (type(e).__name__, e)}
        out[name]["seconds"] = time.perf_counter() - t0
        torch.cuda.synchronize(dev)
        torch.cuda.empty_cache()
    return out


def run_4k(args, rank, local_rank, world, torch, dev, barrier):
    """256 frames 3840x2160 / 8000 features, extraction only, sharded over the ranks in contiguous chunks."""
    from orb_slam2_amd import api, synth
    from orb_slam2_amd.dist import max_over_ranks, shard_frames
    w, h, nfeat, total = 3840, 2160, 8000, args.frames_4k
    mine = shard_frames(total, world, rank)
    n = len(mine)
    distinct = synth.sequence(w, h, 4, seed=4, n_shapes=(w * h) // 768, smax=80)  # four distinct frames, cycled by global frame index
    idx = torch.tensor([g % 4 for g in mine], dtype=torch.long)
    d_imgs = torch.from_numpy(distinct).to(dev)[idx.to(dev)].contiguous()
    per_launch = min(n, args.launch_4k)
    # a small share (a rank's 32 frames at 8 ranks) is split over two handles on their own streams: the quadtree of one
    # half then runs beside the other half's FAST and blur instead of leaving the chip idle
    nsplit = args.split_4k if args.split_4k > 0 else (2 if per_launch <= 64 else 1)
    sx = SplitExtractor(torch, dev, api, nfeat, w, h, per_launch, nsplit, device=local_rank)
    ex, cap = sx.exs[0], sx.cap
    d_k = torch.zeros((n, cap, 28), dtype=torch.uint8, device=dev)
    d_d = torch.zeros((n, cap, 32), dtype=torch.uint8, device=dev)
    d_c = torch.zeros(n, dtype=torch.int32, device=dev)

    def step():
        for lo in range(0, n, per_launch):
            if lo + per_launch <= n:
                sx(d_imgs[lo:], d_k[lo:], d_d[lo:], d_c[lo:])
            else:  # a last, shorter launch goes through the first handle
                ex.extract_batch_device(d_imgs[lo].data_ptr(), n - lo, w, h, w, w * h, d_k[lo].data_ptr(), d_d[lo].data_ptr(), cap,
                                        d_c[lo:].data_ptr())

    def sync():
        torch.cuda.synchronize(dev)

    wall = timed_steps(step, args.steps, args.warmup, sync, barrier)
    rank_report(rank, dev, n * args.steps, wall)
    wall_max = max_over_ranks(wall, reduce_device(dev) if world > 1 else None)
    ex.profile(2)
    step(); sync()
    iso_ms, _ = ex.stage_times()
    ex.profile(False)
    if rank != 0:
        return None
    kp_mean = float(d_c.float().mean().item())
    P, _ = level_pixels(w, h)
    frame_alg = frame_algorithmic_bytes(w, h, kp_mean)
    names = ["pyramid", "fast_cells", "octree", "blur", "describe"]
    dom = 1 + int(np.argmax(iso_ms[1:]))
    alg = {"fast_cells": P, "octree": 4 * 32000, "blur": 2 * P, "describe": int(kp_mean * 60)}[names[dom]]
    m_last = sx.shares[0][1] - sx.shares[0][0] if n % per_launch == 0 else n % per_launch  # frames of the first handle's last launch
    achieved = alg * m_last / (float(iso_ms[dom]) * 1e-3) / 1e9
    return bench_line(
        "frames/sec ORB extract at 3840x2160/8000 kp", "frames/s", total * args.steps / wall_max, world, args.steps, args.warmup,
        wall_max, "strong", "u8", "synthetic (4 distinct frames cycled)",
        {"workload": "256-frame batch 3840x2160, 8000 features, 8 levels: extraction, device-resident, frames sharded in contiguous "
                     "chunks across the ranks",
         "frames_total": total, "frames_this_rank": n, "frames_per_launch": per_launch, "handles_per_launch": len(sx.shares),
         "keypoints_per_frame": kp_mean,
         "parallelism": "%d rank(s), no collective; ranks lined up over %s" % (world, BACKEND["name"] or "nothing (one process)")},
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
    ap.add_argument("--batch", type=int, default=2048, help="tum: frames per sub-batch and GPU")
    ap.add_argument("--sub-batches", type=int, default=8, help="tum: sub-batches (of distinct frames) per step")
    ap.add_argument("--streams", type=int, default=4, help="tum: independent sub-pipelines (HIP streams) a sub-batch is split over")
    ap.add_argument("--frames-4k", type=int, default=256, help="4k: frames of the whole job per step")
    ap.add_argument("--launch-4k", type=int, default=256, help="4k: frames per launch (at most the rank's share)")
    ap.add_argument("--split-4k", type=int, default=0, help="4k: handles / streams a launch is split over (0: two for shares of at most 64 frames)")
    ap.add_argument("--stereo-batch", type=int, default=256, help="configs block: stereo pairs / frame pairs per launch")
    ap.add_argument("--no-latency", action="store_true", help="skip the host-side latency / PCIe probes (profiling runs)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline (and with it the in-run parity check of the main workload)")
    ap.add_argument("--no-configs", action="store_true", help="skip the other benchmark configurations")
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
        gpu_par = out.pop("_parity_gpu", None)
        frames = out.pop("_frames", None)
        if world == 1 and args.workload == "tum":
            if args.no_cpu:
                out["cpu_baseline"] = None
                out["parity_in_run"] = None
            else:
                out["cpu_baseline"], golden = cpu_baseline(frames[:64])
                out["parity_in_run"] = parity_in_run(gpu_par, golden)
            if not args.no_configs:
                out["configs"] = run_configs(args, torch, dev, not args.no_cpu)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
