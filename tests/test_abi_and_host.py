"""CPU-side checks: the C-ABI library loads and exports every declared symbol,
fails loudly without a GPU, and the host-side helpers behave."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "orbgpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(orbgpu_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from orb_slam2_amd import api
    lib = api.load_library()
    declared = _declared_symbols()
    assert len(declared) >= 25
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(api.EXPORTS) == declared


def test_header_structs_match_python_layouts():
    from orb_slam2_amd import api
    assert api.KEYPOINT_DTYPE.itemsize == 28      # sizeof(cv::KeyPoint)
    assert api.WQUERY_DTYPE.itemsize == 32
    import ctypes
    assert ctypes.sizeof(api.Params) == 40


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from orb_slam2_amd import api
    with pytest.raises(api.OrbGpuError):
        api.ORBextractor(1000, 1.2, 8, 20, 7)
    with pytest.raises(api.OrbGpuError):
        api.ORBmatcher()


def test_descriptor_distance_host_entry():
    from orb_slam2_amd.api import ORBmatcher
    rng = np.random.default_rng(0)
    for _ in range(100):
        a = rng.integers(0, 256, 32, dtype=np.uint8); b = rng.integers(0, 256, 32, dtype=np.uint8)
        assert ORBmatcher.DescriptorDistance(a, b) == int(np.unpackbits(a ^ b).sum())
    assert ORBmatcher.DescriptorDistance(np.zeros(32, np.uint8), np.full(32, 255, np.uint8)) == 256


def test_shift_queries_semantics():
    from orb_slam2_amd import api
    k = np.zeros(3, api.KEYPOINT_DTYPE)
    k["x"] = [10, 630, 1]; k["y"] = [10, 470, 1]; k["octave"] = [0, 2, 7]; k["angle"] = [1, 2, 3]
    sc = np.array([1.2 ** i for i in range(8)], np.float32)
    q = api.shift_queries(k, -2, -1, 15, sc, [0, 0, 640, 480])
    assert list(q["minLevel"]) == [-1, 1, 6] and list(q["maxLevel"]) == [1, 3, 8]
    assert np.allclose(q["r"], 15 * sc[[0, 2, 7]])
    assert list(q["flags"]) == [3, 3, 0]  # the third lands outside the image bounds


def test_shard_frames_partitions():
    from orb_slam2_amd.dist import shard_frames
    for n in (0, 1, 7, 256, 1000):
        for world in (1, 2, 3, 8):
            got = [i for r in range(world) for i in shard_frames(n, world, r)]
            assert got == list(range(n))
            sizes = [len(shard_frames(n, world, r)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_frames(10, 2, 2)


def test_opencv_branch_of_the_headers_type_checks():
    """The ORBGPU_WITH_OPENCV branch of include/*.h (what a real ORB-SLAM2 build uses) and the eleven reference-signature
    templates on types that hold cv::Mat like the reference's: compiled with -fsyntax-only against declarations that carry
    OpenCV's signatures (tests/cpp/fake_opencv; OpenCV itself is absent from this image, so nothing is linked or run)."""
    import subprocess
    subprocess.check_call(["g++", "-std=c++14", "-Wall", "-Werror", "-fsyntax-only",
                           "-I" + os.path.join(ROOT, "tests", "cpp", "fake_opencv"), "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "opencv_branch_probe.cc")])


def test_resize_window_row_reciprocal_is_exact():
    """The resize kernel finds the row of a staged source dword as the high word of d * (floor(2^32 / P) + 1)
    (extractor_api.hip sets the multiplier per level): exact for every window the library admits (at most 64 KiB,
    i.e. d < 2^14, and any dword pitch)."""
    d = np.arange(1 << 14, dtype=np.uint64)
    for P in list(range(1, 600)) + [1023, 4095, 16383]:
        magic = np.uint64(0xFFFFFFFF // P + 1)
        assert np.array_equal((d * magic) >> np.uint64(32), d // np.uint64(P)), P
