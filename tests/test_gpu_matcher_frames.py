"""The eleven search methods of ORB_SLAM2::ORBmatcher with the reference's signatures (include/ORBmatcher.h +
include/ORBmatcherFrames.h, templates over Frame / KeyFrame / MapPoint) against their sequential CPU restatements
(oracle/orb_oracle_frames.h) on the fixture types of tests/cpp/slam_types.h.  The comparison itself is C++
(tests/cpp/test_matcher_frames.cc, built by __graft_entry__.build()); integer / index work: exact equality of every
container the reference writes."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_matcher_frames")

SCENARIOS = ["SearchByProjection(F,MapPoints)", "SearchByProjection(Cur,Last)", "SearchByProjection(Cur,KF,found)",
             "SearchByProjection(KF,Scw)", "SearchByBoW(KF,F)", "SearchByBoW(KF,KF)", "SearchForInitialization",
             "SearchForTriangulation", "Fuse(KF,MapPoints)", "Fuse(KF,Scw)", "SearchBySim3", "ComputeThreeMaxima"]


def test_frame_signature_headers_compile_warning_free(tmp_path):
    """include/ORBmatcher.h with its member templates instantiated on the fixture types: -Wall -Werror, no GPU needed."""
    out = tmp_path / "probe"
    subprocess.check_call(["g++", "-std=c++14", "-Wall", "-Werror", "-ffp-contract=off",
                           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "oracle"),
                           "-I" + os.path.join(ROOT, "tests", "cpp"), os.path.join(ROOT, "tests", "cpp", "test_matcher_frames.cc"),
                           "-L" + os.path.join(ROOT, "orb_slam2_amd"), "-L" + os.path.join(ROOT, "oracle"), "-lorbgpu",
                           "-lorb_oracle", "-o", str(out)])


@pytest.mark.gpu
def test_all_eleven_overloads_match_the_oracle():
    assert os.path.exists(EXE), "run __graft_entry__.build() first"
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.splitlines()
    assert lines[-1] == "ALL OK"
    for name in SCENARIOS:
        hits = [ln for ln in lines if ln.startswith(name)]
        assert hits and all(" OK" in ln for ln in hits), name
        # every scenario also ran with Frames / KeyFrames that carry a device-resident copy of their arrays
        dev = [ln for ln in hits if "[device frames]" in ln]
        assert len(dev) * 2 == len(hits), name


@pytest.mark.gpu
def test_the_comparison_notices_a_wrong_keyframe_grid_origin():
    """Self-check: the same program built with the GPU side binning a KeyFrame's keypoints at the int-truncated
    origin (instead of the float bounds of the Frame that filled the grid) must report mismatches, and only in
    scenarios whose image bounds are not integers."""
    r = subprocess.run([EXE + "_mutated"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 1, r.stdout + r.stderr
    bad = [ln for ln in r.stdout.splitlines() if " FAIL" in ln]
    assert bad and all(" dist" in ln.split("gpu")[0] for ln in bad), r.stdout
    # a device-resident frame carries the grid its Frame built: the wrong origin cannot reach it
    assert not any("[device frames]" in ln for ln in bad), r.stdout
