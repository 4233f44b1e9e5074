"""The device-resident frame handle: its contents against the CPU oracle (and against the host-array entry points),
and the *_frame matcher entry points against the oracle's sequential matchers on the frame's own arrays.  Integer /
index work is bit-exact; the undistorted coordinates are double-precision arithmetic rounded to float (tolerance
1e-4 px, observed identical)."""
import numpy as np
import pytest

from orb_slam2_amd import synth

pytestmark = pytest.mark.gpu

K_TUM1 = [517.306408, 516.469215, 318.643040, 255.313989]      # Examples/Monocular/TUM1.yaml:9-12
D_TUM1 = [0.262383, -0.953104, -0.005358, 0.002628, 1.163314]  # TUM1.yaml:14-18
D_NONE = [0, 0, 0, 0, 0]


def _bounds(matcher, w, h, K4, D5):
    """Frame::ComputeImageBounds (src/Frame.cc:560-597): the undistorted corners."""
    from orb_slam2_amd import api
    if D5[0] == 0:
        return np.array([0, 0, w, h], np.float32)
    c = np.zeros(4, api.KEYPOINT_DTYPE)
    c["x"] = [0, w, 0, w]; c["y"] = [0, 0, h, h]
    u = matcher.UndistortKeyPoints(c, K4, D5)
    return np.array([min(u["x"][0], u["x"][2]), min(u["y"][0], u["y"][1]), max(u["x"][1], u["x"][3]), max(u["y"][2], u["y"][3])],
                    np.float32)


@pytest.fixture(scope="module")
def matcher():
    from orb_slam2_amd.api import ORBmatcher
    return ORBmatcher(0.9, True)


@pytest.fixture(scope="module")
def seq():
    return synth.sequence(640, 480, 3, seed=1)


@pytest.fixture(scope="module")
def ex():
    from orb_slam2_amd.api import ORBextractor
    return ORBextractor(1000, 1.2, 8, 20, 7)


@pytest.fixture(scope="module")
def tum_frames(ex, matcher, seq):
    """Three monocular frames of one camera with the TUM1 distortion."""
    from orb_slam2_amd.api import Frame
    b = _bounds(matcher, 640, 480, K_TUM1, D_TUM1)
    out = []
    for img in seq:
        f = Frame(ex.max_keypoints)
        f.build_mono(ex, img, K_TUM1, D_TUM1, b)
        out.append(f)
    return out, b


def test_frame_mono_contents(ex, matcher, oracle, seq, tum_frames):
    frames, b = tum_frames
    oe = oracle.Extractor(1000)
    for img, f in zip(seq, frames):
        rk, rd = oe.extract(img)
        assert f.N == len(rk) and f.N > 900
        assert f.mvKeys.tobytes() == rk.tobytes() and np.array_equal(f.mDescriptors, rd)
        ru = oracle.undistort_keypoints(rk, K_TUM1, D_TUM1)
        un = f.mvKeysUn
        assert np.abs(un["x"] - ru["x"]).max() <= 1e-4 and np.abs(un["y"] - ru["y"]).max() <= 1e-4
        assert np.array_equal(un["octave"], rk["octave"]) and np.array_equal(un["angle"], rk["angle"])
        assert np.array_equal(un, matcher.UndistortKeyPoints(f.mvKeys, K_TUM1, D_TUM1))  # same kernel arithmetic: identical
        assert (f.mvuRight == -1).all() and (f.mvDepth == -1).all() and f.nvalid == 0
        cs, ci = f.grid()
        rcs, rci = oracle.grid_build(un.view(oracle.KEYPOINT_DTYPE), b)
        assert np.array_equal(cs, rcs) and np.array_equal(ci, rci)
        assert cs[-1] == len(ci)
    # the TUM1 model pulls the corners inwards: the bounds lie inside the image
    assert b[0] > 5 and b[2] < 635


def test_frame_outlives_the_extractor_call(ex, tum_frames, seq):
    """A frame owns its buffers: extracting other images afterwards does not disturb it."""
    frames, _ = tum_frames
    before = (frames[0].mvKeys.copy(), frames[0].mDescriptors.copy())
    from orb_slam2_amd.api import Frame
    f = Frame(ex.max_keypoints)
    f.build_mono(ex, seq[2], K_TUM1, D_NONE, [0, 0, 640, 480])
    ex(seq[1])
    frames[0]._cache = {}
    assert np.array_equal(frames[0].mvKeys, before[0]) and np.array_equal(frames[0].mDescriptors, before[1])
    assert np.array_equal(f.mvKeysUn, f.mvKeys)  # k1 == 0: a plain copy (src/Frame.cc:519-523)
    k, d = ex(seq[2])
    assert np.array_equal(f.mvKeys, k) and np.array_equal(f.mDescriptors, d)


def test_frame_rgbd(ex, matcher, oracle, seq):
    from orb_slam2_amd.api import Frame
    rng = np.random.default_rng(12)
    depth = rng.uniform(0.3, 8.0, (480, 640)).astype(np.float32)
    depth[rng.random(depth.shape) < 0.2] = 0.0
    b = _bounds(matcher, 640, 480, K_TUM1, D_TUM1)
    f = Frame(ex.max_keypoints)
    n = f.build_rgbd(ex, seq[0], depth, 40.0, K_TUM1, D_TUM1, b)
    rur, rdp = oracle.stereo_from_rgbd(f.mvKeys.view(oracle.KEYPOINT_DTYPE), f.mvKeysUn.view(oracle.KEYPOINT_DTYPE), depth, 40.0)
    assert np.array_equal(f.mvuRight, rur) and np.array_equal(f.mvDepth, rdp)
    assert f.nvalid == (rdp > 0).sum() and 500 < f.nvalid < n
    cs, ci = f.grid()
    rcs, rci = oracle.grid_build(f.mvKeysUn.view(oracle.KEYPOINT_DTYPE), b)
    assert np.array_equal(cs, rcs) and np.array_equal(ci, rci)


@pytest.mark.parametrize("w,h,nfeat,bf,fx,seed", [(752, 480, 1200, 47.90639384423901, 435.2046959714599, 3),
                                                   (1241, 376, 2000, 386.1448, 718.856, 2)])
def test_frame_stereo(oracle, w, h, nfeat, bf, fx, seed):
    """Frame::Frame stereo on the EuRoC and KITTI geometries: row table, matching, sub-pixel refinement and the
    median-based outlier rejection all on the device, against the oracle's ComputeStereoMatches."""
    from orb_slam2_amd.api import Frame, ORBextractor
    L, R, _ = synth.stereo_pair(w, h, seed=seed)
    exL = ORBextractor(nfeat, 1.2, 8, 20, 7, max_width=w, max_height=h)
    exR = ORBextractor(nfeat, 1.2, 8, 20, 7, max_width=w, max_height=h)
    f = Frame(exL.max_keypoints)
    n, nvalid = f.build_stereo(exL, exR, L, R, bf, bf / fx, [fx, fx, w / 2, h / 2], D_NONE, [0, 0, w, h])
    oe = oracle.Extractor(nfeat)
    kL, dL = oe.extract(L)
    pyrL = [oe.level(l) for l in range(8)]
    kR, dR = oe.extract(R)
    pyrR = [oe.level(l) for l in range(8)]
    assert n == len(kL) and f.mvKeys.tobytes() == kL.tobytes() and np.array_equal(f.mDescriptors, dL)
    t = oracle.tables(oracle.make_params(nfeat))
    rn, ruR, rdep = oracle.stereo_matches(kL, dL, kR, dR, pyrL, pyrR, t["scale"], t["inv_scale"], bf, bf / fx)
    assert nvalid == rn and nvalid > 100
    assert np.array_equal(f.mvuRight, ruR) and np.array_equal(f.mvDepth, rdep)
    assert (f.mvDepth > 0).sum() == nvalid
    # rebuilt with other images, the handle carries nothing over
    n2, nvalid2 = f.build_stereo(exL, exR, R, L, bf, bf / fx, [fx, fx, w / 2, h / 2], D_NONE, [0, 0, w, h])
    assert n2 == len(kR) and np.array_equal(f.mDescriptors, dR)
    rn2, ruR2, rdep2 = oracle.stereo_matches(kR, dR, kL, dL, pyrR, pyrL, t["scale"], t["inv_scale"], bf, bf / fx)
    assert nvalid2 == rn2 and np.array_equal(f.mvuRight, ruR2) and np.array_equal(f.mvDepth, rdep2)


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("th", [7.0, 25.0])
def test_search_by_projection_frame(matcher, oracle, tum_frames, mode, th):
    from orb_slam2_amd import api
    frames, b = tum_frames
    f1, f2 = frames[0], frames[1]
    k1, d1 = f1.mvKeysUn, f1.mDescriptors
    sc = oracle.tables(oracle.make_params(1000))["scale"]
    q = api.shift_queries(k1, -2.0, -1.0, th, sc, b)
    rng = np.random.default_rng(int(th) + mode)
    q["flags"] = np.where(rng.random(len(q)) < 0.1, q["flags"] & 1, q["flags"])
    q["flags"] = np.where(rng.random(len(q)) < 0.05, 0, q["flags"])
    if mode == 0:
        q["minLevel"] = k1["octave"] - 1
        q["maxLevel"] = k1["octave"]
    q = q.astype(api.WQUERY_DTYPE)
    occ = (rng.random(f2.N) < 0.1).astype(np.uint8)
    matcher.mfNNratio = 0.8
    n, assign = matcher.SearchByProjectionFrame(mode, q, d1, f2, occupied=occ, th_dist=100)
    rn, rassign = oracle.search_by_projection(mode, q.view(oracle.WQUERY_DTYPE), d1, f2.mvKeysUn.view(oracle.KEYPOINT_DTYPE),
                                              f2.mDescriptors, b, uright2=f2.mvuRight, occupied=occ, th_dist=100, nnratio=0.8,
                                              checkOri=True)
    assert n == rn and np.array_equal(assign, rassign) and (assign >= 0).sum() > 50
    # and the array entry point on the downloaded copies says the same
    n3, assign3 = matcher.SearchByProjection(mode, q, d1, f2.mvKeysUn, f2.mDescriptors, b, uright2=f2.mvuRight, occupied=occ,
                                             th_dist=100)
    assert n3 == n and np.array_equal(assign3, assign)
    matcher.mfNNratio = 0.9


def test_search_by_projection_frame_uses_uright(ex, matcher, oracle, seq):
    """An RGB-D frame as the train side: the uRight gate of :96-101 reads the frame's own mvuRight."""
    from orb_slam2_amd import api
    rng = np.random.default_rng(5)
    depth = rng.uniform(2.0, 8.0, (480, 640)).astype(np.float32)
    depth[rng.random(depth.shape) < 0.3] = 0.0
    b = np.array([0, 0, 640, 480], np.float32)
    f1, f2 = api.Frame(ex.max_keypoints), api.Frame(ex.max_keypoints)
    f1.build_rgbd(ex, seq[0], depth, 40.0, K_TUM1, D_NONE, b)
    f2.build_rgbd(ex, seq[1], depth, 40.0, K_TUM1, D_NONE, b)
    sc = oracle.tables(oracle.make_params(1000))["scale"]
    q = api.shift_queries(f1.mvKeysUn, -2.0, -1.0, 15.0, sc, b)
    q["ur"] = np.where(f1.mvuRight > 0, f1.mvuRight - 2.0 + rng.normal(0, 6.0, f1.N), -1.0).astype(np.float32)
    q = q.astype(api.WQUERY_DTYPE)
    n, assign = matcher.SearchByProjectionFrame(api.MODE_LAST_FRAME, q, f1.mDescriptors, f2)
    rn, rassign = oracle.search_by_projection(1, q.view(oracle.WQUERY_DTYPE), f1.mDescriptors, f2.mvKeysUn.view(oracle.KEYPOINT_DTYPE),
                                              f2.mDescriptors, b, uright2=f2.mvuRight, th_dist=100, nnratio=0.9, checkOri=True)
    assert n == rn and np.array_equal(assign, rassign) and n > 100
    rn0, _ = oracle.search_by_projection(1, q.view(oracle.WQUERY_DTYPE), f1.mDescriptors, f2.mvKeysUn.view(oracle.KEYPOINT_DTYPE),
                                         f2.mDescriptors, b, th_dist=100, nnratio=0.9, checkOri=True)
    assert rn0 != rn  # the gate matters in this scene


@pytest.mark.parametrize("window,ratio,ori", [(100, 0.9, True), (30, 0.7, False)])
def test_search_for_initialization_frames(oracle, tum_frames, window, ratio, ori):
    from orb_slam2_amd.api import ORBmatcher
    frames, b = tum_frames
    f1, f2 = frames[0], frames[1]
    m = ORBmatcher(ratio, ori)
    k1 = f1.mvKeysUn
    prev = np.stack([k1["x"], k1["y"]], axis=1).astype(np.float32)
    n, m12, prev_out = m.SearchForInitializationFrames(f1, f2, prev, window)
    rn, rm12, rprev = oracle.search_for_initialization(k1.view(oracle.KEYPOINT_DTYPE), f1.mDescriptors,
                                                       f2.mvKeysUn.view(oracle.KEYPOINT_DTYPE), f2.mDescriptors, b, prev, window,
                                                       ratio, ori)
    assert n == rn and np.array_equal(m12, rm12) and np.array_equal(prev_out, rprev)
    if window >= 100:
        assert n > 20
    # second round, as Tracking::MonocularInitialization does with the updated vbPrevMatched
    n2, m12b, prev2 = m.SearchForInitializationFrames(f1, frames[2], prev_out, window)
    rn2, rm12b, rprev2 = oracle.search_for_initialization(k1.view(oracle.KEYPOINT_DTYPE), f1.mDescriptors,
                                                          frames[2].mvKeysUn.view(oracle.KEYPOINT_DTYPE), frames[2].mDescriptors, b,
                                                          rprev, window, ratio, ori)
    assert n2 == rn2 and np.array_equal(m12b, rm12b) and np.array_equal(prev2, rprev2)


@pytest.mark.parametrize("th,chi2,th_dist", [(3.0, True, 50), (7.5, False, 100)])
def test_search_in_area_frame(ex, matcher, oracle, seq, tum_frames, th, chi2, th_dist):
    """Fuse / SearchBySim3 window search against a keyframe's device-resident arrays, with the KeyFrame's
    int-truncated query origin."""
    from orb_slam2_amd import api
    frames, fb = tum_frames
    f1, f2 = frames[0], frames[1]
    sc = oracle.tables(oracle.make_params(1000))["scale"]
    ib = np.trunc(fb).astype(np.float32)
    geom = np.array([fb[0], fb[1], ib[0], ib[1], np.float32(64) / (fb[2] - fb[0]), np.float32(48) / (fb[3] - fb[1])], np.float32)
    k1 = f1.mvKeysUn
    q = api.shift_queries(k1, -2.0, -1.0, th, sc, ib)
    q["minLevel"] = k1["octave"] - 1; q["maxLevel"] = k1["octave"]
    q["ur"] = -1.0
    q = q.astype(api.WQUERY_DTYPE)
    inv_s2 = (1.0 / (sc * sc)).astype(np.float32)
    for use_geom in (True, False):
        if use_geom:
            matcher.set_grid_geometry(geom)
        n, qm = matcher.SearchInAreaFrame(q, f1.mDescriptors, f2, inv_s2 if chi2 else None, chi2, th_dist)
        rn, rqm = oracle.search_in_area(q.view(oracle.WQUERY_DTYPE), f1.mDescriptors, f2.mvKeysUn.view(oracle.KEYPOINT_DTYPE),
                                        f2.mDescriptors, fb, f2.mvuRight if chi2 else None, inv_s2 if chi2 else None, chi2, th_dist,
                                        geom6=geom if use_geom else None)
        assert n == rn and np.array_equal(qm, rqm) and n > 100, use_geom


def _featvec(desc, nbits):
    node = (desc[:, 0].astype(np.int32) >> (8 - nbits))
    return {int(n): np.nonzero(node == n)[0].tolist() for n in np.unique(node)}


def test_bow_and_triangulation_frames(oracle, tum_frames):
    from orb_slam2_amd import api
    frames, _ = tum_frames
    f1, f2 = frames[0], frames[1]
    k1, d1, k2, d2 = f1.mvKeysUn, f1.mDescriptors, f2.mvKeysUn, f2.mDescriptors
    rng = np.random.default_rng(4)
    fv1, fv2 = _featvec(d1, 4), _featvec(d2, 4)
    valid1 = (rng.random(len(k1)) < 0.8).astype(np.uint8)
    m = api.ORBmatcher(0.7, True)
    got = m.SearchByBoWFrame(fv1, k1, d1, valid1, fv2, f2)
    ref = m.SearchByBoW(fv1, k1, d1, valid1, fv2, k2, d2)  # checked against the oracle in test_gpu_match.py
    qi, qs, ss, si = api.bow_segments(fv1, fv2, valid1)
    q = np.zeros(len(qi), oracle.WQUERY_DTYPE)
    q["angle"] = k1["angle"][qi]; q["minLevel"] = qs; q["flags"] = 1
    rn, rassign, rqm = oracle.search_by_bow(q, d1[qi], k2.view(oracle.KEYPOINT_DTYPE), d2, ss, si, None, 50, 0.7, True)
    assert got[0] == ref[0] == rn and rn > 20
    assert np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2])
    rm12 = np.full(len(k1), -1, np.int32); rm12[qi[rqm >= 0]] = rqm[rqm >= 0]
    assert np.array_equal(got[2], rm12)

    has1 = rng.random(len(k1)) < 0.2; has2 = rng.random(len(k2)) < 0.2
    t = np.array([2.0, 1.0], np.float32) / np.float32(np.sqrt(5.0))
    F12 = np.array([[0, 0, t[1]], [0, 0, -t[0]], [-t[1], t[0], 0]], np.float32)
    tab = oracle.tables(oracle.make_params(1000))
    scale2 = tab["scale"]; sigma2 = (scale2 * scale2).astype(np.float32)
    m6 = api.ORBmatcher(0.6, True)
    n, m12 = m6.SearchForTriangulationFrame(fv1, k1, d1, has1, None, fv2, f2, has2, False, F12, (300.0, 200.0), scale2, sigma2)
    use1 = ~has1
    qi, qs, ss, si = api.bow_segments(fv1, fv2, use1)
    q = np.zeros(len(qi), oracle.WQUERY_DTYPE)
    q["u"] = k1["x"][qi]; q["v"] = k1["y"][qi]; q["angle"] = k1["angle"][qi]; q["minLevel"] = qs; q["flags"] = 1
    rn, rqm = oracle.search_for_triangulation(q, d1[qi], k2.view(oracle.KEYPOINT_DTYPE), d2, None, ss, si, has2.astype(np.uint8),
                                              F12, 300.0, 200.0, scale2, sigma2, False, True)
    rm12 = np.full(len(k1), -1, np.int32); rm12[qi[rqm >= 0]] = rqm[rqm >= 0]
    assert n == rn and np.array_equal(m12, rm12) and n > 20


def test_frame_refusals(ex, matcher, seq):
    from orb_slam2_amd import api
    f = api.Frame(ex.max_keypoints)
    q = np.zeros(1, api.WQUERY_DTYPE)
    with pytest.raises(api.OrbGpuError, match="not been built"):
        matcher.SearchByProjectionFrame(1, q, np.zeros((1, 32), np.uint8), f)
    with pytest.raises(api.OrbGpuError, match="not been built"):
        f.grid()
    small = api.Frame(100)
    with pytest.raises(api.OrbGpuError, match="holds 100 keypoints"):
        small.build_mono(ex, seq[0], K_TUM1, D_NONE, [0, 0, 640, 480])
    with pytest.raises(api.OrbGpuError):
        f.build_mono(ex, seq[0], K_TUM1, D_NONE, [0, 0, 0, 480])  # empty bounds
    # an image without corners: an empty frame that every entry point accepts
    n = f.build_mono(ex, np.full((480, 640), 90, np.uint8), K_TUM1, D_TUM1, [0, 0, 640, 480])
    assert n == 0 and len(f.mvKeys) == 0
    cs, ci = f.grid()
    assert cs[-1] == 0 and len(ci) == 0
    nm, assign = matcher.SearchByProjectionFrame(1, q, np.zeros((1, 32), np.uint8), f)
    assert nm == 0 and len(assign) == 0
