"""GPU parity of the extractor against the CPU oracle, stage by stage and end to end.

Integer stages (pyramid, FAST candidates, quadtree selection, blur, coordinates,
octaves, responses, descriptors) must be bit-exact.  Angles: the oracle uses
glibc sinf/cosf where the device rounds double sin/cos to float, so the test
allows the tolerance the north star states (angle within 1e-4 rad, < 0.1 % of
descriptors differing) and reports the observed figures.
"""
import numpy as np
import pytest

from orb_slam2_amd import synth

pytestmark = pytest.mark.gpu


def _mk(w, h, nfeat, **kw):
    from orb_slam2_amd.api import ORBextractor
    return ORBextractor(nfeat, 1.2, 8, 20, 7, max_width=w, max_height=h, **kw)


@pytest.fixture(scope="module")
def tum_pair(oracle):
    seq = synth.sequence(640, 480, 2, seed=1)
    ex = _mk(640, 480, 1000)
    oe = oracle.Extractor(1000)
    return seq, ex, oe


def test_tables_match_oracle(oracle):
    ex = _mk(640, 480, 1000)
    t = oracle.tables(oracle.make_params(1000))
    assert np.array_equal(ex.GetScaleFactors(), t["scale"])
    assert np.array_equal(ex.GetInverseScaleFactors(), t["inv_scale"])
    assert np.array_equal(ex.GetScaleSigmaSquares(), t["sigma2"])
    assert np.array_equal(ex.GetInverseScaleSigmaSquares(), t["inv_sigma2"])
    assert np.array_equal(ex.mnFeaturesPerLevel, t["quota"])


def test_stages_bit_exact(tum_pair, oracle):
    seq, ex, oe = tum_pair
    kps, desc = ex(seq[0])
    okps, odesc = oe.extract(seq[0])
    for l in range(8):
        ref = oe.level(l)
        got = ex.pyramid_level(l)
        assert got.shape == ref.shape
        assert np.array_equal(got, ref), "pyramid level %d differs" % l
        # halo = reflect-101 of the interior
        gb = ex.pyramid_level(l, border=19)
        assert np.array_equal(gb, oracle.border101(ref, 19)), "halo of level %d differs" % l
        rb = oe.level(l, blurred=True)
        if rb is not None:
            assert np.array_equal(ex.pyramid_level(l, blurred=True), rb), "blurred level %d differs" % l
        cx, cy, cr = oracle.level_candidates(ref)
        gx, gy, gr = ex.debug_level_candidates(l)
        assert len(gx) == len(cx), "level %d: %d candidates vs oracle %d" % (l, len(gx), len(cx))
        assert np.array_equal(gx, cx.astype(np.int32)) and np.array_equal(gy, cy.astype(np.int32))
        assert np.array_equal(gr, cr.astype(np.int32))
        sel, _ = oracle.distribute_octree(cx, cy, cr, 16, ref.shape[1] - 16, 16, ref.shape[0] - 16,
                                          int(ex.mnFeaturesPerLevel[l]))
        sx, sy, sr = ex.debug_level_selected(l)
        assert len(sx) == len(sel), "level %d: %d selected vs oracle %d" % (l, len(sx), len(sel))
        assert np.array_equal(sx, cx[sel].astype(np.int32)) and np.array_equal(sy, cy[sel].astype(np.int32))
        assert np.array_equal(sr, cr[sel].astype(np.int32))
    assert len(kps) == len(okps)
    for f in ("x", "y", "size", "response", "octave", "class_id"):
        assert np.array_equal(kps[f], okps[f]), f
    dang = np.abs(kps["angle"] - okps["angle"])
    dang = np.minimum(dang, 360 - dang)
    assert dang.max() <= np.degrees(1e-4), dang.max()
    nbad = int((desc != odesc).any(axis=1).sum())
    print("angles bit-exact: %s; descriptors differing: %d of %d" % (bool((dang == 0).all()), nbad, len(desc)))
    assert nbad <= 0.001 * len(desc)


@pytest.mark.parametrize("w,h,nfeat,seed", [(640, 480, 2000, 11), (752, 480, 1200, 3), (1241, 376, 2000, 2),
                                           (320, 240, 500, 5), (251, 233, 300, 9)])
def test_end_to_end_configs(oracle, w, h, nfeat, seed):
    img = synth.sequence(w, h, 1, seed=seed)[0]
    ex = _mk(w, h, nfeat)
    oe = oracle.Extractor(nfeat)
    kps, desc = ex(img)
    okps, odesc = oe.extract(img)
    assert len(kps) == len(okps)
    for f in ("x", "y", "size", "response", "octave", "class_id"):
        assert np.array_equal(kps[f], okps[f]), f
    dang = np.abs(kps["angle"] - okps["angle"])
    assert np.minimum(dang, 360 - dang).max() <= np.degrees(1e-4)
    assert int((desc != odesc).any(axis=1).sum()) <= 0.001 * len(desc)


def _compare(ex, oe, img):
    kps, desc = ex(img)
    okps, odesc = oe.extract(img)
    assert len(kps) == len(okps)
    for f in ("x", "y", "size", "response", "octave", "class_id"):
        assert np.array_equal(kps[f], okps[f]), f
    if len(kps):
        dang = np.abs(kps["angle"] - okps["angle"])
        assert np.minimum(dang, 360 - dang).max() <= np.degrees(1e-4)
    assert int((desc != odesc).any(axis=1).sum()) <= 0.001 * len(desc)
    return kps, desc


@pytest.mark.parametrize("nfeat,scale,nlevels,ini,mn,gauss", [
    (1000, 1.2, 8, 20, 7, 1),    # the 18,34,48,56,... Gaussian taps of newer OpenCV
    (800, 1.5, 5, 20, 7, 0), (1500, 1.1, 12, 20, 7, 0), (600, 2.0, 3, 20, 7, 0), (500, 1.2, 1, 20, 7, 0),
    (1000, 1.2, 8, 40, 5, 0), (1000, 1.2, 8, 12, 12, 0), (40, 1.2, 8, 20, 7, 0), (3000, 1.2, 8, 9, 3, 0)])
def test_parameter_variants(oracle, nfeat, scale, nlevels, ini, mn, gauss):
    """Constructor arguments other than the shipped YAML settings: scale factor, level count, FAST thresholds,
    feature budget, Gaussian variant."""
    from orb_slam2_amd.api import ORBextractor
    img = synth.sequence(640, 480, 1, seed=int(nfeat + nlevels))[0]
    ex = ORBextractor(nfeat, scale, nlevels, ini, mn, max_width=640, max_height=480, gauss_variant=gauss)
    oe = oracle.Extractor(nfeat, scale, nlevels, ini, mn, gauss_variant=gauss)
    t = oracle.tables(oe.p)
    assert np.array_equal(ex.GetScaleFactors(), t["scale"]) and np.array_equal(ex.mnFeaturesPerLevel, t["quota"])
    kps, _ = _compare(ex, oe, img)
    assert len(kps) > min(nfeat, 300) * 0.5
    for l in range(nlevels):
        assert np.array_equal(ex.pyramid_level(l), oe.level(l)), "level %d" % l
        rb = oe.level(l, blurred=True)
        if rb is not None:
            assert np.array_equal(ex.pyramid_level(l, blurred=True), rb), "blurred level %d" % l


def test_handmade_corner_patterns(oracle):
    """Flat background with hand-placed structures: isolated bright and dark squares of known contrast (also on
    the seams between FAST cells and at the 16 px border of the search area), two-pixel-wide bars whose adjacent
    corners tie and suppress each other, weak structures only the fallback threshold detects."""
    img = np.full((480, 640), 100, np.uint8)
    rng = np.random.default_rng(5)

    def square(x, y, s, v):
        img[y:y + s, x:x + s] = v
    # cells of level 0 are 31 x 32 px starting at (16, 16): put structures on their seams and corners
    for j in range(0, 19):
        for i in range(0, 13):
            x, y = 16 + 31 * j, 16 + 32 * i
            kind = (i * 19 + j) % 6
            if kind == 0:
                square(x - 2, y - 2, 5, 100 + 60)      # strong bright square across a cell corner
            elif kind == 1:
                square(x + 10, y - 1, 4, 100 - 55)     # dark square across a horizontal seam
            elif kind == 2:
                square(x - 1, y + 12, 2, 100 + 90)     # 2 x 2 block: four tied corners
            elif kind == 3:
                square(x + 8, y + 9, 6, 100 + 11)      # weak: below iniTh, above minTh
            elif kind == 4:
                img[y + 5:y + 25, x + 14:x + 16] = 30  # two-pixel-wide bar: tied ends
            else:
                square(x + 5, y + 5, 3, int(rng.integers(0, 256)))
    square(14, 14, 6, 220); square(620, 460, 6, 10); square(0, 0, 12, 255); square(628, 468, 12, 0)  # at the borders
    ex = _mk(640, 480, 1000)
    oe = oracle.Extractor(1000)
    kps, _ = _compare(ex, oe, img)
    assert len(kps) > 200
    for l in range(8):
        ref = oe.level(l)
        cx, cy, cr = oracle.level_candidates(ref)
        gx, gy, gr = ex.debug_level_candidates(l)
        assert np.array_equal(gx, cx.astype(np.int32)) and np.array_equal(gy, cy.astype(np.int32))
        assert np.array_equal(gr, cr.astype(np.int32))
    # the weak squares are found only through the fallback: their cells yield responses below iniTh
    assert (kps["response"] < 20).any() and (kps["response"] >= 20).any()


def test_batch_equals_single(oracle):
    seq = synth.sequence(640, 480, 4, seed=21)
    ex = _mk(640, 480, 1000, max_batch=4)
    res = ex.extract_batch(seq)
    for i in range(4):
        k1, d1 = ex(seq[i])
        assert np.array_equal(res[i][0], k1) and np.array_equal(res[i][1], d1)
    # idempotence: same input twice gives identical output
    k2, d2 = ex(seq[3])
    assert np.array_equal(k1, k2) and np.array_equal(d1, d2)


def test_batch_of_ten_and_profiling_modes(oracle):
    """Batches of 8 or more frames use the XCD-pinned block mapping (frame axis padded to a multiple
    of 8); the stage-timing modes change the schedule (blur on the side stream or not), never the result."""
    seq = synth.sequence(640, 480, 10, seed=33)
    ex = _mk(640, 480, 1000, max_batch=10)
    ref = [ex(seq[i]) for i in range(10)]
    for mode in (0, 1, 2):
        ex.profile(mode)
        res = ex.extract_batch(seq)
        for i in range(10):
            assert np.array_equal(res[i][0], ref[i][0]) and np.array_equal(res[i][1], ref[i][1])
        if mode:
            ms, n = ex.stage_times()
            assert n == 1 and len(ms) == 5 and (np.asarray(ms) > 0).all()
    ex.profile(0)


def test_batch_of_64_dense_frames(oracle):
    """The launch shapes of large batches (four FAST cells per workgroup, the batched quadtree kernel, frames pinned
    to XCDs) on frames that fill the feature budget: every frame equals its single-frame extraction, and a sample
    equals the oracle."""
    seq = synth.sequence(640, 480, 64, seed=5, per_frame_density=True)
    ex = _mk(640, 480, 1000, max_batch=64)
    res = ex.extract_batch(seq)
    one = _mk(640, 480, 1000)
    oe = oracle.Extractor(1000)
    assert min(len(r[0]) for r in res) > 900
    for i in range(64):
        k1, d1 = one(seq[i])
        assert np.array_equal(res[i][0], k1) and np.array_equal(res[i][1], d1), "frame %d" % i
    for i in (0, 31, 63):
        okps, odesc = oe.extract(seq[i])
        assert len(okps) == len(res[i][0])
        for f in ("x", "y", "size", "response", "octave", "class_id"):
            assert np.array_equal(res[i][0][f], okps[f]), f
        assert int((res[i][1] != odesc).any(axis=1).sum()) <= 0.001 * len(odesc)


def test_batch_of_160_frames_level_walk(oracle):
    """From 128 frames per launch on, a workgroup per frame walks the pyramid levels in one launch (k_pyr_walk) instead
    of one launch per level: every level of several frames (halo rows included) equals the single-frame pyramid, and
    every frame's extraction equals its single-frame extraction.  Device-resident batch: one launch of 160."""
    from hiputil import Hip
    from orb_slam2_amd.api import KEYPOINT_DTYPE
    n, w, h = 160, 752, 480  # a width that leaves the last 128-column tile partly empty
    seq = synth.sequence(w, h, n, seed=15, per_frame_density=True)
    ex = _mk(w, h, 1200, max_batch=n)
    one = _mk(w, h, 1200)
    cap = ex.max_keypoints
    hip = Hip()
    try:
        d_img = hip.upload(seq)
        d_k, d_d, d_c = hip.alloc(n * cap * 28), hip.alloc(n * cap * 32), hip.alloc(n * 4)
        ex.extract_batch_device(d_img.value, n, w, h, w, w * h, d_k.value, d_d.value, cap, d_c.value)
        ex.sync()
        K, D, Cn = hip.download(d_k, (n, cap), KEYPOINT_DTYPE), hip.download(d_d, (n, cap, 32), np.uint8), hip.download(d_c, n, np.int32)
    finally:
        hip.free()
    for i in (0, 77, 159):
        k1, d1 = one(seq[i])
        for l in range(8):
            assert np.array_equal(ex.pyramid_level(l, frame=i, border=19)[:, 19:-19], one.pyramid_level(l, border=19)[:, 19:-19]), (i, l)
    for i in range(n):
        k1, d1 = one(seq[i])
        assert Cn[i] == len(k1) and np.array_equal(K[i, :Cn[i]], k1) and np.array_equal(D[i, :Cn[i]], d1), "frame %d" % i


def test_frame_size_changes_on_one_handle(oracle):
    """A handle sized for 640x480 is given smaller frames and the first size again, through the single-frame call and
    the batch call: the tables, buffers and the replayed launch sequence follow the frame size, and every result equals
    that of a fresh handle."""
    big = synth.sequence(640, 480, 3, seed=41)
    small = synth.sequence(320, 240, 2, seed=42)
    odd = synth.sequence(411, 277, 1, seed=43)[0]
    ex = _mk(640, 480, 1000, max_batch=2)

    def fresh(img):
        h, w = img.shape
        return _mk(w, h, 1000)(img)

    for img in (big[0], big[1], small[0], big[2], odd, small[1], big[0]):
        k, d = ex(img)
        rk, rd = fresh(img)
        assert np.array_equal(k, rk) and np.array_equal(d, rd), img.shape
    res = ex.extract_batch(small)
    for i in range(2):
        rk, rd = fresh(small[i])
        assert np.array_equal(res[i][0], rk) and np.array_equal(res[i][1], rd)
    k, d = ex(big[1])
    rk, rd = fresh(big[1])
    assert np.array_equal(k, rk) and np.array_equal(d, rd)
    ok, od = oracle.Extractor(1000).extract(odd)
    k, d = ex(odd)
    assert len(k) == len(ok) and np.array_equal(k["x"], ok["x"]) and np.array_equal(k["y"], ok["y"])


def test_degenerate_images(oracle):
    ex = _mk(640, 480, 1000)
    k, d = ex(np.zeros((0, 0), np.uint8))
    assert len(k) == 0 and d.shape == (0, 32)
    flat = np.full((480, 640), 77, np.uint8)
    k, d = ex(flat)
    assert len(k) == 0
    # noise image: every cell is full of corners, quadtree culls to the quota
    rng = np.random.default_rng(4)
    noise = rng.integers(0, 256, (480, 640), dtype=np.uint8)
    oe = oracle.Extractor(1000)
    k, d = ex(noise)
    ok, od = oe.extract(noise)
    assert len(k) == len(ok)
    assert np.array_equal(k["x"], ok["x"]) and np.array_equal(k["y"], ok["y"])
    assert np.array_equal(k["response"], ok["response"])
    assert int((d != od).any(axis=1).sum()) <= 0.001 * len(d)


def test_strided_input(oracle):
    big = synth.sequence(700, 500, 1, seed=8)[0]
    view = big[10:490, 30:670]  # 640x480 view with stride 700
    ex = _mk(640, 480, 1000)
    k1, d1 = ex(view)
    k2, d2 = ex(np.ascontiguousarray(view))
    assert np.array_equal(k1, k2) and np.array_equal(d1, d2)


def test_unsupported_sizes_report_errors():
    from orb_slam2_amd.api import ORBextractor, OrbGpuError
    with pytest.raises(OrbGpuError):
        ORBextractor(1000, 1.2, 8, 20, 7, max_width=100, max_height=100)  # top levels < 62 px
    ex = _mk(640, 480, 1000)
    with pytest.raises(OrbGpuError):
        ex(np.zeros((600, 800), np.uint8))  # larger than the handle was sized for


def test_4k_frame_8000_features(oracle):
    """4K geometry (3840x2160, 8000 features): one frame, bit-exact against the oracle."""
    img = synth.sequence(3840, 2160, 1, seed=4, n_shapes=3000)[0]
    ex = _mk(3840, 2160, 8000)
    oe = oracle.Extractor(8000)
    kps, desc = ex(img)
    okps, odesc = oe.extract(img)
    assert len(kps) == len(okps) and len(kps) > 4000
    for f in ("x", "y", "size", "response", "octave", "class_id"):
        assert np.array_equal(kps[f], okps[f]), f
    dang = np.abs(kps["angle"] - okps["angle"])
    assert np.minimum(dang, 360 - dang).max() <= np.degrees(1e-4)
    assert int((desc != odesc).any(axis=1).sum()) <= 0.001 * len(desc)
    # size-independent properties at full size: level 0 is the input, every level carries its reflect-101 halo
    assert np.array_equal(ex.pyramid_level(0), img)
    top = ex.pyramid_level(7, border=19)
    assert np.array_equal(top, np.pad(top[19:-19, 19:-19], 19, mode="reflect"))


def test_halo_columns_are_completed_on_request(oracle):
    """The extraction writes the 19 halo rows and no halo column (the blur forms the reflected columns it reads
    itself); the columns are filled when a level is first asked for with its border -- for the images of the latest
    extraction, not an earlier one."""
    seq = synth.sequence(640, 480, 3, seed=17)
    ex = _mk(640, 480, 1000, max_batch=2)
    oe = oracle.Extractor(1000)
    for img in (seq[0], seq[1]):
        ex(img)
        oe.extract(img)
        for border in (3, 4, 19, 7):
            for l in (0, 3, 7):
                assert np.array_equal(ex.pyramid_level(l, border=border), oracle.border101(oe.level(l), border)), (border, l)
    # a batch: every frame of it
    ex.extract_batch(seq[1:3])
    for f in (0, 1):
        oe.extract(seq[1 + f])
        assert np.array_equal(ex.pyramid_level(2, frame=f, border=19), oracle.border101(oe.level(2), 19))


def test_two_handles_from_two_threads(oracle):
    """The reference extracts the left and the right image in two threads on two extractor instances
    (src/Frame.cc:78-81): two handles used concurrently give what each gives alone."""
    import threading
    L, R, _ = synth.stereo_pair(752, 480, seed=3)
    exL, exR = _mk(752, 480, 1200), _mk(752, 480, 1200)
    refL, refR = exL(L), exR(R)
    out = {}

    def work(name, ex, img):
        res = []
        for _ in range(20):
            res.append(ex(img))
        out[name] = res

    tl = threading.Thread(target=work, args=("L", exL, L)); tr = threading.Thread(target=work, args=("R", exR, R))
    tl.start(); tr.start(); tl.join(); tr.join()
    for k, d in out["L"]:
        assert np.array_equal(k, refL[0]) and np.array_equal(d, refL[1])
    for k, d in out["R"]:
        assert np.array_equal(k, refR[0]) and np.array_equal(d, refR[1])


def test_randomized_sweep(oracle):
    """A fixed slice of tools/fuzz_parity.py: random sizes, settings and image kinds (noise, checkerboards,
    saturated blocks, ...) through the extractor, then SearchByProjection and SearchForInitialization of two
    extractions, all against the oracle."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(__file__), "..", "tools", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    st = {}
    assert fz.main(["--cases", "120", "--seconds", "100", "--seed", "7"], st) == 0
    # Exposure to the reference's address-ordered tie-break (src/ORBextractor.cc:906,959): ties straddle the cut on many
    # frames, but only a few percent of the keypoints depend on the order.  The GPU and the oracle agree on one order
    # (latest-created node first); against a live reference that share of the keypoints is not pinned.
    assert st["frames"] >= 100 and st["tie_sampled_kp"] > 0
    assert st["tie_dependent_kp"] <= 0.06 * st["tie_sampled_kp"], st


def test_device_call_overflow_is_reported_and_does_not_stick(oracle):
    """orbgpu_extract_batch_device with a capacity below the keypoint count truncates and raises status bit 1; the
    flags describe that call alone: a following host-pointer call on the same handle succeeds."""
    from tests.hiputil import Hip
    from orb_slam2_amd import api
    img = synth.sequence(640, 480, 1, seed=1)[0]
    ex = _mk(640, 480, 1000)
    k_ref, d_ref = ex(img)
    assert len(k_ref) > 900 and ex.status() == 0
    hip = Hip()
    try:
        cap = 100
        d_img = hip.upload(img)
        d_k = hip.alloc(cap * 28); d_d = hip.alloc(cap * 32); d_c = hip.alloc(4)
        ex.extract_batch_device(d_img.value, 1, 640, 480, 640, 640 * 480, d_k.value, d_d.value, cap, d_c.value)
        assert ex.status() == 2
        assert hip.download(d_c, 1, np.int32)[0] == cap
        k_dev = hip.download(d_k, cap, api.KEYPOINT_DTYPE)
        assert np.array_equal(k_dev, k_ref[:cap])
        k2, d2 = ex(img)  # the host path on the same handle: no stale flag
        assert np.array_equal(k2, k_ref) and np.array_equal(d2, d_ref) and ex.status() == 0
        # replayed graph after an overflowing device call as well
        ex.extract_batch_device(d_img.value, 1, 640, 480, 640, 640 * 480, d_k.value, d_d.value, cap, d_c.value)
        k3, _ = ex(img)
        assert np.array_equal(k3, k_ref)
    finally:
        hip.free()
    kps = np.zeros(ex.max_keypoints, api.KEYPOINT_DTYPE); desc = np.zeros((ex.max_keypoints, 32), np.uint8)
    n = api.C.c_int32()
    rc = ex._lib.orbgpu_extract(ex._h, api._p(img), 640, 480, 600, api._p(kps), api._p(desc), ex.max_keypoints, api.C.byref(n))
    assert rc == -1  # ORBGPU_E_INVALID: row pitch below the width


def test_chunked_host_batches_pageable_pinned_and_strided(oracle):
    """Host batches of 32 frames and more run as overlapped chunks; page-locked caller memory is used in place.
    Every variant returns what frame-by-frame calls return."""
    from orb_slam2_amd import api
    B = 80
    seq = synth.sequence(320, 240, B, seed=9, per_frame_density=True)
    one = _mk(320, 240, 500)
    ref = [one(f) for f in seq]
    ex = _mk(320, 240, 500, max_batch=B)
    cap = ex.max_keypoints

    def check(kps, desc, counts, nb):
        for i in range(nb):
            n = counts[i]
            assert n == len(ref[i][0]), i
            assert np.array_equal(kps[i, :n], ref[i][0]) and np.array_equal(desc[i, :n], ref[i][1]), i

    # pageable in, pageable out, capacity below the handle's own bound
    out = ex.extract_batch(seq)
    for i in range(B):
        assert np.array_equal(out[i][0], ref[i][0]) and np.array_equal(out[i][1], ref[i][1])
    # page-locked frames and page-locked results in place (capacity == max_keypoints), twice on the same handle
    pimg = api.pinned_empty(seq.shape, np.uint8); pimg[...] = seq
    pk = api.pinned_empty((B, cap), api.KEYPOINT_DTYPE); pd = api.pinned_empty((B, cap, 32), np.uint8)
    cnt = np.zeros(B, np.int32)
    for nb in (B, 37):
        pk.view(np.uint8)[...] = 0xEE; cnt[:] = -1
        ex.extract_batch_into(pimg[:nb], pk[:nb], pd[:nb], cnt[:nb])
        check(pk, pd, cnt, nb)
    # page-locked frames, pageable results
    k2 = np.zeros((B, cap), api.KEYPOINT_DTYPE); d2 = np.zeros((B, cap, 32), np.uint8)
    ex.extract_batch_into(pimg, k2, d2, cnt)
    check(k2, d2, cnt, B)
    # strided pageable frames (rows of a wider buffer)
    wide = np.zeros((B, 240, 352), np.uint8); wide[:, :, 16:336] = seq
    ptrs = (api.C.c_void_p * B)(*[wide[i, :, 16:].ctypes.data for i in range(B)])
    api._check(ex._lib.orbgpu_extract_batch(ex._h, ptrs, B, 320, 240, 352, api._p(k2), api._p(d2), cap, api._p(cnt)), "strided")
    check(k2, d2, cnt, B)
    # an output capacity that is too small is reported, and the handle keeps working
    small = np.zeros((B, 50), api.KEYPOINT_DTYPE); sd = np.zeros((B, 50, 32), np.uint8)
    with pytest.raises(api.OrbGpuError, match="capacity"):
        ex.extract_batch_into(seq, small, sd, cnt)
    ex.extract_batch_into(pimg, k2, d2, cnt)
    check(k2, d2, cnt, B)


def test_4k_batch_of_eight_equals_single_frames(oracle):
    """Eight 4K frames in one launch (the XCD-pinned frame / block mapping at the 4K pitch, four FAST cells per
    workgroup, the batched quadtree): each frame equals its single-frame extraction, one of them the oracle."""
    base = synth.sequence(3840, 2160, 2, seed=4, n_shapes=3000)
    rng = np.random.default_rng(8)
    frames = np.stack([np.roll(base[i % 2], (3 * i, 5 * i), (0, 1)) for i in range(8)])
    frames[5] = (frames[5].astype(np.int16) + rng.integers(-3, 4, frames[5].shape)).clip(0, 255).astype(np.uint8)
    ex = _mk(3840, 2160, 8000, max_batch=8)
    res = ex.extract_batch(frames)
    for i in range(8):
        k1, d1 = ex(frames[i])
        assert len(k1) > 4000
        assert np.array_equal(res[i][0], k1) and np.array_equal(res[i][1], d1), i
    okps, odesc = oracle.Extractor(8000).extract(frames[5])
    kps, desc = res[5]
    assert len(kps) == len(okps)
    for f in ("x", "y", "size", "response", "octave"):
        assert np.array_equal(kps[f], okps[f]), f
    assert int((desc != odesc).any(axis=1).sum()) <= 0.001 * len(desc)


def test_mono_init_extractor_in_a_batch(oracle):
    """Tracking builds a second extractor with twice the features for monocular initialisation
    (src/Tracking.cc:168): 2000 features on 640x480, batched, against the oracle and against single frames."""
    seq = synth.sequence(640, 480, 12, seed=5, per_frame_density=True)
    ex = _mk(640, 480, 2000, max_batch=12)
    assert list(ex.mnFeaturesPerLevel) == [434, 362, 302, 251, 209, 175, 145, 122]
    res = ex.extract_batch(seq)
    oe = oracle.Extractor(2000)
    for i in (0, 7, 11):
        okps, odesc = oe.extract(seq[i])
        kps, desc = res[i]
        assert len(kps) == len(okps) and len(kps) > 1500
        for f in ("x", "y", "size", "response", "octave"):
            assert np.array_equal(kps[f], okps[f]), (i, f)
        dang = np.abs(kps["angle"] - okps["angle"])
        assert np.minimum(dang, 360 - dang).max() <= np.degrees(1e-4)
        assert int((desc != odesc).any(axis=1).sum()) <= 0.001 * len(desc)
    k1, d1 = ex(seq[3])
    assert np.array_equal(res[3][0], k1) and np.array_equal(res[3][1], d1)


def test_many_roots_and_many_features_per_level(oracle):
    """Shapes beyond the on-chip quadtree storage: a strip whose search area is 37 times as wide as it is high (37
    root nodes, src/ORBextractor.cc:696) and a level quota above 4092 features (sort keys in global memory)."""
    rng = np.random.default_rng(3)
    strip = rng.integers(0, 256, (80, 2000), dtype=np.uint8)  # white noise: corners everywhere
    from orb_slam2_amd.api import ORBextractor
    ex = ORBextractor(1500, 1.2, 2, 20, 7, max_width=2000, max_height=80)
    _compare(ex, oracle.Extractor(1500, 1.2, 2, 20, 7), strip)
    big = synth.sequence(1920, 1080, 1, seed=6, n_shapes=40000)[0]
    big = (big.astype(np.int16) + rng.integers(-12, 13, big.shape)).clip(0, 255).astype(np.uint8)
    ex = ORBextractor(25000, 1.2, 8, 12, 5, max_width=1920, max_height=1080)
    assert ex.mnFeaturesPerLevel[0] > 4092
    kps, desc = ex(big)
    okps, odesc = oracle.Extractor(25000, 1.2, 8, 12, 5).extract(big)
    assert len(kps) == len(okps) and (kps["octave"] == 0).sum() > 4092
    for f in ("x", "y", "size", "response", "octave"):
        assert np.array_equal(kps[f], okps[f]), f
    assert int((desc != odesc).any(axis=1).sum()) <= 0.001 * len(desc)


def test_quadtree_tie_rule_is_locked(oracle):
    """Among quadtree nodes of equal size the partial passes take the latest-created first (the reference orders them
    by heap address, src/ORBextractor.cc:906).  On a frame where such ties straddle the cut the device must agree
    with the oracle under that rule, and the comparison must notice the other order (earliest-created first)."""
    from orb_slam2_amd.api import ORBextractor
    ex = ORBextractor(1000, 1.2, 8, 20, 7)
    oe = oracle.Extractor(1000)
    seen = 0
    for img in synth.sequence(640, 480, 4, seed=3):
        kps, desc = ex(img)
        okps, _ = oe.extract(img)
        assert oe.tie_straddles() > 0  # these frames do have ties at the cut
        assert kps.tobytes() == okps.tobytes()
        oracle.set_tie_rule(1)
        try:
            akps, _ = oe.extract(img)
        finally:
            oracle.set_tie_rule(0)
        mine = set(zip(kps["x"].tolist(), kps["y"].tolist(), kps["octave"].tolist()))
        other = set(zip(akps["x"].tolist(), akps["y"].tolist(), akps["octave"].tolist()))
        seen += len(mine - other)
    assert seen > 0  # the other order selects different keypoints: the rule is what the equality above pins


def test_fast_dense_path_when_the_survivor_queue_overflows(oracle, monkeypatch):
    """The FAST kernel's survivor queue is bounded (so that a CU holds 32 of its workgroups); a cell with more
    survivors is evaluated densely, every pixel of it.  With the queue capped at 24 entries (ORBGPU_FAST_QUEUE_CAP, a
    test hook read when a handle is created) nearly every cell takes that path, in the two-wave single-frame kernel and in
    the one-wave batch kernel: candidates (order included) and the extraction equal the oracle's, noise included."""
    monkeypatch.setenv("ORBGPU_FAST_QUEUE_CAP", "24")
    rng = np.random.default_rng(3)
    seq = synth.sequence(640, 480, 24, seed=8, per_frame_density=True)
    seq[5] = rng.integers(0, 256, (480, 640), dtype=np.uint8)
    oe = oracle.Extractor(1000)
    one = _mk(640, 480, 1000)
    for i in (0, 5):
        _compare(one, oe, seq[i])
        for level in (0, 3, 7):
            gx, gy, gr = one.debug_level_candidates(level)
            cx, cy, cr = oracle.level_candidates(oe.level(level))
            assert np.array_equal(gx, cx.astype(np.int32)) and np.array_equal(gy, cy.astype(np.int32)), (i, level)
            assert np.array_equal(gr, cr.astype(np.int32)), (i, level)
    ex = _mk(640, 480, 1000, max_batch=24)  # 24 x 815 cells: the batch kernel
    res = ex.extract_batch(seq)
    monkeypatch.delenv("ORBGPU_FAST_QUEUE_CAP")
    ref = _mk(640, 480, 1000)
    for i in range(24):
        k1, d1 = ref(seq[i])
        assert np.array_equal(res[i][0], k1) and np.array_equal(res[i][1], d1), "frame %d" % i


def test_trace_ranges_leave_the_results_alone(tmp_path):
    """ORBGPU_ROCTX=1 (roctx ranges around every stage's enqueue, looked up at run time) changes nothing but the
    trace: a child process extracts and matches with the ranges on, the parent without."""
    import os
    import subprocess
    import sys
    from orb_slam2_amd import api
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, numpy as np\n"
            "sys.path.insert(0, %r)\n"
            "from orb_slam2_amd import api, synth\n"
            "seq = synth.sequence(320, 240, 2, seed=5)\n"
            "ex = api.ORBextractor(500, 1.2, 6, 20, 7, max_width=320, max_height=240)\n"
            "k, d = ex(seq[0])\n"
            "np.save(%r, d)\n"
            "print(len(k))\n") % (root, str(tmp_path / "d.npy"))
    env = dict(os.environ, ORBGPU_ROCTX="1")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr
    seq = synth.sequence(320, 240, 2, seed=5)
    ex = api.ORBextractor(500, 1.2, 6, 20, 7, max_width=320, max_height=240)
    k, d = ex(seq[0])
    assert int(out.stdout.split()[-1]) == len(k) > 50
    assert np.array_equal(np.load(tmp_path / "d.npy"), d)
