"""CPU tests of the oracle itself: the known answers the reference states, and
independent numpy/scipy re-derivations of every primitive (same integers
expected).  The reference has no tests or golden vectors for this path, so
these are what stands behind the oracle ("parity unpinned" beyond them).
"""
import math

import numpy as np
import pytest
from scipy import ndimage

from orb_slam2_amd import synth


# --------------------------------------------------------------------------
# known answers stated in the reference
# --------------------------------------------------------------------------
def test_umax_table(oracle):
    # comment at src/ORBextractor.cc:538
    assert list(oracle.tables(oracle.make_params())["umax"]) == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]


def test_constants(oracle):
    import ctypes  # noqa: F401
    from orb_slam2_amd import api
    assert (api.TH_HIGH, api.TH_LOW, api.HISTO_LENGTH) == (100, 50, 30)  # src/ORBmatcher.cc:37-39


def test_quotas_and_level_sizes(oracle):
    p = oracle.make_params(1000)
    t = oracle.tables(p)
    assert list(t["quota"]) == [217, 181, 151, 126, 105, 87, 73, 60]
    assert int(t["quota"].sum()) == 1000
    assert list(oracle.tables(oracle.make_params(2000))["quota"]) == [434, 362, 302, 251, 209, 175, 145, 122]
    assert list(oracle.tables(oracle.make_params(1200))["quota"]) == [261, 217, 181, 151, 126, 105, 87, 72]
    sizes = [oracle.level_size(p, 640, 480, l) for l in range(8)]
    assert sizes == [(640, 480), (533, 400), (444, 333), (370, 278), (309, 231), (257, 193), (214, 161), (179, 134)]
    sizes = [oracle.level_size(p, 1241, 376, l) for l in range(8)]
    assert sizes[1] == (1034, 313) and sizes[7] == (346, 105)
    # float32 chain mvScaleFactor[i] = mvScaleFactor[i-1]*scaleFactor (src/ORBextractor.cc:476-482)
    sc = np.float32(1.0)
    for i in range(8):
        assert t["scale"][i] == sc
        assert t["inv_scale"][i] == np.float32(1.0) / sc
        assert t["sigma2"][i] == sc * sc
        sc = np.float32(sc * np.float32(1.2))


def test_hamming_kats(oracle):
    z = np.zeros(32, np.uint8); o = np.full(32, 255, np.uint8)
    assert oracle.hamming(z, o) == 256 and oracle.hamming(o, o) == 0
    for bit in (0, 7, 100, 255):
        a = z.copy(); a[bit // 8] = 1 << (bit % 8)
        assert oracle.hamming(z, a) == 1
    rng = np.random.default_rng(1)
    for _ in range(200):
        a = rng.integers(0, 256, 32, dtype=np.uint8); b = rng.integers(0, 256, 32, dtype=np.uint8)
        assert oracle.hamming(a, b) == sum(bin(int(x)).count("1") for x in (a ^ b))


def test_fast_atan2_octants(oracle):
    for deg in range(0, 360, 5):
        y, x = math.sin(math.radians(deg)) * 1000, math.cos(math.radians(deg)) * 1000
        got = oracle.fast_atan2(y, x)
        err = abs(got - deg)
        assert min(err, 360 - err) < 0.3, (deg, got)
    assert oracle.fast_atan2(0.0, 0.0) == 0.0
    assert oracle.fast_atan2(0.0, 5.0) == 0.0
    assert abs(oracle.fast_atan2(5.0, 0.0) - 90.0) < 1e-4
    assert abs(oracle.fast_atan2(0.0, -5.0) - 180.0) < 1e-4
    assert abs(oracle.fast_atan2(-5.0, 0.0) - 270.0) < 1e-4


def test_cvround_is_half_to_even(oracle):
    lib = oracle.lib()
    import ctypes as C
    for v, r in [(0.5, 0), (1.5, 2), (2.5, 2), (-0.5, 0), (-1.5, -2), (2.4999, 2), (2.5001, 3)]:
        assert lib.orc_cvround_f(C.c_float(v)) == r


# --------------------------------------------------------------------------
# independent re-derivations
# --------------------------------------------------------------------------
@pytest.fixture(scope="module")
def img():
    return synth.sequence(200, 150, 1, seed=12)[0]


def test_border_is_numpy_reflect(oracle, img):
    assert np.array_equal(oracle.border101(img, 19), np.pad(img, 19, mode="reflect"))


@pytest.mark.parametrize("variant,kernel", [(0, [18, 34, 49, 55, 49, 34, 18]), (1, [18, 34, 48, 56, 48, 34, 18])])
def test_gauss7_against_scipy(oracle, img, variant, kernel):
    # float kernel exp(-x^2/8)/sum, times 256, rounded: the 8-bit taps of variant 0
    g = np.exp(-np.arange(-3, 4) ** 2 / 8.0); g /= g.sum()
    if variant == 0:
        assert list(np.rint(g * 256).astype(int)) == kernel
    k = np.array(kernel, np.int64)
    t = ndimage.correlate1d(img.astype(np.int64), k, axis=1, mode="mirror")
    t = ndimage.correlate1d(t, k, axis=0, mode="mirror")
    ref = np.minimum((t + 32768) >> 16, 255).astype(np.uint8)
    assert np.array_equal(oracle.gauss7(img, variant), ref)


def _resize_numpy(src, dw, dh):
    """cv::resize INTER_LINEAR 8U fixed-point arithmetic, vectorised re-derivation."""
    sh, sw = src.shape

    def taps(d, s):
        scale = 1.0 / (np.float64(d) / s)
        f = ((np.arange(d) + 0.5) * scale - 0.5).astype(np.float32)
        i = np.floor(f).astype(np.int32)
        f = (f - i.astype(np.float32)).astype(np.float32)
        return i, f

    sx, fx = taps(dw, sw)
    fx = np.where((sx < 0) | (sx >= sw - 1), np.float32(0), fx)
    sx = np.clip(sx, 0, sw - 1)
    a0 = np.rint((np.float32(1) - fx) * np.float32(2048)).astype(np.int32)
    a1 = np.rint(fx * np.float32(2048)).astype(np.int32)
    sy, fy = taps(dh, sh)
    b0 = np.rint((np.float32(1) - fy) * np.float32(2048)).astype(np.int32)
    b1 = np.rint(fy * np.float32(2048)).astype(np.int32)
    y0 = np.clip(sy, 0, sh - 1); y1 = np.clip(sy + 1, 0, sh - 1)
    s = src.astype(np.int32)
    sx1 = np.minimum(sx + 1, sw - 1)
    h0 = s[y0][:, sx] * a0 + s[y0][:, sx1] * a1
    h1 = s[y1][:, sx] * a0 + s[y1][:, sx1] * a1
    out = (((b0[:, None] * (h0 >> 4)) >> 16) + ((b1[:, None] * (h1 >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)


@pytest.mark.parametrize("dw,dh", [(167, 125), (100, 75), (199, 149), (50, 40)])
def test_resize_against_numpy(oracle, img, dw, dh):
    assert np.array_equal(oracle.resize_linear(img, dw, dh), _resize_numpy(img, dw, dh))


def test_resize_of_constant_image_is_constant(oracle):
    for v in (0, 1, 127, 255):
        assert (oracle.resize_linear(np.full((60, 80), v, np.uint8), 67, 50) == v).all()


RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0),
        (-3, 1), (-2, 2), (-1, 3)]


def _fast_numpy(im, t, nonmax=True):
    """Definitional FAST-9/16: corner iff 9 contiguous ring pixels all < v-t or all > v+t; score = the
    largest t' for which it still is a corner (== A-1); strict 3x3 NMS with zero outside."""
    h, w = im.shape
    v = im[3:h - 3, 3:w - 3].astype(np.int32)
    ring = np.stack([im[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx].astype(np.int32) for dx, dy in RING])
    d = v[None] - ring
    best = np.full(v.shape, -999, np.int32)
    for s in range(16):
        arc = np.stack([d[(s + k) % 16] for k in range(9)])
        best = np.maximum(best, np.maximum(arc.min(0), (-arc).min(0)))
    score = np.zeros((h, w), np.int32)
    score[3:h - 3, 3:w - 3] = np.where(best > t, best - 1, 0)
    iscorner = np.zeros((h, w), bool)
    iscorner[3:h - 3, 3:w - 3] = best > t
    out = []
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            if not iscorner[y, x]:
                continue
            s = score[y, x]
            if nonmax:
                nb = score[y - 1:y + 2, x - 1:x + 2].copy(); nb[1, 1] = -1
                if not (s > nb.max()):
                    continue
            out.append((x, y, s))
    return out


@pytest.mark.parametrize("t", [7, 20])
def test_fast9_against_definition(oracle, img, t):
    sub = img[20:90, 30:110]
    xs, ys, sc = oracle.fast9(sub, t, True)
    assert list(zip(xs.tolist(), ys.tolist(), sc.tolist())) == _fast_numpy(sub, t, True)
    xs, ys, sc = oracle.fast9(sub, t, False)
    assert list(zip(xs.tolist(), ys.tolist(), sc.tolist())) == _fast_numpy(sub, t, False)


def test_fast9_handmade_corners(oracle):
    # isolated bright square corner on a dark background: known contrast => known score
    im = np.full((21, 21), 50, np.uint8)
    im[10:, 10:] = 150
    xs, ys, sc = oracle.fast9(im, 20, False)
    assert (10, 10) in set(zip(xs.tolist(), ys.tolist())) and sc.max() == 99   # contrast 100 -> A = 100 -> response 99
    # two identical adjacent maxima suppress each other (strict '>') -> empty result
    flat = np.full((15, 15), 100, np.uint8)
    assert len(oracle.fast9(flat, 7, True)[0]) == 0
    blob = flat.copy(); blob[7, 7] = 200; blob[7, 8] = 200
    x1, _, s1 = oracle.fast9(blob, 7, False)
    x2, _, _ = oracle.fast9(blob, 7, True)
    assert len(x1) >= 2 and len(x2) < len(x1)
    # threshold-free score map agrees with per-threshold detection
    sm = oracle.fast_score_map(im)
    for t in (7, 20, 60):
        xs, ys, sc = oracle.fast9(im, t, False)
        mask = sm >= t
        mask[:3] = mask[-3:] = False; mask[:, :3] = mask[:, -3:] = False
        assert set(zip(xs.tolist(), ys.tolist())) == set(zip(*np.nonzero(mask)[::-1]))


def test_level_candidates_cells(oracle):
    im = synth.sequence(300, 220, 1, seed=5)[0]
    xs, ys, rs = oracle.level_candidates(im, 20, 7)
    assert len(xs) > 50
    # coordinates relative to (16,16), inside the FAST-able area, at least 19 px from every edge
    assert xs.min() >= 3 and ys.min() >= 3 and xs.max() < 300 - 32 - 3 and ys.max() < 220 - 32 - 3
    assert rs.min() >= 7
    # every candidate is a corner of the threshold-free map with exactly that response
    sm = oracle.fast_score_map(im)
    assert all(sm[int(y) + 16, int(x) + 16] == int(r) for x, y, r in zip(xs, ys, rs))
    # order: cell row-major, inside a cell row-major
    w_, h_ = 300 - 32, 220 - 32
    ncols, nrows = int(w_ / 30), int(h_ / 30)
    wc, hc = math.ceil(w_ / ncols), math.ceil(h_ / nrows)
    keys = [(int((y - 3) // hc), int((x - 3) // wc), int(y), int(x)) for x, y in zip(xs, ys)]
    assert keys == sorted(keys)
    # a flat image yields nothing; lowering both thresholds never yields fewer candidates per image
    assert len(oracle.level_candidates(np.full((220, 300), 9, np.uint8))[0]) == 0


def test_distribute_octree_properties(oracle):
    rng = np.random.default_rng(3)
    W, H = 608, 448
    pts = set()
    while len(pts) < 3000:
        pts.add((int(rng.integers(3, W - 3)), int(rng.integers(3, H - 3))))
    pts = np.array(sorted(pts), np.float32)
    resp = rng.integers(7, 200, len(pts)).astype(np.float32)
    for N in (1, 50, 217, 1000, 5000):
        sel, _ = oracle.distribute_octree(pts[:, 0], pts[:, 1], resp, 16, 16 + W, 16, 16 + H, N)
        assert len(set(sel.tolist())) == len(sel)
        assert len(sel) <= max(N + 3, 4)
        if N <= len(pts):
            assert len(sel) >= min(N, len(pts)) or N == 1
        else:
            assert len(sel) == len(pts)        # every point ends alone in a node
    # all points in one quadrant: the first split leaves one node, the node count does not
    # grow, and the loop condition `size == prevSize` (src/ORBextractor.cc:874) stops the
    # subdivision -- a single keypoint survives (a quirk of the reference, kept)
    q = pts[(pts[:, 0] < 100) & (pts[:, 1] < 100)]
    assert len(q) > 20
    sel, _ = oracle.distribute_octree(q[:, 0], q[:, 1], np.ones(len(q), np.float32), 16, 16 + W, 16, 16 + H, 20)
    assert list(sel) == [0]
    # ties in response: the first point of a node (input order) wins
    two = np.array([[10, 10], [11, 10]], np.float32)
    sel, _ = oracle.distribute_octree(two[:, 0], two[:, 1], np.array([5, 5], np.float32), 16, 16 + W, 16, 16 + H, 1)
    assert list(sel) == [0]
    # KITTI-like aspect ratio: 4 root nodes, result order = list order (last created first)
    Wk, Hk = 1209, 344
    pk = np.array([[50, 50], [400, 60], [700, 70], [1100, 80]], np.float32)
    sel, _ = oracle.distribute_octree(pk[:, 0], pk[:, 1], np.ones(4, np.float32), 16, 16 + Wk, 16, 16 + Hk, 4)
    assert sorted(sel.tolist()) == [0, 1, 2, 3]


def test_ic_angle_against_numpy(oracle, img):
    umax = oracle.tables(oracle.make_params())["umax"]
    for cx, cy in [(40, 40), (100, 75), (160, 100), (30, 120)]:
        m10 = m01 = 0
        for v in range(-15, 16):
            um = int(umax[abs(v)])
            row = img[cy + v, cx - um:cx + um + 1].astype(np.int64)
            m10 += int((np.arange(-um, um + 1) * row).sum())
            m01 += v * int(row.sum())
        assert oracle.ic_angle(img, cx, cy, umax) == oracle.fast_atan2(float(m01), float(m10))
    assert sum(2 * int(u) + 1 for u in umax[1:]) * 2 + 2 * int(umax[0]) + 1 == 749  # patch size


def test_brief_against_python(oracle, img):
    pat = np.loadtxt(__import__("os").path.join(__import__("os").path.dirname(__file__), "..", "oracle", "brief_pattern.inc"),
                     delimiter=",", comments="//", usecols=range(16)).astype(np.int32).reshape(-1, 4)
    assert pat.shape == (256, 4) and np.abs(pat).max() <= 13
    bl = oracle.gauss7(img)
    for cx, cy, ang in [(60, 60, 0.0), (100, 70, 33.3), (120, 90, 271.5)]:
        a32 = np.float32(ang) * np.float32(math.pi / 180.0)
        a, b = np.float32(math.cos(a32)), np.float32(math.sin(a32))
        bits = []
        for x0, y0, x1, y1 in pat:
            def tap(x, y):
                r = int(np.rint(np.float32(x) * b + np.float32(y) * a)); c = int(np.rint(np.float32(x) * a - np.float32(y) * b))
                return int(bl[cy + r, cx + c])
            bits.append(1 if tap(x0, y0) < tap(x1, y1) else 0)
        ref = np.packbits(np.array(bits, np.uint8), bitorder="little")
        assert np.array_equal(oracle.brief(bl, cx, cy, ang), ref)


def test_extract_is_consistent_with_its_stages(oracle):
    im = synth.sequence(320, 240, 1, seed=3)[0]
    oe = oracle.Extractor(500)
    kps, desc = oe.extract(im)
    assert 400 <= len(kps) <= 500 + 8 * 3
    assert np.array_equal(oe.level(0), im)
    off = 0
    t = oracle.tables(oe.p)
    for l in range(8):
        n = oe.level_count(l)
        k = kps[off:off + n]
        assert (k["octave"] == l).all() and (k["class_id"] == -1).all()
        assert (k["size"] == np.float32(int(31 * t["scale"][l]))).all()
        lv = oe.level(l)
        xs = np.rint(k["x"] / t["scale"][l]); ys = np.rint(k["y"] / t["scale"][l])
        assert xs.min() >= 19 and ys.min() >= 19 and xs.max() < lv.shape[1] - 19 and ys.max() < lv.shape[0] - 19
        if l > 0:
            prev = oe.level(l - 1)
            assert np.array_equal(lv, oracle.resize_linear(prev, lv.shape[1], lv.shape[0]))
        assert np.array_equal(oe.level(l, blurred=True), oracle.gauss7(lv))
        off += n
    assert off == len(kps)
    assert ((kps["angle"] >= 0) & (kps["angle"] < 360)).all()
    assert oe.extract(np.zeros((0, 0), np.uint8))[0].size == 0


# --------------------------------------------------------------------------
# matchers: independent Python restatements on small inputs
# --------------------------------------------------------------------------
def test_bruteforce_against_numpy(oracle):
    Q = synth.descriptors(40, 1); T = synth.descriptors(90, 2)
    T[10] = T[50]; Q[5] = T[50]
    d = np.unpackbits(Q[:, None, :] ^ T[None, :, :], axis=2).sum(2)
    bi, b1, b2 = oracle.bruteforce(Q, T)
    assert np.array_equal(bi, d.argmin(1))          # argmin returns the first minimum
    assert np.array_equal(b1, d.min(1))
    assert np.array_equal(b2, np.sort(d, axis=1)[:, 1])
    assert bi[5] == 10


def _grid_py(kps, b):
    invw = np.float32(64) / (np.float32(b[2]) - np.float32(b[0])); invh = np.float32(48) / (np.float32(b[3]) - np.float32(b[1]))
    cells = {}
    for i, k in enumerate(kps):
        px = int(math.floor(float((k["x"] - np.float32(b[0])) * invw) + 0.5)); py = int(math.floor(float((k["y"] - np.float32(b[1])) * invh) + 0.5))
        if 0 <= px < 64 and 0 <= py < 48:
            cells.setdefault((px, py), []).append(i)
    return cells, invw, invh


def _area_py(kps, cells, invw, invh, b, x, y, r, lo, hi):
    x, y, r = np.float32(x), np.float32(y), np.float32(r)
    c0 = max(0, int(math.floor((x - b[0] - r) * invw))); c1 = min(63, int(math.ceil((x - b[0] + r) * invw)))
    r0 = max(0, int(math.floor((y - b[1] - r) * invh))); r1 = min(47, int(math.ceil((y - b[1] + r) * invh)))
    if c0 >= 64 or c1 < 0 or r0 >= 48 or r1 < 0:
        return []
    out = []
    check = lo > 0 or hi >= 0
    for ix in range(c0, c1 + 1):
        for iy in range(r0, r1 + 1):
            for i in cells.get((ix, iy), []):
                k = kps[i]
                if check and (k["octave"] < lo or (hi >= 0 and k["octave"] > hi)):
                    continue
                if abs(k["x"] - x) < r and abs(k["y"] - y) < r:
                    out.append(i)
    return out


@pytest.fixture(scope="module")
def two_frames(oracle):
    seq = synth.sequence(320, 240, 2, seed=3)
    oe = oracle.Extractor(500)
    return [oe.extract(f) for f in seq]


def test_grid_and_area_against_python(oracle, two_frames):
    k2 = two_frames[1][0]
    b = np.array([0, 0, 320, 240], np.float32)
    cs, ci = oracle.grid_build(k2, b)
    cells, invw, invh = _grid_py(k2, b)
    for (px, py), lst in cells.items():
        c = px * 48 + py
        assert list(ci[cs[c]:cs[c + 1]]) == lst
    assert cs[-1] == sum(len(v) for v in cells.values())
    rng = np.random.default_rng(0)
    for _ in range(60):
        x, y, r = rng.uniform(-20, 340), rng.uniform(-20, 260), rng.uniform(1, 80)
        lo, hi = [(-1, -1), (0, 0), (1, 3), (2, -1)][int(rng.integers(0, 4))]
        got = oracle.features_in_area(k2, b, cs, ci, x, y, r, lo, hi)
        assert list(got) == _area_py(k2, cells, invw, invh, b, x, y, r, lo, hi)


def _ham(a, b):
    return int(np.unpackbits(a ^ b).sum())


def _three_max(sizes):
    m1 = m2 = m3 = 0; i1 = i2 = i3 = -1
    for i, s in enumerate(sizes):
        if s > m1:
            m3, m2, m1, i3, i2, i1 = m2, m1, s, i2, i1, i
        elif s > m2:
            m3, m2, i3, i2 = m2, s, i2, i
        elif s > m3:
            m3, i3 = s, i
    if m2 < np.float32(0.1) * np.float32(m1):
        i2 = i3 = -1
    elif m3 < np.float32(0.1) * np.float32(m1):
        i3 = -1
    return i1, i2, i3


def _rot_bin(a1, a2):
    rot = np.float32(a1) - np.float32(a2)
    if rot < 0:
        rot = np.float32(rot + np.float32(360))
    v = float(np.float32(rot * np.float32(1.0 / 30)))
    b = int(math.floor(v + 0.5))
    return 0 if b == 30 else b


def test_three_maxima(oracle):
    rng = np.random.default_rng(2)
    for _ in range(200):
        s = rng.integers(0, 12, 30) * (rng.random(30) < 0.4)
        assert oracle.three_maxima(s) == _three_max(list(s))
    assert oracle.three_maxima(np.zeros(30, int)) == (-1, -1, -1)
    assert oracle.three_maxima([100] + [5] * 29) == (0, -1, -1)


def test_search_by_projection_last_frame_against_python(oracle, two_frames):
    from orb_slam2_amd import api
    (k1, d1), (k2, d2) = two_frames
    b = np.array([0, 0, 320, 240], np.float32)
    sc = oracle.tables(oracle.make_params(500))["scale"]
    q = api.shift_queries(k1, -2.0, -1.0, 15.0, sc, b)
    n, assign = oracle.search_by_projection(1, q.view(oracle.WQUERY_DTYPE), d1, k2, d2, b, th_dist=100, checkOri=True)
    cells, invw, invh = _grid_py(k2, b)
    occ = np.zeros(len(k2), bool); ref = np.full(len(k2), -1); cnt = 0
    hist = [[] for _ in range(30)]
    for i, qi in enumerate(q):
        if not qi["flags"] & 1:
            continue
        best, bidx = 256, -1
        for t in _area_py(k2, cells, invw, invh, b, qi["u"], qi["v"], qi["r"], qi["minLevel"], qi["maxLevel"]):
            if occ[t]:
                continue
            d = _ham(d1[i], d2[t])
            if d < best:
                best, bidx = d, t
        if best <= 100:
            ref[bidx] = i; occ[bidx] = True; cnt += 1
            hist[_rot_bin(k1[i]["angle"], k2[bidx]["angle"])].append(bidx)
    keep = _three_max([len(h) for h in hist])
    for bi_, h in enumerate(hist):
        if bi_ not in keep:
            for t in h:
                ref[t] = -2; cnt -= 1  # cleared by the rotation filter (NULL in the reference), not merely untouched
    assert n == cnt and np.array_equal(assign, ref) and cnt > 100


@pytest.mark.parametrize("strict", [False, True])
def test_search_by_bow_against_python(oracle, two_frames, strict):
    """The reference's loops over the two FeatureVectors (src/ORBmatcher.cc:175-264, 596-683) written out in
    Python on {node: [indices]} maps, against the oracle's segment form."""
    from orb_slam2_amd import api
    (k1, d1), (k2, d2) = two_frames
    node1 = d1[:, 0] >> 5; node2 = d2[:, 0] >> 5
    fv1 = {int(n): np.nonzero(node1 == n)[0].tolist() for n in np.unique(node1)}
    fv2 = {int(n): np.nonzero(node2 == n)[0].tolist() for n in np.unique(node2)}
    fv1.pop(sorted(fv1)[2]); fv2.pop(sorted(fv2)[0])
    rng = np.random.default_rng(5)
    valid1 = rng.random(len(k1)) < 0.85
    has2 = rng.random(len(k2)) < 0.9 if strict else np.ones(len(k2), bool)
    ratio = 0.8
    # reference form
    m12 = np.full(len(k1), -1); m21 = np.full(len(k2), -1); cnt = 0
    hist = [[] for _ in range(30)]
    n1s, n2s = sorted(fv1), sorted(fv2)
    a = b = 0
    while a < len(n1s) and b < len(n2s):
        if n1s[a] == n2s[b]:
            for i1 in fv1[n1s[a]]:
                if not valid1[i1]:
                    continue
                best1, bidx, best2 = 256, -1, 256
                for i2 in fv2[n2s[b]]:
                    if m21[i2] >= 0 or not has2[i2]:
                        continue
                    d = _ham(d1[i1], d2[i2])
                    if d < best1:
                        best2, best1, bidx = best1, d, i2
                    elif d < best2:
                        best2 = d
                ok = best1 < 50 if strict else best1 <= 50
                if ok and np.float32(best1) < np.float32(ratio) * np.float32(best2):
                    m12[i1] = bidx; m21[bidx] = i1; cnt += 1
                    hist[_rot_bin(k1[i1]["angle"], k2[bidx]["angle"])].append(i1)
            a += 1; b += 1
        elif n1s[a] < n2s[b]:
            while a < len(n1s) and n1s[a] < n2s[b]:
                a += 1
        else:
            while b < len(n2s) and n2s[b] < n1s[a]:
                b += 1
    keep = _three_max([len(h) for h in hist])
    for bi_, h in enumerate(hist):
        if bi_ not in keep:
            for i1 in h:
                m21[m12[i1]] = -1; m12[i1] = -1; cnt -= 1
    # oracle form
    qi, qs, ss, si = api.bow_segments(fv1, fv2, valid1)
    q = np.zeros(len(qi), oracle.WQUERY_DTYPE)
    q["angle"] = k1["angle"][qi]; q["minLevel"] = qs; q["flags"] = 1
    n, assign, qm = oracle.search_by_bow(q, d1[qi], k2, d2, ss, si, (~has2).astype(np.uint8) if strict else None,
                                         49 if strict else 50, ratio, True)
    om12 = np.full(len(k1), -1); om12[qi[qm >= 0]] = qm[qm >= 0]
    om21 = np.where(assign >= 0, qi[np.clip(assign, 0, None)], -1)
    assert n == cnt and cnt > 20
    assert np.array_equal(om12, m12) and np.array_equal(om21, m21)


def test_search_for_triangulation_against_python(oracle, two_frames):
    """src/ORBmatcher.cc:708-874 with CheckDistEpipolarLine (:140-157) written out in Python (float32 steps)."""
    from orb_slam2_amd import api
    f32 = np.float32
    (k1, d1), (k2, d2) = two_frames
    node1 = d1[:, 0] >> 6; node2 = d2[:, 0] >> 6
    fv1 = {int(n): np.nonzero(node1 == n)[0].tolist() for n in np.unique(node1)}
    fv2 = {int(n): np.nonzero(node2 == n)[0].tolist() for n in np.unique(node2)}
    rng = np.random.default_rng(9)
    has1 = rng.random(len(k1)) < 0.2; has2 = rng.random(len(k2)) < 0.2
    ur1 = np.where(rng.random(len(k1)) < 0.5, 1.0, -1.0).astype(f32); ur2 = np.where(rng.random(len(k2)) < 0.5, 1.0, -1.0).astype(f32)
    t = np.array([2.0, 1.0], f32) / f32(np.sqrt(5.0))
    F = np.array([[0, 0, t[1]], [0, 0, -t[0]], [-t[1], t[0], 0]], f32)
    ex, ey = f32(150.0), f32(100.0)
    scale2 = oracle.tables(oracle.make_params(500))["scale"]; sigma2 = (scale2 * scale2).astype(f32)
    m12 = np.full(len(k1), -1); cnt = 0
    hist = [[] for _ in range(30)]
    for node in sorted(set(fv1) & set(fv2)):
        for i1 in fv1[node]:
            if has1[i1]:
                continue
            st1 = ur1[i1] >= 0
            x1, y1 = k1[i1]["x"], k1[i1]["y"]
            best, bidx = 50, -1
            for i2 in fv2[node]:
                if has2[i2]:
                    continue
                st2 = ur2[i2] >= 0
                d = _ham(d1[i1], d2[i2])
                if d > 50 or d > best:
                    continue
                x2, y2, o2 = k2[i2]["x"], k2[i2]["y"], k2[i2]["octave"]
                if not st1 and not st2:
                    dx, dy = f32(ex - x2), f32(ey - y2)
                    if f32(f32(dx * dx) + f32(dy * dy)) < f32(f32(100) * scale2[o2]):
                        continue
                a = f32(f32(f32(x1 * F[0, 0]) + f32(y1 * F[1, 0])) + F[2, 0])
                b = f32(f32(f32(x1 * F[0, 1]) + f32(y1 * F[1, 1])) + F[2, 1])
                c = f32(f32(f32(x1 * F[0, 2]) + f32(y1 * F[1, 2])) + F[2, 2])
                num = f32(f32(f32(a * x2) + f32(b * y2)) + c)
                den = f32(f32(a * a) + f32(b * b))
                if den == 0:
                    continue
                if float(f32(f32(num * num) / den)) < 3.84 * float(sigma2[o2]):
                    best, bidx = d, i2
            if bidx >= 0:
                m12[i1] = bidx; cnt += 1
                hist[_rot_bin(k1[i1]["angle"], k2[bidx]["angle"])].append(i1)
    keep = _three_max([len(h) for h in hist])
    for bi_, h in enumerate(hist):
        if bi_ not in keep:
            for i1 in h:
                m12[i1] = -1; cnt -= 1
    qi, qs, ss, si = api.bow_segments(fv1, fv2, ~has1)
    q = np.zeros(len(qi), oracle.WQUERY_DTYPE)
    q["u"] = k1["x"][qi]; q["v"] = k1["y"][qi]; q["angle"] = k1["angle"][qi]; q["minLevel"] = qs
    q["flags"] = 1 + 4 * (ur1[qi] >= 0)
    n, qm = oracle.search_for_triangulation(q, d1[qi], k2, d2, ur2, ss, si, has2.astype(np.uint8), F, ex, ey, scale2, sigma2,
                                            False, True)
    om12 = np.full(len(k1), -1); om12[qi[qm >= 0]] = qm[qm >= 0]
    assert n == cnt and cnt > 10 and np.array_equal(om12, m12)


@pytest.mark.parametrize("chi2", [False, True])
def test_search_in_area_against_python(oracle, two_frames, chi2):
    """Candidate loops of Fuse (src/ORBmatcher.cc:929-979, with the chi-square gate) and of Fuse(Scw) / SearchBySim3
    (:1097-1115, 1240-1256) in Python on the grid restatement."""
    from orb_slam2_amd import api
    f32 = np.float32
    (k1, d1), (k2, d2) = two_frames
    b = np.array([0, 0, 320, 240], f32)
    sc = oracle.tables(oracle.make_params(500))["scale"]
    q = api.shift_queries(k1, -2.0, -1.0, 4.0, sc, b)
    q["minLevel"] = k1["octave"] - 1; q["maxLevel"] = k1["octave"]; q["ur"] = q["u"] - 3.0
    rng = np.random.default_rng(2)
    ur2 = np.where(rng.random(len(k2)) < 0.5, k2["x"] - 3.5, -1.0).astype(f32)
    inv_s2 = (1.0 / (sc * sc)).astype(f32)
    n, qm = oracle.search_in_area(q.view(oracle.WQUERY_DTYPE), d1, k2, d2, b, ur2, inv_s2, chi2, 50)
    cells, invw, invh = _grid_py(k2, b)
    ref = np.full(len(q), -1); cnt = 0
    for i, qi in enumerate(q):
        if not qi["flags"] & 1:
            continue
        best, bidx = 256, -1
        for t in _area_py(k2, cells, invw, invh, b, qi["u"], qi["v"], qi["r"], -1, -1):
            lvl = k2[t]["octave"]
            if lvl < qi["minLevel"] or lvl > qi["maxLevel"]:
                continue
            if chi2:
                ex, ey = f32(qi["u"] - k2[t]["x"]), f32(qi["v"] - k2[t]["y"])
                if ur2[t] >= 0:
                    er = f32(qi["ur"] - ur2[t])
                    e2 = f32(f32(f32(ex * ex) + f32(ey * ey)) + f32(er * er))
                    if float(f32(e2 * inv_s2[lvl])) > 7.8:
                        continue
                else:
                    e2 = f32(f32(ex * ex) + f32(ey * ey))
                    if float(f32(e2 * inv_s2[lvl])) > 5.99:
                        continue
            d = _ham(d1[i], d2[t])
            if d < best:
                best, bidx = d, t
        if best <= 50:
            ref[i] = bidx; cnt += 1
    assert n == cnt and cnt > 50 and np.array_equal(qm, ref)


def test_keyframe_projection_overload_is_mode_keyframe(oracle, two_frames):
    """SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th) (src/ORBmatcher.cc:362-398): nearest unmatched
    keypoint of the window with level in [predicted-1, predicted], accepted at <= TH_LOW, a match makes the keypoint
    unavailable to later points, no rotation check -- i.e. mode 2 of orc_search_by_projection with th_dist 50."""
    from orb_slam2_amd import api
    (k1, d1), (k2, d2) = two_frames
    b = np.array([0, 0, 320, 240], np.float32)
    sc = oracle.tables(oracle.make_params(500))["scale"]
    q = api.shift_queries(k1, -2.0, -1.0, 10.0, sc, b)
    q["minLevel"] = k1["octave"] - 1; q["maxLevel"] = k1["octave"]
    rng = np.random.default_rng(6)
    already = rng.random(len(k2)) < 0.15
    n, assign = oracle.search_by_projection(2, q.view(oracle.WQUERY_DTYPE), d1, k2, d2, b, occupied=already.astype(np.uint8),
                                            th_dist=50, checkOri=False)
    cells, invw, invh = _grid_py(k2, b)
    matched = already.copy(); ref = np.full(len(k2), -1); cnt = 0
    for i, qi in enumerate(q):
        if not qi["flags"] & 1:
            continue
        best, bidx = 256, -1
        for t in _area_py(k2, cells, invw, invh, b, qi["u"], qi["v"], qi["r"], -1, -1):
            if matched[t]:
                continue
            if k2[t]["octave"] < qi["minLevel"] or k2[t]["octave"] > qi["maxLevel"]:
                continue
            d = _ham(d1[i], d2[t])
            if d < best:
                best, bidx = d, t
        if best <= 50:
            ref[bidx] = i; matched[bidx] = True; cnt += 1
    assert n == cnt and cnt > 50 and np.array_equal(assign, ref)


def test_distinctive_descriptors_against_numpy(oracle):
    """src/MapPoint.cc:297-334 with numpy: full distance matrix, sorted rows, element int(0.5*(N-1)), first minimum."""
    rng = np.random.default_rng(21)
    sizes = [1, 2, 5, 8, 33, 0, 12]
    start = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    desc = rng.integers(0, 256, (start[-1], 32), dtype=np.uint8)
    desc[8:16] = desc[8] ^ (rng.random((8, 32)) < 0.05).astype(np.uint8)  # near-duplicates: ties in the medians
    bi, bm = oracle.distinctive_descriptors(desc, start)
    for p, n in enumerate(sizes):
        if n == 0:
            assert bi[p] == -1
            continue
        D = desc[start[p]:start[p + 1]]
        dist = np.unpackbits(D[:, None, :] ^ D[None, :, :], axis=2).sum(axis=2)
        med = np.sort(dist, axis=1)[:, int(0.5 * (n - 1))]
        assert bi[p] == int(np.argmin(med)) and bm[p] == med.min()


def test_undistort_keypoints_inverts_the_distortion_model(oracle, two_frames):
    """Property check of the restated cv::undistortPoints: pushing the result through the forward radial + tangential
    model must give the original pixel back (the five iterations leave well under 0.01 px here)."""
    k = two_frames[0][0]
    fx, fy, cx, cy = 258.6, 258.2, 159.3, 127.6                       # TUM1.yaml:9-12 scaled to 320x240
    k1, k2, p1, p2, k3 = 0.262383, -0.953104, -0.005358, 0.002628, 1.163314
    un = oracle.undistort_keypoints(k, [fx, fy, cx, cy], [k1, k2, p1, p2, k3])
    x = (un["x"].astype(np.float64) - cx) / fx; y = (un["y"].astype(np.float64) - cy) / fy
    r2 = x * x + y * y
    rad = 1 + k1 * r2 + k2 * r2 * r2 + k3 * r2 ** 3
    xd = x * rad + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * rad + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    assert np.abs(xd * fx + cx - k["x"]).max() < 0.01 and np.abs(yd * fy + cy - k["y"]).max() < 0.01
    assert np.abs(un["x"] - k["x"]).max() > 0.5
    assert np.array_equal(oracle.undistort_keypoints(k, [fx, fy, cx, cy], [0, 0, 0, 0, 0]), k)


def test_stereo_from_rgbd_against_numpy(oracle, two_frames):
    k = two_frames[0][0]
    ku = k.copy(); ku["x"] -= 0.5
    rng = np.random.default_rng(22)
    depth = rng.uniform(0.0, 5.0, (240, 320)).astype(np.float32)
    depth[depth < 1.0] = 0.0
    ur, dp = oracle.stereo_from_rgbd(k, ku, depth, 30.0)
    d = depth[k["y"].astype(np.int32), k["x"].astype(np.int32)]
    ok = d > 0
    assert np.array_equal(dp, np.where(ok, d, np.float32(-1)))
    ref = np.where(ok, ku["x"] - np.float32(30.0) / np.where(ok, d, np.float32(1)), np.float32(-1)).astype(np.float32)
    assert np.array_equal(ur, ref)


def test_search_for_initialization_against_python(oracle, two_frames):
    (k1, d1), (k2, d2) = two_frames
    b = np.array([0, 0, 320, 240], np.float32)
    prev = np.stack([k1["x"], k1["y"]], axis=1).astype(np.float32)
    n, m12, prev_out = oracle.search_for_initialization(k1, d1, k2, d2, b, prev, 60, 0.9, True)
    cells, invw, invh = _grid_py(k2, b)
    ref12 = np.full(len(k1), -1); m21 = np.full(len(k2), -1); md = np.full(len(k2), 2 ** 31 - 1, np.int64); cnt = 0
    hist = [[] for _ in range(30)]
    for i in range(len(k1)):
        if k1[i]["octave"] > 0:
            continue
        b1 = b2 = 2 ** 31 - 1; bidx = -1
        for t in _area_py(k2, cells, invw, invh, b, prev[i, 0], prev[i, 1], 60, 0, 0):
            d = _ham(d1[i], d2[t])
            if md[t] <= d:
                continue
            if d < b1:
                b2, b1, bidx = b1, d, t
            elif d < b2:
                b2 = d
        if b1 <= 50 and np.float32(b1) < np.float32(b2) * np.float32(0.9):
            if m21[bidx] >= 0:
                ref12[m21[bidx]] = -1; cnt -= 1
            ref12[i] = bidx; m21[bidx] = i; md[bidx] = b1; cnt += 1
            hist[_rot_bin(k1[i]["angle"], k2[bidx]["angle"])].append(i)
    keep = _three_max([len(h) for h in hist])
    for bi_, h in enumerate(hist):
        if bi_ not in keep:
            for i in h:
                if ref12[i] >= 0:
                    ref12[i] = -1; cnt -= 1
    assert n == cnt and np.array_equal(m12, ref12) and cnt > 20
    moved = ref12 >= 0
    assert np.array_equal(prev_out[moved, 0], k2["x"][ref12[moved]])
    assert np.array_equal(prev_out[~moved], prev[~moved])


def test_stereo_matches_on_synthetic_pair(oracle):
    L, R, disp = synth.stereo_pair(400, 300, seed=7, dmax=40)
    oeL, oeR = oracle.Extractor(600), oracle.Extractor(600)
    kL, dL = oeL.extract(L); kR, dR = oeR.extract(R)
    t = oracle.tables(oeL.p)
    pyrL = [oeL.level(l) for l in range(8)]; pyrR = [oeR.level(l) for l in range(8)]
    bf, b = 40.0, 0.1   # max disparity bf/b = 400 px
    n, uR, dep = oracle.stereo_matches(kL, dL, kR, dR, pyrL, pyrR, t["scale"], t["inv_scale"], bf, b)
    ok = uR >= 0
    assert n == ok.sum() and n > 100
    assert np.allclose(dep[ok], bf / (kL["x"][ok] - uR[ok]), rtol=1e-6)
    # the recovered disparity equals the planted per-row disparity for the bulk of the matches
    d_est = kL["x"][ok] - uR[ok]
    d_true = disp[np.clip(kL["y"][ok].astype(int), 0, 299)]
    assert np.mean(np.abs(d_est - d_true) < 1.5) > 0.8
    assert (uR[~ok] == -1).all() and (dep[~ok] == -1).all()


# ---- DBoW2 vocabulary transform -------------------------------------------------------------------

def _py_voc_transform(voc, desc, levelsup):
    """TemplatedVocabulary::transform (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1259) on Python dicts:
    dict insertion/+= for addWeight / addIfNotExist, sorted() for std::map order, float = IEEE double."""
    import math
    children, words = {}, {}
    for i, p in enumerate(voc["parent"]):
        children.setdefault(int(p), []).append(i + 1)
    for i, l in enumerate(voc["is_leaf"]):
        if l:
            words[i + 1] = len(words)
    L, weighting, scoring = voc["L"], voc["weighting"], voc["scoring"]
    tf = weighting in (0, 1)
    must, l2 = scoring != 5, scoring == 1
    bow, fv, wids, nids = {}, {}, [], []
    for f, d in enumerate(desc):
        node, level = 0, 0
        nid = 0 if L - levelsup <= 0 else None
        while True:
            level += 1
            ch = children[node]
            dist = [int(np.unpackbits(d ^ voc["desc"][c - 1]).sum()) for c in ch]
            node = ch[dist.index(min(dist))]  # first minimum: `d < best_d` is strict
            if level == L - levelsup:
                nid = node
            if node not in children:
                break
        if nid is None:
            nid = node
        w = float(voc["weight"][node - 1])
        wids.append(words[node]); nids.append(nid)
        if w > 0:
            if words[node] not in bow:
                bow[words[node]] = w
            elif tf:
                bow[words[node]] += w
            fv.setdefault(nid, []).append(f)
    ids = sorted(bow)
    vals = [bow[i] for i in ids]
    if tf and ids and not must:
        vals = [v / float(len(ids)) for v in vals]
    if must:
        norm = 0.0
        for v in vals:
            norm += v * v if l2 else abs(v)
        if l2:
            norm = math.sqrt(norm)
        if norm > 0.0:
            vals = [v / norm for v in vals]
    return wids, nids, ids, vals, fv


def _voc_features(voc, n, seed, flip=0.02):
    """Features near the words (so that words repeat), a few exact copies of inner nodes and a few random ones."""
    rng = np.random.default_rng(seed)
    leaves = voc["desc"][voc["is_leaf"] == 1]
    pick = rng.integers(0, max(len(leaves) // 3, 1), n)  # a third of the words: several features per word
    d = leaves[pick].copy()
    bits = np.unpackbits(d, axis=1)
    bits ^= (rng.random(bits.shape) < flip).astype(np.uint8)
    d = np.packbits(bits, axis=1)
    d[::17] = voc["desc"][rng.integers(0, len(voc["desc"]), len(d[::17]))]
    d[::29] = rng.integers(0, 256, (len(d[::29]), 32), dtype=np.uint8)
    return d


@pytest.mark.parametrize("k,L,ragged,weighting,scoring,levelsup,n", [
    (10, 3, True, 0, 0, 1, 300), (10, 3, False, 0, 0, 2, 300), (4, 4, True, 1, 1, 2, 200), (6, 3, True, 2, 0, 4, 200),
    (5, 3, True, 3, 5, 1, 150), (9, 2, True, 0, 5, 1, 150), (3, 5, True, 0, 2, 0, 120)])
def test_voc_transform_against_python_restatement(oracle, k, L, ragged, weighting, scoring, levelsup, n):
    voc = synth.vocabulary(k, L, seed=k * 10 + L, ragged=ragged, scoring=scoring, weighting=weighting)
    desc = _voc_features(voc, n, seed=n + k)
    wid, nid, (bi, bv), (fn, fs, ff) = oracle.voc_transform(voc, desc, levelsup)
    pw, pn, pids, pvals, pfv = _py_voc_transform(voc, desc, levelsup)
    assert list(wid) == pw and list(nid) == pn
    assert list(bi) == pids
    assert np.array_equal(bv.view(np.uint64), np.asarray(pvals, np.float64).view(np.uint64))  # bit-identical doubles
    assert list(fn) == sorted(pfv)
    for j, node in enumerate(fn):
        assert list(ff[fs[j]:fs[j + 1]]) == pfv[int(node)]
    kept = sum(len(v) for v in pfv.values())
    assert fs[-1] == kept and 0 < kept <= n
    if weighting in (0, 1):
        assert max(len(v) for v in pfv.values()) >= 3  # several features per node / word
    if scoring == 0:
        assert abs(bv.sum() - 1.0) < 1e-12


def test_voc_transform_edge_cases(oracle):
    voc = synth.vocabulary(10, 3, seed=5)
    # no features
    wid, nid, (bi, bv), (fn, fs, ff) = oracle.voc_transform(voc, np.zeros((0, 32), np.uint8), 1)
    assert len(wid) == 0 and len(bi) == 0 and len(fn) == 0 and list(fs) == [0]
    # every word stopped: both containers stay empty
    stopped = dict(voc); stopped["weight"] = np.zeros_like(voc["weight"])
    d = _voc_features(voc, 50, 1)
    wid, nid, (bi, bv), (fn, fs, ff) = oracle.voc_transform(stopped, d, 1)
    assert len(wid) == 50 and len(bi) == 0 and len(fn) == 0
    # levelsup >= L: every feature hangs off the root (node 0)
    wid, nid, (bi, bv), (fn, fs, ff) = oracle.voc_transform(voc, d, 3)
    assert (nid == 0).all() and list(fn) == [0]
    # an exact copy of a leaf descriptor reaches a word at distance 0 from it
    leaf = int(np.flatnonzero(voc["is_leaf"])[7])
    wid, nid, _, _ = oracle.voc_transform(voc, voc["desc"][leaf][None], 0)
    words = np.cumsum(voc["is_leaf"]) - 1
    reached = int(np.flatnonzero((words == wid[0]) & (voc["is_leaf"] == 1))[0])
    assert nid[0] == reached + 1  # levelsup 0 at full depth: the node is the leaf itself, or a shallower leaf
    # inconsistent trees are refused
    bad = dict(voc); bad["parent"] = voc["parent"].copy(); bad["parent"][3] = 50
    with pytest.raises(ValueError):
        oracle.voc_transform(bad, d, 1)
    bad = dict(voc); bad["is_leaf"] = voc["is_leaf"].copy(); bad["is_leaf"][0] ^= 1
    with pytest.raises(ValueError):
        oracle.voc_transform(bad, d, 1)
