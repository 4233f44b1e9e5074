#!/usr/bin/env python3
"""Randomized parity sweep of the device path against the CPU oracle (run on the GPU box):
random image sizes, extractor settings and image kinds through the extractor, then random
windowed matching of two extractions.  Prints one line per mismatch with the seed that
reproduces it, and a summary.  usage: python tools/fuzz_parity.py [--seconds 240] [--seed 0]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from orb_slam2_amd import api, synth  # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))  # hiputil
from oracle import oracle as O  # noqa: E402  (the checker)

KINDS = ["scene", "noise", "lownoise", "blocks", "checker", "dots", "saturated", "smooth"]


def make_image(kind, w, h, rng):
    if kind == "scene":
        return synth.sequence(w, h, 1, seed=int(rng.integers(1 << 30)))[0]
    if kind == "noise":
        return rng.integers(0, 256, (h, w), dtype=np.uint8)
    if kind == "lownoise":
        amp = int(rng.integers(4, 40))
        return (128 + rng.integers(-amp, amp + 1, (h, w))).astype(np.uint8)
    if kind == "blocks":
        img = np.full((h, w), int(rng.integers(256)), np.uint8)
        for _ in range(int(rng.integers(20, 400))):
            x, y = int(rng.integers(w)), int(rng.integers(h))
            img[y:y + int(rng.integers(1, 60)), x:x + int(rng.integers(1, 60))] = int(rng.integers(256))
        return img
    if kind == "checker":
        p = int(rng.integers(2, 40))
        a, b = int(rng.integers(256)), int(rng.integers(256))
        yy, xx = np.mgrid[0:h, 0:w]
        return np.where(((yy // p) + (xx // p)) & 1, a, b).astype(np.uint8)
    if kind == "dots":
        img = np.full((h, w), int(rng.integers(60, 200)), np.uint8)
        n = int(rng.integers(1, 3000))
        img[rng.integers(0, h, n), rng.integers(0, w, n)] = rng.integers(0, 256, n)
        return img
    if kind == "saturated":
        img = np.where(rng.random((h // 8 + 1, w // 8 + 1)) < 0.5, 0, 255).astype(np.uint8)
        return np.kron(img, np.ones((8, 8), np.uint8))[:h, :w].copy()
    yy, xx = np.mgrid[0:h, 0:w]
    img = 127 + 100 * np.sin(xx / float(rng.uniform(3, 40))) * np.cos(yy / float(rng.uniform(3, 40)))
    return (img + rng.integers(-3, 4, (h, w))).clip(0, 255).astype(np.uint8)


def compare_extract(ex, oe, img, stats):
    kps, desc = ex(img)
    okps, odesc = oe.extract(img)
    if len(kps) != len(okps):
        return "keypoint count %d vs oracle %d" % (len(kps), len(okps)), None
    for f in ("x", "y", "size", "response", "octave", "class_id"):
        if not np.array_equal(kps[f], okps[f]):
            return "field %s differs at %d keypoints" % (f, int((kps[f] != okps[f]).sum())), None
    if len(kps):
        dang = np.abs(kps["angle"] - okps["angle"])
        dang = np.minimum(dang, 360 - dang)
        stats["angle_max_deg"] = max(stats["angle_max_deg"], float(dang.max()))
        stats["angle_inexact"] += int((dang != 0).sum())
        if dang.max() > np.degrees(1e-4):
            return "angle differs by %g deg" % dang.max(), None
    nbad = int((desc != odesc).any(axis=1).sum())
    stats["desc_diff"] += nbad
    stats["desc_total"] += len(desc)
    stats["kp_total"] += len(kps)
    # the reference breaks size ties between quadtree nodes by heap address (src/ORBextractor.cc:906): count the levels
    # on which such a tie straddles the cut (:959) and, on a sample, the keypoints that depend on the order
    ties = oe.tie_straddles()
    stats["frames"] += 1
    stats["tie_levels"] += ties
    stats["tie_frames"] += 1 if ties else 0
    if stats["frames"] % 4 == 0:  # sampled whether or not a tie straddles the cut: the order alone can change too
        O.set_tie_rule(1)
        try:
            akps, _ = oe.extract(img)
        finally:
            O.set_tie_rule(0)
        mine = set(zip(okps["x"].tolist(), okps["y"].tolist(), okps["octave"].tolist()))
        other = set(zip(akps["x"].tolist(), akps["y"].tolist(), akps["octave"].tolist()))
        stats["tie_sampled_kp"] += len(mine)
        stats["tie_dependent_kp"] += len(mine - other)
        # per level: does the other tie order select another set, or the same set in another output order (the
        # windowed matchers resolve their claims in keypoint order, so the order is an output too)
        for lv in np.union1d(okps["octave"], akps["octave"]):
            a = [(x, y) for x, y in zip(okps["x"][okps["octave"] == lv].tolist(), okps["y"][okps["octave"] == lv].tolist())]
            b = [(x, y) for x, y in zip(akps["x"][akps["octave"] == lv].tolist(), akps["y"][akps["octave"] == lv].tolist())]
            stats["tie_levels_sampled"] += 1
            if set(a) != set(b):
                stats["tie_levels_set_differs"] += 1
            elif a != b:
                stats["tie_levels_order_only"] += 1
    return None, (okps, odesc)


def compare_match(m, a, b, w, h, scale, rng):
    (k1, d1), (k2, d2) = a, b
    if len(k1) == 0 or len(k2) == 0:
        return None
    bounds = np.array([0, 0, w, h], np.float32)
    mode = int(rng.integers(0, 3))
    th = float(rng.uniform(2, 60))
    q = api.shift_queries(k1, float(rng.uniform(-4, 4)), float(rng.uniform(-4, 4)), th, scale, bounds)
    q["flags"] = np.where(rng.random(len(q)) < 0.1, q["flags"] & 1, q["flags"])
    if mode == 0:
        q["minLevel"] = k1["octave"] - 1
        q["maxLevel"] = k1["octave"]
    occ = (rng.random(len(k2)) < 0.1).astype(np.uint8)
    q = q.astype(api.WQUERY_DTYPE)
    ratio = float(rng.uniform(0.5, 1.0))
    th_dist = int(rng.integers(20, 257))
    ori = bool(rng.integers(2))
    m.mfNNratio = ratio
    m.mbCheckOrientation = ori
    n, assign = m.SearchByProjection(mode, q, d1, k2, d2, bounds, occupied=occ, th_dist=th_dist)
    rn, rassign = O.search_by_projection(mode, q.view(O.WQUERY_DTYPE), d1, k2.view(O.KEYPOINT_DTYPE), d2, bounds, occupied=occ,
                                         th_dist=th_dist, nnratio=ratio, checkOri=ori)
    if n != rn or not np.array_equal(assign, rassign):
        return "SearchByProjection mode %d th %.2f ratio %.3f th_dist %d ori %d: %d vs %d matches, %d assignments differ" % (
            mode, th, ratio, th_dist, ori, n, rn, int((assign != rassign).sum()))
    win = int(rng.integers(5, 300))
    mi = api.ORBmatcher(ratio, ori)
    prev = np.stack([k1["x"], k1["y"]], 1).astype(np.float32) + rng.uniform(-3, 3, (len(k1), 2)).astype(np.float32)
    got = mi.SearchForInitialization(k1, d1, k2, d2, bounds, prev, win)
    ref = O.search_for_initialization(k1.view(O.KEYPOINT_DTYPE), d1, k2.view(O.KEYPOINT_DTYPE), d2, bounds, prev, win, ratio, ori)
    if got[0] != ref[0] or not np.array_equal(got[1], ref[1]) or not np.array_equal(got[2], ref[2]):
        return "SearchForInitialization window %d ratio %.3f ori %d: %d vs %d" % (win, ratio, ori, got[0], ref[0])
    return None


def _featvec(desc, nbits):
    """Stand-in FeatureVector: node id = leading descriptor bits."""
    node = desc[:, 0].astype(np.int32) >> (8 - nbits)
    return {int(n): np.nonzero(node == n)[0].tolist() for n in np.unique(node)}


def compare_match2(a, b, w, h, scale, rng):
    """SearchByBoW, SearchForTriangulation and the area search of Fuse / SearchBySim3 with random settings."""
    (k1, d1), (k2, d2) = a, b
    if len(k1) == 0 or len(k2) == 0:
        return None
    bounds = np.array([0, 0, w, h], np.float32)
    ratio, ori = float(rng.uniform(0.5, 1.0)), bool(rng.integers(2))
    m = api.ORBmatcher(ratio, ori)
    nbits = int(rng.integers(0, 7))
    fv1, fv2 = _featvec(d1, nbits), _featvec(d2, nbits)
    if len(fv1) > 3 and len(fv2) > 3:
        fv1.pop(sorted(fv1)[1]); fv2.pop(sorted(fv2)[-2])
    strict = bool(rng.integers(2))
    valid1 = (rng.random(len(k1)) < 0.8).astype(np.uint8)
    excl2 = (rng.random(len(k2)) < 0.1).astype(np.uint8) if strict else None
    n, a1, m12 = m.SearchByBoW(fv1, k1, d1, valid1, fv2, k2, d2, excl2, strict)
    qi, qs, ss, si = api.bow_segments(fv1, fv2, valid1)
    q = np.zeros(len(qi), O.WQUERY_DTYPE)
    q["angle"] = k1["angle"][qi]; q["minLevel"] = qs; q["flags"] = 1
    rn, rassign, rqm = O.search_by_bow(q, d1[qi], k2.view(O.KEYPOINT_DTYPE), d2, ss, si, excl2, 49 if strict else 50, ratio, ori)
    ra1 = np.where(rassign >= 0, qi[np.clip(rassign, 0, None)], -1) if len(qi) else np.full(len(k2), -1, np.int32)
    rm12 = np.full(len(k1), -1, np.int32); rm12[qi[rqm >= 0]] = rqm[rqm >= 0]
    if n != rn or not np.array_equal(a1, ra1) or not np.array_equal(m12, rm12):
        return "SearchByBoW nbits %d ratio %.3f ori %d strict %d: %d vs %d" % (nbits, ratio, ori, strict, n, rn)

    stereo, only_stereo = bool(rng.integers(2)), bool(rng.integers(2))
    has1 = rng.random(len(k1)) < 0.2; has2 = rng.random(len(k2)) < 0.2
    ur1 = np.where(rng.random(len(k1)) < 0.6, k1["x"] - 7.0, -1.0).astype(np.float32) if stereo else None
    ur2 = np.where(rng.random(len(k2)) < 0.6, k2["x"] - 7.0, -1.0).astype(np.float32) if stereo else None
    ang = float(rng.uniform(0, 2 * np.pi))
    t = np.array([np.cos(ang), np.sin(ang)], np.float32)
    F12 = np.array([[0, 0, t[1]], [0, 0, -t[0]], [-t[1], t[0], 0]], np.float32)
    epipole = (float(rng.uniform(0, w)), float(rng.uniform(0, h)))
    scale2 = np.asarray(scale, np.float32); sigma2 = (scale2 * scale2).astype(np.float32)
    n, m12 = m.SearchForTriangulation(fv1, k1, d1, has1, ur1, fv2, k2, d2, has2, ur2, F12, epipole, scale2, sigma2, only_stereo)
    stereo1 = np.zeros(len(k1), bool) if ur1 is None else ur1 >= 0
    use1 = ~has1 & (stereo1 if only_stereo else True)
    qi, qs, ss, si = api.bow_segments(fv1, fv2, use1)
    q = np.zeros(len(qi), O.WQUERY_DTYPE)
    q["u"] = k1["x"][qi]; q["v"] = k1["y"][qi]; q["angle"] = k1["angle"][qi]; q["minLevel"] = qs; q["flags"] = 1 + 4 * stereo1[qi]
    rn, rqm = O.search_for_triangulation(q, d1[qi], k2.view(O.KEYPOINT_DTYPE), d2, ur2, ss, si, has2.astype(np.uint8), F12,
                                         epipole[0], epipole[1], scale2, sigma2, only_stereo, ori)
    rm12 = np.full(len(k1), -1, np.int32); rm12[qi[rqm >= 0]] = rqm[rqm >= 0]
    if n != rn or not np.array_equal(m12, rm12):
        return "SearchForTriangulation nbits %d stereo %d/%d ori %d: %d vs %d" % (nbits, stereo, only_stereo, ori, n, rn)

    th, chi2, th_dist = float(rng.uniform(1, 40)), bool(rng.integers(2)), int(rng.integers(20, 257))
    q = api.shift_queries(k1, float(rng.uniform(-4, 4)), float(rng.uniform(-4, 4)), th, scale2, bounds)
    q["minLevel"] = k1["octave"] - 1; q["maxLevel"] = k1["octave"]
    q["u"] += rng.normal(0, 1.0, len(q)).astype(np.float32); q["v"] += rng.normal(0, 1.0, len(q)).astype(np.float32)
    q["ur"] = q["u"] - 6.0
    q["flags"] = np.where(rng.random(len(q)) < 0.1, 0, 3)
    q = q.astype(api.WQUERY_DTYPE)
    inv_s2 = (1.0 / sigma2).astype(np.float32)
    n, qm = m.SearchInArea(q, d1, k2, d2, bounds, ur2, inv_s2, chi2, th_dist)
    rn, rqm = O.search_in_area(q.view(O.WQUERY_DTYPE), d1, k2.view(O.KEYPOINT_DTYPE), d2, bounds, ur2, inv_s2, chi2, th_dist)
    if n != rn or not np.array_equal(qm, rqm):
        return "SearchInArea th %.2f chi2 %d th_dist %d: %d vs %d" % (th, chi2, th_dist, n, rn)
    return None


def compare_bow(a, rng):
    """The vocabulary transform of one view's descriptors against a random vocabulary."""
    d = a[1]
    k, L = int(rng.integers(2, 21)), int(rng.integers(1, 5))
    if k ** L > 20000:
        L = 2
    weighting, scoring = int(rng.integers(4)), int(rng.integers(6))
    voc = synth.vocabulary(k, L, seed=int(rng.integers(1 << 30)), ragged=bool(rng.integers(2)), scoring=scoring, weighting=weighting)
    # half of the vocabulary's words are descriptors of the view, so that the descent is not decided at the root
    leaves = np.nonzero(voc["is_leaf"] == 1)[0]
    if len(d):
        take = leaves[rng.random(len(leaves)) < 0.5]
        voc["desc"][take] = d[rng.integers(0, len(d), len(take))]
    v = api.ORBVocabulary.from_arrays(voc["k"], voc["L"], voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"],
                                      scoring=scoring, weighting=weighting)
    levelsup = int(rng.integers(0, L + 2))
    got = v.transform_arrays(d, levelsup)
    ref = O.voc_transform(voc, d, levelsup)
    ok = (np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]) and np.array_equal(got[2][0], ref[2][0])
          and np.array_equal(got[2][1].view(np.uint64), ref[2][1].view(np.uint64))
          and all(np.array_equal(x, y) for x, y in zip(got[3], ref[3])))
    if not ok:
        return "vocabulary transform k %d L %d weighting %d scoring %d levelsup %d differs" % (k, L, weighting, scoring, levelsup)
    return None


def compare_stereo(m, exL, exR, a, b, nlevels, rng):
    """Frame::ComputeStereoMatches on the two views as a rectified pair (the extractors still hold their pyramids)."""
    (kL, dL), (kR, dR) = a, b
    if len(kL) == 0 or len(kR) == 0:
        return None
    fx = float(rng.uniform(200, 900))
    bf = fx * float(rng.uniform(0.05, 0.6))
    pyrL = [exL.pyramid_level(l) for l in range(nlevels)]
    pyrR = [exR.pyramid_level(l) for l in range(nlevels)]
    n, uR, dep = m.ComputeStereoMatches(exL, exR, kL.view(api.KEYPOINT_DTYPE), dL, kR.view(api.KEYPOINT_DTYPE), dR, bf, bf / fx)
    sc = np.asarray(exL.GetScaleFactors(), np.float32)
    isc = np.asarray(exL.GetInverseScaleFactors(), np.float32)
    rn, ruR, rdep = O.stereo_matches(kL.view(O.KEYPOINT_DTYPE), dL, kR.view(O.KEYPOINT_DTYPE), dR, pyrL, pyrR, sc, isc, bf, bf / fx)
    if n != rn or not np.array_equal(uR, ruR) or not np.array_equal(dep, rdep):
        return "ComputeStereoMatches bf %.3f fx %.3f: %d vs %d matched, %d uR differ" % (bf, fx, n, rn, int((uR != ruR).sum()))
    return None


def compare_frame(m, ex, ex2, img, img2, a, b, w, h, nlevels, rng):
    """The device-resident frame (orbgpu_frame_build_*) against the oracle: contents, grid, and one search with the
    frame as the train side."""
    (kA, dA), (kB, dB) = a, b
    fx, fy = float(rng.uniform(200, 900)), float(rng.uniform(200, 900))
    K4 = [fx, fy, w / 2 + float(rng.uniform(-20, 20)), h / 2 + float(rng.uniform(-20, 20))]
    D5 = [0, 0, 0, 0, 0] if rng.random() < 0.4 else [float(rng.uniform(-0.3, 0.3)), float(rng.uniform(-0.2, 0.2)),
                                                     float(rng.uniform(-0.01, 0.01)), float(rng.uniform(-0.01, 0.01)),
                                                     float(rng.uniform(-0.1, 0.1))]
    if D5[0] == 0:
        bounds = np.array([0, 0, w, h], np.float32)
    else:
        c = np.zeros(4, O.KEYPOINT_DTYPE)
        c["x"] = [0, w, 0, w]; c["y"] = [0, 0, h, h]
        u = O.undistort_keypoints(c, K4, D5)
        bounds = np.array([min(u["x"][0], u["x"][2]), min(u["y"][0], u["y"][1]), max(u["x"][1], u["x"][3]), max(u["y"][2], u["y"][3])],
                          np.float32)
        if not (bounds[2] > bounds[0] + 8 and bounds[3] > bounds[1] + 8):
            return None  # a distortion that folds the image: no frame to speak of
    kind = int(rng.integers(3))
    f = api.Frame(ex.max_keypoints)
    bf = fx * float(rng.uniform(0.05, 0.6))
    if kind == 0:
        f.build_mono(ex, img, K4, D5, bounds)
        rur = np.full(len(kA), -1, np.float32); rdp = rur.copy(); rnv = 0
    elif kind == 1:
        depth = rng.uniform(0.3, 8.0, (h, w)).astype(np.float32)
        depth[rng.random(depth.shape) < 0.2] = 0.0
        f.build_rgbd(ex, img, depth, bf, K4, D5, bounds)
    else:
        _, nv = f.build_stereo(ex, ex2, img, img2, bf, bf / fx, K4, D5, bounds)
        pyrL = [ex.pyramid_level(l) for l in range(nlevels)]
        pyrR = [ex2.pyramid_level(l) for l in range(nlevels)]
        sc = np.asarray(ex.GetScaleFactors(), np.float32); isc = np.asarray(ex.GetInverseScaleFactors(), np.float32)
        rnv, rur, rdp = O.stereo_matches(kA, dA, kB, dB, pyrL, pyrR, sc, isc, bf, bf / fx)
    what = ("mono", "rgbd", "stereo")[kind]
    if f.N != len(kA):
        return "frame (%s): %d keypoints, the extraction %d" % (what, f.N, len(kA))
    for fld in ("x", "y", "size", "response", "octave", "class_id"):
        if not np.array_equal(f.mvKeys[fld], kA[fld]):
            return "frame (%s): keypoint field %s differs from the extraction" % (what, fld)
    # angles and descriptors: the bar of compare_extract (the oracle's float sin / cos may round differently)
    if f.N and (np.abs(f.mvKeys["angle"] - kA["angle"]).max() > np.degrees(1e-4) or (f.mDescriptors != dA).any(axis=1).sum() > max(1, f.N // 1000)):
        return "frame (%s): angles / descriptors differ from the extraction" % what
    ru = O.undistort_keypoints(kA, K4, D5) if D5[0] != 0 else kA
    un = f.mvKeysUn
    if len(kA) and (np.abs(un["x"] - ru["x"]).max() > 1e-3 or np.abs(un["y"] - ru["y"]).max() > 1e-3 or not np.array_equal(un["octave"], kA["octave"])):
        return "frame (%s): undistorted keypoints differ by %g px" % (what, float(np.abs(un["x"] - ru["x"]).max()))
    if kind == 1:
        rur, rdp = O.stereo_from_rgbd(kA, un.view(O.KEYPOINT_DTYPE), depth, bf)
        rnv = int((rdp > 0).sum())
    if not np.array_equal(f.mvuRight, rur) or not np.array_equal(f.mvDepth, rdp) or f.nvalid != rnv:
        return "frame (%s): uRight / depth differ at %d keypoints (valid %d vs %d)" % (what, int((f.mvuRight != rur).sum()), f.nvalid, rnv)
    cs, ci = f.grid()
    rcs, rci = O.grid_build(un.view(O.KEYPOINT_DTYPE), bounds)
    if not np.array_equal(cs, rcs) or not np.array_equal(ci, rci):
        return "frame (%s): grid differs" % what
    if len(kB) == 0 or f.N == 0:
        return None
    mode = int(rng.integers(0, 3))
    th = float(rng.uniform(2, 40))
    q = api.shift_queries(kB, float(rng.uniform(-4, 4)), float(rng.uniform(-4, 4)), th, np.asarray(ex.GetScaleFactors(), np.float32), bounds)
    if mode == 0:
        q["minLevel"] = kB["octave"] - 1
        q["maxLevel"] = kB["octave"]
    q["ur"] = q["u"] - float(rng.uniform(0, 30))
    q = q.astype(api.WQUERY_DTYPE)
    occ = (rng.random(f.N) < 0.1).astype(np.uint8)
    n, assign = m.SearchByProjectionFrame(mode, q, dB, f, occupied=occ, th_dist=100)
    rn, rassign = O.search_by_projection(mode, q.view(O.WQUERY_DTYPE), dB, un.view(O.KEYPOINT_DTYPE), dA, bounds, uright2=f.mvuRight,
                                         occupied=occ, th_dist=100, nnratio=m.mfNNratio, checkOri=m.mbCheckOrientation)
    if n != rn or not np.array_equal(assign, rassign):
        return "frame (%s): SearchByProjection mode %d against the resident frame: %d vs %d" % (what, mode, n, rn)
    return None


def compare_batch(m, img, img2, a, b, w, h, nfeat, scale, nlevels, ini, mn, gauss, rng):
    """The batch entry points on device-resident data: orbgpu_extract_batch_device of a few views, then
    orbgpu_stereo_match_batch_device (left = views, right = the same views in another order),
    orbgpu_search_for_initialization_batch_device between them, and the brute-force scan on both kernels."""
    from hiputil import Hip
    B = int(rng.integers(2, 6))
    views = [img, img2] + [(np.roll(img, int(rng.integers(-3, 4)), 1).astype(np.int16) + rng.integers(-2, 3, img.shape)).clip(0, 255).astype(np.uint8)
                           for _ in range(B - 2)]
    left = np.stack(views)
    right = np.stack([views[(i + 1) % B] for i in range(B)])
    exL = api.ORBextractor(nfeat, scale, nlevels, ini, mn, max_width=w, max_height=h, gauss_variant=gauss, max_batch=B)
    exR = api.ORBextractor(nfeat, scale, nlevels, ini, mn, max_width=w, max_height=h, gauss_variant=gauss, max_batch=B)
    cap = exL.max_keypoints
    fx = float(rng.uniform(200, 900))
    bf = fx * float(rng.uniform(0.05, 0.6))
    win = int(rng.integers(5, 300))
    ratio, ori = float(rng.uniform(0.5, 1.0)), bool(rng.integers(2))
    mi = api.ORBmatcher(ratio, ori)
    bounds = np.array([0, 0, w, h], np.float32)
    hip = Hip()
    try:
        dimg = [hip.upload(left), hip.upload(right)]
        kk = [hip.alloc(B * cap * 28) for _ in range(2)]; dd = [hip.alloc(B * cap * 32) for _ in range(2)]; cc = [hip.alloc(B * 4) for _ in range(2)]
        for e, i in ((exL, 0), (exR, 1)):
            e.extract_batch_device(dimg[i].value, B, w, h, w, w * h, kk[i].value, dd[i].value, cap, cc[i].value)
            e.sync()
        uR, dep, nv = hip.alloc(B * cap * 4), hip.alloc(B * cap * 4), hip.alloc(B * 4)
        m12, nm = hip.alloc(B * cap * 4), hip.alloc(B * 4)
        m.stereo_match_batch_device(exL, exR, B, kk[0].value, dd[0].value, cc[0].value, cap, kk[1].value, dd[1].value, cc[1].value, cap,
                                    bf, bf / fx, uR.value, dep.value, nv.value)
        m.sync()
        mi.search_for_initialization_batch_device(B, kk[0].value, dd[0].value, cc[0].value, cap, kk[1].value, dd[1].value, cc[1].value, cap,
                                                  bounds, m12.value, nm.value, windowSize=win)
        mi.sync()
        N = [hip.download(cc[i], B, np.int32) for i in range(2)]
        K = [hip.download(kk[i], (B, cap), api.KEYPOINT_DTYPE) for i in range(2)]
        D = [hip.download(dd[i], (B, cap, 32), np.uint8) for i in range(2)]
        UR, DEP, NV = hip.download(uR, (B, cap), np.float32), hip.download(dep, (B, cap), np.float32), hip.download(nv, B, np.int32)
        M12, NM = hip.download(m12, (B, cap), np.int32), hip.download(nm, B, np.int32)
    finally:
        hip.free()
    sc = np.asarray(exL.GetScaleFactors(), np.float32); isc = np.asarray(exL.GetInverseScaleFactors(), np.float32)
    for p in range(B):
        kL, kR, dL, dR = K[0][p, :N[0][p]], K[1][p, :N[1][p]], D[0][p, :N[0][p]], D[1][p, :N[1][p]]
        if p == 0 and (len(kL) != len(a[0]) or not np.array_equal(kL["x"], a[0]["x"]) or not np.array_equal(dL, a[1])):
            return "batch of %d: frame 0 differs from its single-frame extraction" % B
        pyrL = [exL.pyramid_level(l, frame=p) for l in range(nlevels)]; pyrR = [exR.pyramid_level(l, frame=p) for l in range(nlevels)]
        rn, ruR, rdep = O.stereo_matches(kL.view(O.KEYPOINT_DTYPE), dL, kR.view(O.KEYPOINT_DTYPE), dR, pyrL, pyrR, sc, isc, bf, bf / fx)
        if NV[p] != rn or not np.array_equal(UR[p, :len(kL)], ruR) or not np.array_equal(DEP[p, :len(kL)], rdep):
            return "stereo batch of %d, pair %d, bf %.3f fx %.3f: %d vs %d with a depth" % (B, p, bf, fx, NV[p], rn)
        prev = np.stack([kL["x"], kL["y"]], 1).astype(np.float32).reshape(-1)
        rni, rm12, _ = O.search_for_initialization(kL.view(O.KEYPOINT_DTYPE), dL, kR.view(O.KEYPOINT_DTYPE), dR, bounds, prev, win, ratio, ori)
        if NM[p] != rni or not np.array_equal(M12[p, :len(kL)], rm12):
            return "SearchForInitialization batch of %d, pair %d, window %d ratio %.3f ori %d: %d vs %d" % (B, p, win, ratio, ori, NM[p], rni)
    # brute force: frame 0's descriptors against frame 1's (with a few duplicated rows: ties), both kernels
    Q, T = D[0][0, :N[0][0]], D[1][0, :N[1][0]]
    if len(Q) and len(T):
        T = T.copy(); T[rng.integers(0, len(T), 3)] = T[0]
        ref = O.bruteforce(Q, T)
        for which in (1, 2):
            m.set_bruteforce_kernel(which)
            try:
                got = m.BruteForce(Q, T)
            finally:
                m.set_bruteforce_kernel(0)
            if not all(np.array_equal(g, r) for g, r in zip(got, ref)):
                return "brute force %d x %d on kernel %d differs" % (len(Q), len(T), which)
    return None


def main(argv=None, stats_out=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--case", type=int, default=-1, help="run only this case number")
    ap.add_argument("--cases", type=int, default=1 << 30, help="stop after this many cases")
    args = ap.parse_args(argv)
    O.build()
    stats = dict(angle_max_deg=0.0, angle_inexact=0, desc_diff=0, desc_total=0, kp_total=0, frames=0, tie_levels=0, tie_frames=0,
                 tie_sampled_kp=0, tie_dependent_kp=0, tie_levels_sampled=0, tie_levels_set_differs=0, tie_levels_order_only=0)
    t_end = time.time() + args.seconds
    case, nfail, nskip, nmatch = 0, 0, 0, 0
    m = api.ORBmatcher(0.9, True)
    while time.time() < t_end and case < args.cases:
        if args.case >= 0:
            case = args.case
        rng = np.random.default_rng([args.seed, case])
        kind = KINDS[int(rng.integers(len(KINDS)))]
        scale = float(rng.choice([1.1, 1.2, 1.2, 1.2, 1.3, 1.5, 2.0]))
        nlevels = int(rng.integers(1, 11))
        w, h = int(rng.integers(70, 1300)), int(rng.integers(70, 1000))
        if rng.random() < 0.95:
            # portrait frames whose search area (side - 26 on every level) is more than twice as tall as wide get no
            # quadtree root, in the reference (which then indexes an empty vector) as here: draw few of them
            h = min(h, int(1.45 * w))
        # the smallest level must keep a search area: sides of at least 62 px
        while nlevels > 1 and min(w, h) / scale ** (nlevels - 1) < 64:
            nlevels -= 1
        nfeat = int(rng.choice([1, 7, 50, 300, 500, 1000, 1000, 2000, 4000, 8000]))
        ini = int(rng.integers(5, 61))
        mn = int(rng.integers(1, ini + 1))
        gauss = int(rng.integers(2))
        tag = "case %d (seed %d): %s %dx%d nfeat %d scale %.1f levels %d fast %d/%d gauss %d" % (
            case, args.seed, kind, w, h, nfeat, scale, nlevels, ini, mn, gauss)
        try:
            ex = api.ORBextractor(nfeat, scale, nlevels, ini, mn, max_width=w, max_height=h, gauss_variant=gauss)
        except api.OrbGpuError as e:
            nskip += 1
            print("SKIP %s: %s" % (tag, e), flush=True)
            case += 1
            if args.case >= 0:
                break
            continue
        ex2 = api.ORBextractor(nfeat, scale, nlevels, ini, mn, max_width=w, max_height=h, gauss_variant=gauss)
        oe = O.Extractor(nfeat, scale, nlevels, ini, mn, gauss_variant=gauss)
        img = make_image(kind, w, h, rng)
        err, a = compare_extract(ex, oe, img, stats)
        if err is None:
            l = int(rng.integers(nlevels))
            if not np.array_equal(ex.pyramid_level(l, border=19), O.border101(oe.level(l), 19)):
                err = "pyramid level %d (with its border) differs" % l
            rb = oe.level(l, blurred=True)
            if err is None and rb is not None and not np.array_equal(ex.pyramid_level(l, blurred=True), rb):
                err = "blurred level %d differs" % l
        if err is None:
            # a second view: the same image shifted by a few pixels with a little noise
            dx, dy = int(rng.integers(-3, 4)), int(rng.integers(-3, 4))
            img2 = np.roll(img, (dy, dx), (0, 1))
            img2 = (img2.astype(np.int16) + rng.integers(-2, 3, img2.shape)).clip(0, 255).astype(np.uint8)
            err, b = compare_extract(ex2, oe, img2, stats)
            if err is None:
                err = compare_stereo(m, ex, ex2, a, b, nlevels, rng) if rng.random() < 0.5 else None
            if err is None:
                err = compare_match(m, a, b, w, h, np.asarray(ex.GetScaleFactors(), np.float32), rng)
            if err is None and rng.random() < 0.5:
                err = compare_match2(a, b, w, h, np.asarray(ex.GetScaleFactors(), np.float32), rng)
            if err is None and rng.random() < 0.3 and len(a[0]) <= 16384:
                err = compare_bow(a, rng)
            if err is None and rng.random() < 0.5:
                err = compare_frame(m, ex, ex2, img, img2, a, b, w, h, nlevels, rng)
            if err is None and rng.random() < 0.3 and w * h <= 800 * 800:
                err = compare_batch(m, img, img2, a, b, w, h, nfeat, scale, nlevels, ini, mn, gauss, rng)
            nmatch += 1
        if err is not None:
            nfail += 1
            print("FAIL %s: %s" % (tag, err), flush=True)
        elif case % 20 == 0:
            print("ok   %s (%d keypoints)" % (tag, len(a[0])), flush=True)
        case += 1
        if args.case >= 0:
            break
    print("cases %d, failed %d, skipped %d, matched pairs %d; keypoints %d, angles not bit-identical %d (max %.3g deg), "
          "descriptors differing %d of %d" % (case, nfail, nskip, nmatch, stats["kp_total"], stats["angle_inexact"],
                                             stats["angle_max_deg"], stats["desc_diff"], stats["desc_total"]), flush=True)
    print("quadtree size ties straddling the cut (the reference orders them by heap address): %d of %d frames, %d levels; "
          "keypoints that change with the tie order (earliest- instead of latest-created node first, sampled): %d of %d"
          % (stats["tie_frames"], stats["frames"], stats["tie_levels"], stats["tie_dependent_kp"], stats["tie_sampled_kp"]), flush=True)
    print("levels sampled with both tie orders: %d; another keypoint set on %d, the same set in another output order on %d"
          % (stats["tie_levels_sampled"], stats["tie_levels_set_differs"], stats["tie_levels_order_only"]), flush=True)
    if stats_out is not None:
        stats_out.update(stats)
    return 1 if nfail else 0


if __name__ == "__main__":
    sys.exit(main())
