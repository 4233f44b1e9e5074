"""Committed golden fixtures (tests/golden, made by tools/make_golden.py from the oracle).

CPU: the oracle still reproduces them.  GPU: the HIP library reproduces them
through the C ABI without the oracle in the loop.
"""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


def test_oracle_reproduces_extraction_goldens(oracle):
    g = _load("extract_320x240.npz")
    oe = oracle.Extractor(int(g["params"][0]), float(g["scale"]), int(g["params"][1]), int(g["params"][2]), int(g["params"][3]))
    for i in range(2):
        k, d = oe.extract(g["images"][i])
        assert np.array_equal(k, g["kps%d" % i]) and np.array_equal(d, g["desc%d" % i])
    g = _load("extract_noise_300x260.npz")
    on = oracle.Extractor(int(g["params"][0]))
    k, d = on.extract(g["image"])
    assert np.array_equal(k, g["kps"]) and np.array_equal(d, g["desc"])


def test_oracle_reproduces_match_goldens(oracle):
    e = _load("extract_320x240.npz"); g = _load("match_320x240.npz")
    b = np.array([0, 0, 320, 240], np.float32)
    for mode in (0, 1, 2):
        n, a = oracle.search_by_projection(mode, g["q%d" % mode], e["desc0"], e["kps1"], e["desc1"], b, th_dist=100,
                                           nnratio=0.8, checkOri=True)
        assert n == int(g["n%d" % mode]) and np.array_equal(a, g["assign%d" % mode])
    prev = np.stack([e["kps0"]["x"], e["kps0"]["y"]], axis=1).astype(np.float32)
    n, m12, po = oracle.search_for_initialization(e["kps0"], e["desc0"], e["kps1"], e["desc1"], b, prev, 100, 0.9, True)
    assert n == int(g["init_n"]) and np.array_equal(m12, g["init_m12"]) and np.array_equal(po, g["init_prev"])
    bi, b1, b2 = oracle.bruteforce(g["bf_Q"], g["bf_T"])
    assert np.array_equal(bi, g["bf_idx"]) and np.array_equal(b1, g["bf_d1"]) and np.array_equal(b2, g["bf_d2"])


def _match2(run_bow, run_tri, run_area, run_dd):
    e = _load("extract_320x240.npz"); g = _load("match2_320x240.npz")
    b = np.array([0, 0, 320, 240], np.float32)
    qi = g["bow_qi"]
    n, assign, qm = run_bow(g["bow_q"], e["desc0"][qi], e["kps1"], e["desc1"], g["bow_ss"], g["bow_si"])
    assert n == int(g["bow_n"]) and np.array_equal(assign, g["bow_assign"]) and np.array_equal(qm, g["bow_qm"])
    sc = np.float32(1.2) ** np.arange(8, dtype=np.float32)
    n, qm = run_tri(g["bow_q"], e["desc0"][qi], e["kps1"], e["desc1"], g["bow_ss"], g["bow_si"], g["tri_has1"], g["tri_F"])
    assert n == int(g["tri_n"]) and np.array_equal(qm, g["tri_qm"])
    n, qm = run_area(g["area_q"], e["desc0"], e["kps1"], e["desc1"], b, g["area_ur"])
    assert n == int(g["area_n"]) and np.array_equal(qm, g["area_qm"])
    bi, bm = run_dd(g["dd_obs"], g["dd_start"])
    assert np.array_equal(bi, g["dd_idx"]) and np.array_equal(bm, g["dd_med"])
    del sc


def _scale_tables(oracle):
    sc = oracle.tables(oracle.make_params(500))["scale"]
    return sc, (sc * sc).astype(np.float32), (1.0 / (sc * sc)).astype(np.float32)


def test_oracle_reproduces_match2_goldens(oracle):
    sc, s2, inv = _scale_tables(oracle)
    _match2(lambda q, qd, k, d, ss, si: oracle.search_by_bow(q, qd, k, d, ss, si, None, 50, 0.75, True),
            lambda q, qd, k, d, ss, si, has, F: oracle.search_for_triangulation(q, qd, k, d, None, ss, si, has, F, 150.0, 100.0, sc,
                                                                               s2, False, True),
            lambda q, qd, k, d, b, ur: oracle.search_in_area(q, qd, k, d, b, ur, inv, True, 50),
            oracle.distinctive_descriptors)


@pytest.mark.gpu
def test_gpu_reproduces_match2_goldens(oracle):
    from orb_slam2_amd import api
    sc, s2, inv = _scale_tables(oracle)
    m = api.ORBmatcher(0.75, True)
    lib, C = api.load_library(), api.C

    def bow(q, qd, k, d, ss, si):
        q = np.ascontiguousarray(q); qd = np.ascontiguousarray(qd); k = np.ascontiguousarray(k); d = np.ascontiguousarray(d)
        a = np.zeros(len(k), np.int32); qm = np.zeros(len(q), np.int32); n = C.c_int32()
        assert lib.orbgpu_search_by_bow(m._h, api._p(q), api._p(qd), len(q), api._p(k), api._p(d), len(k), api._p(ss), api._p(si),
                                        len(ss) - 1, None, 50, 0.75, 1, api._p(a), api._p(qm), C.byref(n)) == 0
        return n.value, a, qm

    def tri(q, qd, k, d, ss, si, has, F):
        q = np.ascontiguousarray(q); qd = np.ascontiguousarray(qd); k = np.ascontiguousarray(k); d = np.ascontiguousarray(d)
        F = np.ascontiguousarray(F, np.float32).reshape(9)
        qm = np.zeros(len(q), np.int32); n = C.c_int32()
        assert lib.orbgpu_search_for_triangulation(m._h, api._p(q), api._p(qd), len(q), api._p(k), api._p(d), None, len(k), api._p(ss),
                                                   api._p(si), len(ss) - 1, api._p(has), api._p(F), 150.0, 100.0, api._p(sc),
                                                   api._p(s2), len(sc), 0, 1, api._p(qm), C.byref(n)) == 0
        return n.value, qm

    _match2(bow, tri, lambda q, qd, k, d, b, ur: m.SearchInArea(q, qd, k, d, b, ur, inv, True, 50), m.ComputeDistinctiveDescriptors)


@pytest.mark.gpu
def test_gpu_reproduces_extraction_goldens():
    from orb_slam2_amd.api import ORBextractor
    g = _load("extract_320x240.npz")
    ex = ORBextractor(int(g["params"][0]), float(g["scale"]), int(g["params"][1]), int(g["params"][2]), int(g["params"][3]),
                      max_width=320, max_height=240)
    for i in range(2):
        k, d = ex(g["images"][i])
        ref = g["kps%d" % i]
        assert len(k) == len(ref)
        for f in ("x", "y", "size", "response", "octave", "class_id"):
            assert np.array_equal(k[f], ref[f]), f
        da = np.abs(k["angle"] - ref["angle"])
        assert np.minimum(da, 360 - da).max() <= np.degrees(1e-4)
        assert int((d != g["desc%d" % i]).any(axis=1).sum()) <= 0.001 * len(d)
    g = _load("extract_noise_300x260.npz")
    ex = ORBextractor(int(g["params"][0]), 1.2, 8, 20, 7, max_width=300, max_height=260)
    k, d = ex(g["image"])
    assert np.array_equal(k["x"], g["kps"]["x"]) and np.array_equal(k["y"], g["kps"]["y"])
    assert np.array_equal(k["response"], g["kps"]["response"]) and np.array_equal(k["octave"], g["kps"]["octave"])
    assert int((d != g["desc"]).any(axis=1).sum()) <= 0.001 * len(d)


@pytest.mark.gpu
def test_gpu_reproduces_match_goldens():
    from orb_slam2_amd.api import ORBmatcher
    e = _load("extract_320x240.npz"); g = _load("match_320x240.npz")
    b = np.array([0, 0, 320, 240], np.float32)
    m = ORBmatcher(0.8, True)
    for mode in (0, 1, 2):
        n, a = m.SearchByProjection(mode, g["q%d" % mode], e["desc0"], e["kps1"], e["desc1"], b, th_dist=100)
        assert n == int(g["n%d" % mode]) and np.array_equal(a, g["assign%d" % mode])
    m9 = ORBmatcher(0.9, True)
    prev = np.stack([e["kps0"]["x"], e["kps0"]["y"]], axis=1).astype(np.float32)
    n, m12, po = m9.SearchForInitialization(e["kps0"], e["desc0"], e["kps1"], e["desc1"], b, prev, 100)
    assert n == int(g["init_n"]) and np.array_equal(m12, g["init_m12"]) and np.array_equal(po, g["init_prev"])
    bi, b1, b2 = m9.BruteForce(g["bf_Q"], g["bf_T"])
    assert np.array_equal(bi, g["bf_idx"]) and np.array_equal(b1, g["bf_d1"]) and np.array_equal(b2, g["bf_d2"])


def _bow_golden(transform):
    """transform(voc dict, features, levelsup) -> word ids, node ids, (bow ids, values), (fv nodes, starts, features)."""
    g = _load("bow_k8_L3.npz")
    for tag in "abc":
        weighting, scoring, levelsup = (int(x) for x in g["cfg_" + tag])
        voc = dict(k=int(g["k"]), L=int(g["L"]), parent=g["parent"], is_leaf=g["is_leaf"], desc=g["node_desc"], weight=g["weight"],
                   weighting=weighting, scoring=scoring)
        wid, nid, (bi, bv), (fn, fs, ff) = transform(voc, g["feat"], levelsup)
        assert np.array_equal(wid, g["wid_" + tag]) and np.array_equal(nid, g["nid_" + tag])
        assert np.array_equal(bi, g["bi_" + tag])
        assert np.array_equal(np.asarray(bv).view(np.uint64), g["bv_" + tag].view(np.uint64))
        assert np.array_equal(fn, g["fn_" + tag]) and np.array_equal(fs, g["fs_" + tag]) and np.array_equal(ff, g["ff_" + tag])
        assert len(bi) > 30 and len(fn) >= 1


def test_oracle_reproduces_vocabulary_goldens(oracle):
    _bow_golden(oracle.voc_transform)


@pytest.mark.gpu
def test_gpu_reproduces_vocabulary_goldens():
    from orb_slam2_amd.api import ORBVocabulary

    def transform(voc, feat, levelsup):
        v = ORBVocabulary.from_arrays(voc["k"], voc["L"], voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"],
                                      scoring=voc["scoring"], weighting=voc["weighting"])
        return v.transform_arrays(feat, levelsup)

    _bow_golden(transform)
