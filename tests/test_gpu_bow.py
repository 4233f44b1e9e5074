"""GPU parity of the vocabulary (bag-of-words) transform against the CPU oracle: ids bit-exact, BowVector
doubles bit-identical.  The vocabularies are seeded stand-ins (the reference's ORBvoc.txt is not available)."""
import numpy as np
import pytest

from orb_slam2_amd import synth
from hiputil import Hip as _Hip

pytestmark = pytest.mark.gpu


def _features(voc, n, seed, flip=0.02):
    rng = np.random.default_rng(seed)
    leaves = voc["desc"][voc["is_leaf"] == 1]
    d = leaves[rng.integers(0, max(len(leaves) // 3, 1), n)].copy()
    bits = np.unpackbits(d, axis=1)
    bits ^= (rng.random(bits.shape) < flip).astype(np.uint8)
    d = np.packbits(bits, axis=1)
    d[::17] = voc["desc"][rng.integers(0, len(voc["desc"]), len(d[::17]))]
    d[::29] = rng.integers(0, 256, (len(d[::29]), 32), dtype=np.uint8)
    return d


def _make(voc):
    from orb_slam2_amd.api import ORBVocabulary
    return ORBVocabulary.from_arrays(voc["k"], voc["L"], voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"],
                                     scoring=voc["scoring"], weighting=voc["weighting"])


def _same(got, ref):
    gw, gn, (gbi, gbv), (gfn, gfs, gff) = got
    rw, rn, (rbi, rbv), (rfn, rfs, rff) = ref
    assert np.array_equal(gw, rw), "word ids differ"
    assert np.array_equal(gn, rn), "node ids differ"
    assert np.array_equal(gbi, rbi), "BowVector ids differ"
    assert np.array_equal(gbv.view(np.uint64), rbv.view(np.uint64)), "BowVector values are not bit-identical"
    assert np.array_equal(gfn, rfn) and np.array_equal(gfs, rfs) and np.array_equal(gff, rff), "FeatureVector differs"


@pytest.mark.parametrize("k,L,ragged,weighting,scoring,levelsup,n", [
    (10, 3, True, 0, 0, 1, 1000), (10, 4, False, 0, 0, 2, 1100), (4, 4, True, 1, 1, 2, 700), (6, 3, True, 2, 0, 4, 333),
    (5, 3, True, 3, 5, 1, 64), (9, 2, True, 0, 5, 1, 1), (3, 5, True, 0, 2, 0, 2500), (20, 2, False, 0, 0, 1, 900),
    (17, 3, True, 0, 3, 1, 2049)])
def test_transform_matches_oracle(oracle, k, L, ragged, weighting, scoring, levelsup, n):
    voc = synth.vocabulary(k, L, seed=k * 10 + L, ragged=ragged, scoring=scoring, weighting=weighting)
    desc = _features(voc, n, seed=n + k)
    v = _make(voc)
    assert v.size() == int(voc["is_leaf"].sum()) and not v.empty()
    assert (v.getBranchingFactor(), v.getDepthLevels(), v.getScoringType(), v.getWeightingType()) == (k, L, scoring, weighting)
    got = v.transform_arrays(desc, levelsup)
    ref = oracle.voc_transform(voc, desc, levelsup)
    _same(got, ref)
    assert len(ref[2][0]) > 0 or n < 5
    # the map view: {WordId: value}, {NodeId: [features]}
    bow, fv = v.transform(desc, levelsup)
    assert list(bow) == list(ref[2][0]) and sum(len(x) for x in fv.values()) == ref[3][1][-1]
    # idempotence
    _same(v.transform_arrays(desc, levelsup), got)


def test_text_file_round_trip(oracle, tmp_path):
    from orb_slam2_amd.api import ORBVocabulary
    voc = synth.vocabulary(10, 3, seed=4)
    desc = _features(voc, 500, seed=9)
    ref = oracle.voc_transform(voc, desc, 1)
    for trailing in (True, False):  # with and without the final newline: the blank last line adds no node
        path = tmp_path / ("voc%d.txt" % trailing)
        synth.save_vocabulary_text(str(path), voc, trailing_newline=trailing)
        v = ORBVocabulary()
        assert v.empty()
        assert v.loadFromTextFile(str(path)), v.LastError()
        assert v.size() == int(voc["is_leaf"].sum())
        _same(v.transform_arrays(desc, 1), ref)
    v = ORBVocabulary()
    assert not v.loadFromTextFile(str(tmp_path / "missing.txt")) and "cannot open" in v.LastError()
    bad = tmp_path / "bad.txt"
    bad.write_text("10 3 0 0\n0 1 1 2 3\n")
    assert not v.loadFromTextFile(str(bad)) and "line 2" in v.LastError()
    bad.write_text("30 3 0 0\n")
    assert not v.loadFromTextFile(str(bad)) and "outside" in v.LastError()


def test_edge_cases(oracle):
    from orb_slam2_amd import api
    voc = synth.vocabulary(10, 3, seed=5)
    v = _make(voc)
    d = _features(voc, 50, 1)
    wid, nid, (bi, bv), (fn, fs, ff) = v.transform_arrays(np.zeros((0, 32), np.uint8), 1)
    assert len(wid) == 0 and len(bi) == 0 and len(fn) == 0 and list(fs) == [0]
    stopped = dict(voc); stopped["weight"] = np.zeros_like(voc["weight"])
    _same(_make(stopped).transform_arrays(d, 1), oracle.voc_transform(stopped, d, 1))
    got = v.transform_arrays(d, 3)  # levelsup >= L: everything hangs off the root
    _same(got, oracle.voc_transform(voc, d, 3))
    assert (got[1] == 0).all()
    # a vocabulary without nodes is empty(): transform clears both containers
    e = api.ORBVocabulary.from_arrays(10, 3, np.zeros(0, np.int32), np.zeros(0, np.uint8), np.zeros((0, 32), np.uint8),
                                      np.zeros(0, np.float64))
    assert e.empty()
    wid, nid, (bi, bv), (fn, fs, ff) = e.transform_arrays(d, 1)
    assert (wid == -1).all() and len(bi) == 0 and len(fn) == 0
    # refused inputs
    bad = voc["parent"].copy(); bad[3] = 50
    with pytest.raises(api.OrbGpuError, match="earlier node"):
        api.ORBVocabulary.from_arrays(10, 3, bad, voc["is_leaf"], voc["desc"], voc["weight"])
    leaf = voc["is_leaf"].copy(); leaf[0] ^= 1
    with pytest.raises(api.OrbGpuError, match="children"):
        api.ORBVocabulary.from_arrays(10, 3, voc["parent"], leaf, voc["desc"], voc["weight"])
    with pytest.raises(api.OrbGpuError, match="16384"):
        v.transform_arrays(np.zeros((16385, 32), np.uint8), 1)


def test_largest_set(oracle):
    """16384 features in one set: the sort keys fill 128 KB of LDS."""
    voc = synth.vocabulary(10, 4, seed=8, ragged=False)
    desc = _features(voc, 16384, seed=2, flip=0.05)
    _same(_make(voc).transform_arrays(desc, 2), oracle.voc_transform(voc, desc, 2))


def test_batch_device_equals_single(oracle):
    """The device entry point on the layout the extractor's batch call writes: sets of different sizes (one empty)."""
    voc = synth.vocabulary(10, 3, seed=6)
    v = _make(voc)
    cap, counts = 1200, [1000, 0, 1200, 37, 512]
    ns = len(counts)
    desc = np.zeros((ns, cap, 32), np.uint8)
    for s, c in enumerate(counts):
        desc[s, :c] = _features(voc, max(c, 1), seed=20 + s)[:c]
    hip = _Hip()
    try:
        d_desc = hip.upload(desc)
        d_cnt = hip.upload(np.asarray(counts, np.int32))
        wid, nid, bi, fn, ff = [hip.alloc(ns * cap * 4) for _ in range(5)]
        fs = hip.alloc(ns * (cap + 1) * 4)
        bv = hip.alloc(ns * cap * 8)
        nb, nf = hip.alloc(ns * 4), hip.alloc(ns * 4)
        v.transform_batch_device(d_desc, d_cnt, ns, cap, 1, [wid, nid, bi, bv, nb, fn, fs, ff, nf])
        v.sync()
        wid, nid, bi, fn, ff = [hip.download(p, (ns, cap), np.int32) for p in (wid, nid, bi, fn, ff)]
        fs = hip.download(fs, (ns, cap + 1), np.int32)
        bv = hip.download(bv, (ns, cap), np.float64)
        nb, nf = hip.download(nb, ns, np.int32), hip.download(nf, ns, np.int32)
    finally:
        hip.free()
    for s, c in enumerate(counts):
        ref = oracle.voc_transform(voc, desc[s, :c], 1)
        b, f = int(nb[s]), int(nf[s])
        kept = int(fs[s, f])
        _same((wid[s, :c], nid[s, :c], (bi[s, :b], bv[s, :b]), (fn[s, :f], fs[s, :f + 1], ff[s, :kept])), ref)


def test_compute_bow_feeds_search_by_bow(oracle):
    """Frame::ComputeBoW -> ORBmatcher::SearchByBoW (src/ORBmatcher.cc:159-288) end to end on two synthetic frames:
    the FeatureVectors of the device transform select the same candidates, hence the same matches, as the oracle's."""
    from orb_slam2_amd import api
    seq = synth.sequence(640, 480, 2, seed=1)
    oe = oracle.Extractor(1000)
    (k1, d1), (k2, d2) = [oe.extract(f) for f in seq]
    # vocabulary whose words are descriptors of a third view of the scene, so that both frames spread over many nodes
    voc = synth.vocabulary(10, 3, seed=12, ragged=False)
    v = _make(voc)
    fvs = []
    for d in (d1, d2):
        bow, fv = v.transform(d, 2)
        _, _, _, (rfn, rfs, rff) = oracle.voc_transform(voc, d, 2)
        assert sorted(fv) == list(rfn)
        fvs.append({int(k): np.asarray(x, np.int32) for k, x in fv.items()})
    m = api.ORBmatcher(0.7, True)
    valid1 = np.ones(len(k1), np.uint8)
    n, a1, m12 = m.SearchByBoW(fvs[0], k1, d1, valid1, fvs[1], k2, d2, None, False)
    qi, qs, ss, si = api.bow_segments(fvs[0], fvs[1], valid1)
    q = np.zeros(len(qi), oracle.WQUERY_DTYPE)
    q["angle"] = k1["angle"][qi]; q["minLevel"] = qs; q["flags"] = 1
    rn, rassign, rqm = oracle.search_by_bow(q, d1[qi], k2.view(oracle.KEYPOINT_DTYPE), d2, ss, si, None, 50, 0.7, True)
    assert n == rn
    rm12 = np.full(len(k1), -1, np.int32); rm12[qi[rqm >= 0]] = rqm[rqm >= 0]
    assert np.array_equal(m12, rm12)
