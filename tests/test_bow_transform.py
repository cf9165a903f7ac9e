"""DBoW2 transform: Frame::ComputeBoW src/Frame.cc:750-765 =
TemplatedVocabulary<FORB>::transform(features, BowVector&, FeatureVector&, 4)
(Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1136-1216, 1240-1285).  ORBvoc.txt is not in the image (blocked),
so the vocabularies are synthetic: random k-ary trees in DBoW2's node numbering, regular and irregular (variable child
counts, leaves above depth L, stopped words with weight 0), every weighting / scoring combination.
CPU: the C oracle against a literal Python restatement with dict-based maps.  GPU: k_bow_transform + host vector builder
against the oracle, bit-exact including the double-precision values."""
import numpy as np
import pytest
import oracle


def random_vocabulary(rng, k=10, L=3, irregular=False, weighting=0, scoring=0, stop=0.05):
    """nodes in creation order like HKmeansStep / loadFromTextFile: children appended after their parent"""
    children, depth = [[]], [0]
    frontier = [0]
    while frontier:
        nxt = []
        for p in frontier:
            if depth[p] >= L:
                continue
            nc = k if not irregular else int(rng.integers(0, k + 1)) if depth[p] > 0 else k
            for _ in range(nc):
                children.append([]); depth.append(depth[p] + 1)
                children[p].append(len(children) - 1); nxt.append(len(children) - 1)
        frontier = nxt
    n = len(children)
    begin = np.zeros(n + 1, np.int32)
    for i in range(n):
        begin[i + 1] = begin[i] + len(children[i])
    desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    weight = rng.uniform(0.1, 9.0, n)
    weight[rng.uniform(size=n) < stop] = 0.0                     # stopWords()
    word_id = np.zeros(n, np.uint32); nw = 0
    for i in range(n):
        if not children[i]:
            word_id[i] = nw; nw += 1
    return dict(n_nodes=n, k=k, L=L, child_begin=begin, child_ids=np.array([c for ch in children for c in ch], np.uint32),
                desc=desc, weight=weight, word_id=word_id, weighting=weighting, scoring=scoring), children


def py_transform(voc, children, feats, levelsup):
    """literal restatement with Python dicts standing in for the std::maps"""
    def dist(a, b): return int(np.unpackbits(np.bitwise_xor(a, b)).sum())
    bow, fv = {}, {}
    per = []
    for i, f in enumerate(feats):
        node, level, nid = 0, 0, 0
        while children[node]:
            level += 1
            best, bd = children[node][0], dist(f, voc["desc"][children[node][0]])
            for c in children[node][1:]:
                d = dist(f, voc["desc"][c])
                if d < bd: bd, best = d, c
            node = best
            if level == voc["L"] - levelsup: nid = node
        wid, w = int(voc["word_id"][node]), float(voc["weight"][node])
        per.append((wid, w, nid))
        if w > 0:
            if voc["weighting"] in (0, 1): bow[wid] = bow[wid] + w if wid in bow else w
            elif wid not in bow: bow[wid] = w
            fv.setdefault(nid, []).append(i)
    keys = sorted(bow)
    vals = [bow[k] for k in keys]
    norm = {1: 2, 5: 0}.get(voc["scoring"], 1)
    if voc["weighting"] in (0, 1) and keys and norm == 0:
        vals = [v / float(len(keys)) for v in vals]
    if norm:
        s = 0.0
        for v in vals: s += abs(v) if norm == 1 else v * v
        if norm == 2: s = float(np.sqrt(np.float64(s)))
        if s > 0.0: vals = [v / s for v in vals]
    return per, (keys, vals), {k: fv[k] for k in sorted(fv)}


CASES = [(10, 3, False, 0, 0, 4), (10, 3, False, 0, 0, 1), (6, 4, True, 0, 0, 2), (4, 5, True, 1, 1, 3), (20, 2, False, 2, 0, 1),
         (17, 2, True, 3, 5, 1), (10, 3, False, 1, 5, 2), (3, 6, False, 0, 2, 4)]


@pytest.mark.parametrize("k,L,irr,weighting,scoring,levelsup", CASES)
def test_oracle_transform_equals_python_restatement(k, L, irr, weighting, scoring, levelsup):
    rng = np.random.default_rng(k * 100 + L)
    voc, children = random_vocabulary(rng, k, L, irr, weighting, scoring)
    feats = rng.integers(0, 256, (150, 32), dtype=np.uint8)
    feats[:20] = voc["desc"][rng.integers(1, voc["n_nodes"], 20)]          # exact hits and ties with node descriptors
    wid, w, nid, (bw, bv), (fn, fb, fi) = oracle.bow_transform(voc, feats, levelsup)
    per, (keys, vals), fv = py_transform(voc, children, feats, levelsup)
    assert [(int(a), float(b), int(c)) for a, b, c in zip(wid, w, nid)] == per
    assert bw.tolist() == keys and bv.tolist() == vals                        # doubles compared exactly
    assert fn.tolist() == list(fv) and [fi[fb[i]:fb[i + 1]].tolist() for i in range(len(fn))] == list(fv.values())


@pytest.mark.gpu
@pytest.mark.parametrize("k,L,irr,weighting,scoring,levelsup", CASES)
def test_gpu_transform_equals_oracle(k, L, irr, weighting, scoring, levelsup):
    from orb_slam2_detailed_comments_amd import ORBextractor, ORBVocabulary
    rng = np.random.default_rng(k * 100 + L + 7)
    voc, _ = random_vocabulary(rng, k, L, irr, weighting, scoring)
    ex = ORBextractor(500)
    V = ORBVocabulary(ex, **voc)
    for n in (0, 1, 15, 16, 17, 1000):
        feats = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        if n >= 16:
            feats[:8] = voc["desc"][rng.integers(1, voc["n_nodes"], 8)]
        wid, w, nid = V.transform_features(feats, levelsup)
        owid, ow, onid, (obw, obv), (ofn, ofb, ofi) = oracle.bow_transform(voc, feats, levelsup)
        assert np.array_equal(wid, owid) and np.array_equal(w.view(np.uint64), ow.view(np.uint64)) and np.array_equal(nid, onid)
        (bw, bv), (fn, fb, fi) = V.transform(feats, levelsup)
        assert np.array_equal(bw, obw) and np.array_equal(bv.view(np.uint64), obv.view(np.uint64))
        assert np.array_equal(fn, ofn) and np.array_equal(fb, ofb) and np.array_equal(fi, ofi)


@pytest.mark.gpu
def test_gpu_transform_on_extracted_descriptors_feeds_search_by_bow(tmp_path):
    """end to end: extract -> ComputeBoW -> SearchByBoW(KF, F) with the FeatureVectors the transform produced; also the
    ORBvoc.txt text format round trip (loadFromTextFile numbering)"""
    from orb_slam2_detailed_comments_amd import ORBextractor, ORBVocabulary, ORBmatcher, OrbxError, synth
    rng = np.random.default_rng(5)
    frames = synth.stream(640, 480, 2, stream_id=51)
    ex = ORBextractor(1000, max_batch=2)
    (k1, d1), (k2, d2) = ex.extract_batch(frames)
    # vocabulary "trained" on the first frame's descriptors so that neighbours share nodes
    voc, children = random_vocabulary(rng, 8, 3, False)
    voc["desc"][1:] = d1[rng.integers(0, len(d1), voc["n_nodes"] - 1)]
    path = tmp_path / "voc.txt"
    parent = np.zeros(voc["n_nodes"], int)
    for p, ch in enumerate(children):
        for c in ch: parent[c] = p
    with open(path, "w") as f:
        f.write("8 3 0 0\n")
        for i in range(1, voc["n_nodes"]):
            f.write(f"{parent[i]} {int(not children[i])} " + " ".join(str(int(b)) for b in voc["desc"][i]) + f" {float(voc['weight'][i])!r}\n")
    V = ORBVocabulary.load_text(ex, str(path))
    fvs = []
    for d in (d1, d2):
        (bw, bv), fv = V.transform(d, 1)
        obw, obv = oracle.bow_transform(voc, d, 1)[3]
        assert np.array_equal(bw, obw) and np.array_equal(bv.view(np.uint64), obv.view(np.uint64))
        assert abs(bv.sum() - 1.0) < 1e-12                                          # L1-normalised
        fvs.append(fv)
    kf = dict(keys_un=k1, desc=d1, has_map_point=np.ones(len(k1), np.uint8), feat_vec=fvs[0])
    m = ORBmatcher(0.7, True, extractor=ex)
    n, out = m.SearchByBoW(kf, k2, d2, fvs[1])
    to_dict = lambda fv: {int(fv[0][i]): fv[2][fv[1][i]:fv[1][i + 1]].tolist() for i in range(len(fv[0]))}
    on, oout = oracle.search_by_bow_kf_frame(dict(kf, feat_vec=to_dict(fvs[0])), k2, d2, to_dict(fvs[1]), 0.7, True)
    assert n == on and np.array_equal(out, oout) and n > 20
    bad = dict(voc, child_ids=voc["child_ids"].copy()); bad["child_ids"][0] = 0     # child not greater than its parent
    with pytest.raises(OrbxError):
        ORBVocabulary(ex, **bad)
