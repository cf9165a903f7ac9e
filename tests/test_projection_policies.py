"""Projection-guided back-end policies: Fuse src/ORBmatcher.cc:1100-1280 and :1282-1430,
SearchByProjection(KF, Scw, ...) :415-560, SearchBySim3 :1433-1690, SearchByProjection(Frame, KF, ...) :1873-2020.

The MapPoints are synthesised from the keypoints the HIP path extracts from frame t: their projection into frame t+1 is
the known image translation plus noise, their descriptor a slightly corrupted copy, their predicted level the octave
(sometimes off by one) -- the fields the reference computes with cv::Mat algebra before the part restated here.
CPU part: the C oracle against a pure-Python restatement; GPU part: the product path against the C oracle."""
import numpy as np
import pytest
import oracle

TH_LOW, TH_HIGH = 50, 100


def popcount_dist(a, b):
    return int(np.unpackbits(np.bitwise_xor(a, b)).sum())


def synth_keys(rng, n, w=640, h=480):
    k = np.zeros(n, oracle.KP_DTYPE)
    k["x"] = rng.uniform(16, w - 16, n).astype(np.float32); k["y"] = rng.uniform(16, h - 16, n).astype(np.float32)
    k["angle"] = rng.uniform(0, 360, n).astype(np.float32); k["octave"] = rng.integers(0, 8, n)
    return k, rng.integers(0, 256, (n, 32), dtype=np.uint8)


def make_target(keys, desc, rng, w=640, h=480):
    sf = (np.float32(1.2) ** np.arange(8)).astype(np.float32)
    return dict(keys_un=keys, desc=desc, bounds=(0.0, float(w), 0.0, float(h)), scale_factors=sf,
                inv_level_sigma2=(np.float32(1.0) / (sf * sf)).astype(np.float32),
                u_right=np.where(rng.uniform(size=len(keys)) < 0.4, keys["x"] - rng.uniform(2, 30, len(keys)), -1).astype(np.float32))


def points_from(keys, desc, rng, shift=(3.0, 2.0), nflip=10, p_valid=0.85, dup=0.15):
    """MapPoints seen in `keys`, projected into the other view (shifted), some duplicated so that features are contended"""
    idx = np.concatenate([np.arange(len(keys)), rng.choice(len(keys), int(dup * len(keys)))])
    rng.shuffle(idx)
    n = len(idx)
    uv = np.stack([keys["x"][idx] + shift[0], keys["y"][idx] + shift[1]], 1).astype(np.float32)
    uv += rng.normal(0, 1.0, uv.shape).astype(np.float32)
    d = desc[idx].copy()
    for row in d:
        for b in rng.integers(0, 256, nflip): row[b >> 3] ^= np.uint8(1 << (b & 7))
    level = np.clip(keys["octave"][idx] + rng.integers(0, 2, n), 0, 7).astype(np.int32)
    return dict(valid=(rng.uniform(size=n) < p_valid).astype(np.uint8), uv=uv, level=level, desc=d,
                u_right=(uv[:, 0] - rng.uniform(2, 30, n)).astype(np.float32), angle=keys["angle"][idx].copy()), idx


def py_grid_query(keys, x, y, r, lo=-1, hi=-1):
    """GetFeaturesInArea membership (order irrelevant for a strict-first-best scan only if we replay the cell order:
    the oracle's own grid is used for the order, this helper only checks membership)"""
    m = (np.abs(keys["x"] - np.float32(x)) < np.float32(r)) & (np.abs(keys["y"] - np.float32(y)) < np.float32(r))
    if lo > 0 or hi >= 0:
        m &= keys["octave"] >= lo
        if hi >= 0: m &= keys["octave"] <= hi
    return set(np.nonzero(m)[0].tolist())


def test_oracle_fuse_sim3_selection_properties():
    rng = np.random.default_rng(0)
    k2, d2 = synth_keys(rng, 500)
    tgt = make_target(k2, d2, rng)
    pts, src = points_from(k2, d2, rng, shift=(0.0, 0.0))
    th = 4.0
    n, best = oracle.fuse_sim3(tgt, pts, th)
    assert n == (best >= 0).sum() and n > 100
    for i in np.nonzero(best >= 0)[0][:200]:
        r = np.float32(th) * tgt["scale_factors"][pts["level"][i]]
        cands = [j for j in py_grid_query(k2, *pts["uv"][i], r) if pts["level"][i] - 1 <= k2["octave"][j] <= pts["level"][i]]
        dists = {j: popcount_dist(pts["desc"][i], d2[j]) for j in cands}
        assert best[i] in dists and dists[best[i]] == min(dists.values()) <= TH_LOW
    assert not (best[pts["valid"] == 0] >= 0).any()


def test_oracle_search_by_projection_sim3_respects_taken_and_order():
    rng = np.random.default_rng(1)
    k2, d2 = synth_keys(rng, 400)
    tgt = make_target(k2, d2, rng)
    pts, src = points_from(k2, d2, rng, shift=(0.0, 0.0), dup=0.5)
    matched = (rng.uniform(size=len(k2)) < 0.2).astype(np.uint8)
    pre = matched.copy()
    n, best = oracle.search_by_projection_sim3(tgt, pts, matched, 4)
    got = best[best >= 0]
    assert n == len(got) == len(set(got)) and not pre[got].any() and matched[got].all()
    assert (matched.astype(int) - pre.astype(int)).sum() == n


def test_oracle_search_by_sim3_is_mutual():
    rng = np.random.default_rng(2)
    k1, d1 = synth_keys(rng, 300)
    k2 = k1.copy(); k2["x"] += np.float32(3); k2["y"] += np.float32(2)
    d2 = d1.copy()
    for row in d2:
        for b in rng.integers(0, 256, 6): row[b >> 3] ^= np.uint8(1 << (b & 7))
    t1, t2 = make_target(k1, d1, rng), make_target(k2, d2, rng)
    p12 = dict(valid=(rng.uniform(size=300) < 0.8).astype(np.uint8), uv=np.stack([k1["x"] + 3, k1["y"] + 2], 1).astype(np.float32),
               level=k1["octave"].astype(np.int32), desc=d1)
    p21 = dict(valid=(rng.uniform(size=300) < 0.8).astype(np.uint8), uv=np.stack([k2["x"] - 3, k2["y"] - 2], 1).astype(np.float32),
               level=k2["octave"].astype(np.int32), desc=d2)
    n, m12 = oracle.search_by_sim3(t1, t2, p12, p21, 7.5)
    idx = np.nonzero(m12 >= 0)[0]
    assert n == len(idx) > 100 and (m12[idx] == idx).mean() > 0.95      # same ordering in both views by construction
    assert p12["valid"][idx].all() and p21["valid"][m12[idx]].all()


@pytest.mark.gpu
@pytest.mark.parametrize("fp", [0, 1])
def test_gpu_projection_policies_equal_oracle(fp):
    from orb_slam2_detailed_comments_amd import ORBextractor, ORBmatcher, synth
    frames = synth.stream(640, 480, 2, stream_id=41)
    ex = ORBextractor(1000, max_batch=2, fp_mode=fp)
    (k1, d1), (k2, d2) = ex.extract_batch(frames)
    rng = np.random.default_rng(11)
    sf = ex.GetScaleFactors()
    mk = lambda k, d: dict(make_target(k, d, rng, 678, 518), scale_factors=sf, inv_level_sigma2=ex.GetInverseScaleSigmaSquares())
    t1, t2 = mk(k1, d1), mk(k2, d2)
    # the stream moves the scene by (-3, -2) px per frame in image coordinates
    pts, _ = points_from(k1, d1, rng, shift=(-3.0, -2.0))
    m = ORBmatcher(0.7, True, extractor=ex)
    n, best = m.Fuse(t2, pts, 3.0)
    on, obest = oracle.fuse(t2, pts, 3.0, fp)
    assert n == on and np.array_equal(best, obest) and n > 100
    n, best = m.FuseSim3(t2, pts, 4.0)
    on, obest = oracle.fuse_sim3(t2, pts, 4.0)
    assert n == on and np.array_equal(best, obest) and n > 100
    matched = (rng.uniform(size=len(k2)) < 0.2).astype(np.uint8); omatched = matched.copy()
    n, best = m.SearchByProjectionSim3(t2, pts, matched, 10)
    on, obest = oracle.search_by_projection_sim3(t2, pts, omatched, 10)
    assert n == on and np.array_equal(best, obest) and np.array_equal(matched, omatched) and n > 100
    # SearchBySim3: one projected point per feature of each keyframe
    p12 = dict(valid=(rng.uniform(size=len(k1)) < 0.8).astype(np.uint8), uv=np.stack([k1["x"] - 3, k1["y"] - 2], 1).astype(np.float32),
               level=k1["octave"].astype(np.int32), desc=d1)
    p21 = dict(valid=(rng.uniform(size=len(k2)) < 0.8).astype(np.uint8), uv=np.stack([k2["x"] + 3, k2["y"] + 2], 1).astype(np.float32),
               level=k2["octave"].astype(np.int32), desc=d2)
    n, m12 = m.SearchBySim3(t1, t2, p12, p21, 7.5)
    on, om12 = oracle.search_by_sim3(t1, t2, p12, p21, 7.5)
    assert n == on and np.array_equal(m12, om12) and n > 50
    # SearchByProjection(Frame, KF): relocalisation
    has = (rng.uniform(size=len(k2)) < 0.3).astype(np.uint8); ohas = has.copy()
    for ori in (True, False):
        mm = ORBmatcher(0.7, ori, extractor=ex)
        h1, h2 = has.copy(), ohas.copy()
        n, mp = mm.SearchByProjectionKeyFrame(t2, pts, h1, 10.0, 100)
        on, omp = oracle.search_by_projection_kf(t2, pts, h2, 10.0, 100, ori)
        assert n == on and np.array_equal(mp, omp) and np.array_equal(h1, h2) and n > 100


@pytest.mark.gpu
@pytest.mark.parametrize("fp", [0, 1])
def test_gpu_fuse_batch_equals_single_calls_and_oracle(fp):
    """orbx_fuse_batch / orbx_fuse_sim3_batch (the Fuse loops of LocalMapping::SearchInNeighbors, src/LocalMapping.cc:750-768, and
    LoopClosing::SearchAndFuse as ONE call): per problem the same best_idx / count as the single call and as the oracle; targets
    of different sizes, an empty target, an empty point set, the same points against several keyframes"""
    from orb_slam2_detailed_comments_amd import ORBextractor, ORBmatcher, OrbxError, synth
    frames = synth.stream(640, 480, 4, stream_id=43)
    ex = ORBextractor(1000, max_batch=4, fp_mode=fp)
    res = ex.extract_batch(frames)
    rng = np.random.default_rng(17)
    sf = ex.GetScaleFactors()
    mk = lambda k, d: dict(make_target(k, d, rng, 678, 518), scale_factors=sf, inv_level_sigma2=ex.GetInverseScaleSigmaSquares())
    tg = [mk(k, d) for k, d in res]
    tg[2] = mk(res[2][0][:300], res[2][1][:300])                       # a smaller keyframe
    tg.append(mk(res[0][0][:0], res[0][1][:0]))                          # and one without features
    # SearchInNeighbors: the CURRENT keyframe's points against every neighbour (same point set), plus one other set and an empty one
    p0, _ = points_from(res[0][0], res[0][1], rng, shift=(-3.0, -2.0))
    p1, _ = points_from(res[1][0], res[1][1], rng, shift=(-3.0, -2.0))
    empty = {kk: v[:0] for kk, v in p0.items()}
    probs = [(tg[1], p0), (tg[2], p0), (tg[3], p0), (tg[2], p1), (tg[4], p0), (tg[1], empty)]
    m = ORBmatcher(0.7, True, extractor=ex)
    for sim3, th, single, orc in ((False, 3.0, m.Fuse, lambda t, p: oracle.fuse(t, p, 3.0, fp)),
                                  (True, 4.0, m.FuseSim3, lambda t, p: oracle.fuse_sim3(t, p, 4.0))):
        ns, bests = m.FuseBatch([t for t, _ in probs], [p for _, p in probs], th, sim3=sim3)
        assert len(ns) == len(probs)
        for k, (t, p) in enumerate(probs):
            n1, b1 = single(t, p, th)
            assert ns[k] == n1 and np.array_equal(bests[k], b1), f"problem {k} differs from the single call"
            if len(t["keys_un"]) and len(p["valid"]):
                on, ob = orc(t, p)
                assert ns[k] == on and np.array_equal(bests[k], ob), f"problem {k} differs from the oracle"
        assert ns[0] > 100 and ns[4] == 0 and ns[5] == 0
    assert m.FuseBatch([], [], 3.0) == ([], [])
    other = dict(tg[1], bounds=(0.0, 700.0, 0.0, 518.0))
    with pytest.raises(OrbxError):
        m.FuseBatch([tg[1], other], [p0, p0], 3.0)                       # the targets of a batch share the image bounds


@pytest.mark.gpu
def test_gpu_projection_policies_edge_cases():
    from orb_slam2_detailed_comments_amd import ORBextractor, ORBmatcher, OrbxError
    rng = np.random.default_rng(12)
    k, d = synth_keys(rng, 50)
    tgt = make_target(k, d, rng)
    pts, _ = points_from(k, d, rng, shift=(0.0, 0.0))
    m = ORBmatcher(0.7, True, extractor=ORBextractor(500))
    none = dict(pts, valid=np.zeros(len(pts["valid"]), np.uint8))
    assert m.FuseSim3(tgt, none, 4.0)[0] == 0
    empty = {kk: v[:0] for kk, v in pts.items()}
    assert m.Fuse(tgt, empty, 3.0)[0] == 0
    far = dict(pts, uv=(pts["uv"] + np.float32(5000)))                 # every projection outside the grid
    assert m.FuseSim3(tgt, far, 4.0)[0] == 0
    with pytest.raises(OrbxError):
        m.SearchBySim3(tgt, tgt, pts, pts, 7.5)                        # needs one point per keyframe feature
