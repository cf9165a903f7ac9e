"""BoW-guided matcher policies: SearchByBoW(KF, F) src/ORBmatcher.cc:248-410, SearchByBoW(KF, KF)
:722-866, SearchForTriangulation :879-1087.

CPU part (no GPU): the C oracle against an independent pure-Python restatement of the same loops on small random inputs.
GPU part: the product path (GPU Hamming matrix + host selection, through the C ABI) against the C oracle on keypoints and
descriptors extracted by the HIP path; a synthetic vocabulary (node = a few descriptor bits) stands in for ORBvoc.txt, which
is not in the image (the policies only read the node ids)."""
import numpy as np
import pytest
import oracle

TH_LOW, HISTO = 50, 30


def popcount_dist(a, b):
    return int(np.unpackbits(np.bitwise_xor(a, b)).sum())


def make_featvec(desc, bits=5, shuffle_rng=None):
    """toy vocabulary: node id = top `bits` bits of byte 0 xor byte 7 (similar descriptors often share a node)"""
    fv = {}
    order = np.arange(len(desc))
    if shuffle_rng is not None:
        order = shuffle_rng.permutation(len(desc))
    for i in order:
        node = int((desc[i, 0] ^ desc[i, 7]) >> (8 - bits)) * 3 + 1
        fv.setdefault(node, []).append(int(i))
    return fv


def three_maxima(sizes):
    m1 = m2 = m3 = 0; i1 = i2 = i3 = -1
    for i, s in enumerate(sizes):
        if s > m1: m3, m2, m1, i3, i2, i1 = m2, m1, s, i2, i1, i
        elif s > m2: m3, m2, i3, i2 = m2, s, i2, i
        elif s > m3: m3, i3 = s, i
    if m2 < np.float32(0.1) * np.float32(m1): i2 = i3 = -1
    elif m3 < np.float32(0.1) * np.float32(m1): i3 = -1
    return i1, i2, i3


def rot_bin(a1, a2):
    rot = np.float32(a1) - np.float32(a2)
    if rot < 0.0: rot = np.float32(rot + np.float32(360.0))
    v = float(np.float32(rot * np.float32(HISTO / 360.0)))
    b = int(np.floor(abs(v) + 0.5)) * (1 if v >= 0 else -1)   # round(): half away from zero
    return 0 if b == HISTO else b


def py_bow_kf_frame(kf, fk, fd, ffv, ratio, ori):
    out = np.full(len(fk), -1, np.int32); n = 0
    hist = [[] for _ in range(HISTO)]
    for node in sorted(set(kf["feat_vec"]) & set(ffv)):
        for ik in kf["feat_vec"][node]:
            if not kf["has_map_point"][ik]: continue
            b1 = b2 = 256; bi = -1
            for jf in ffv[node]:
                if out[jf] >= 0: continue
                d = popcount_dist(kf["desc"][ik], fd[jf])
                if d < b1: b2, b1, bi = b1, d, jf
                elif d < b2: b2 = d
            if b1 <= TH_LOW and np.float32(b1) < np.float32(ratio) * np.float32(b2):
                out[bi] = ik; n += 1
                if ori: hist[rot_bin(kf["keys_un"]["angle"][ik], fk["angle"][bi])].append(bi)
    if ori:
        keep = three_maxima([len(h) for h in hist])
        for i in range(HISTO):
            if i in keep: continue
            for j in hist[i]: out[j] = -1; n -= 1
    return n, out


def random_kf(rng, n, p_mp=0.7):
    k = np.zeros(n, oracle.KP_DTYPE)
    k["x"] = rng.uniform(20, 600, n).astype(np.float32); k["y"] = rng.uniform(20, 440, n).astype(np.float32)
    k["angle"] = rng.uniform(0, 360, n).astype(np.float32); k["octave"] = rng.integers(0, 8, n)
    d = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    return dict(keys_un=k, desc=d, has_map_point=(rng.uniform(size=n) < p_mp).astype(np.uint8),
                u_right=np.where(rng.uniform(size=n) < 0.3, rng.uniform(0, 600, n), -1).astype(np.float32))


def perturbed_copy(rng, kf, nflip=12, drop=0.2):
    """second view of the same features: a few descriptor bits flipped, small angle change, some features dropped"""
    keep = rng.uniform(size=len(kf["desc"])) > drop
    d = kf["desc"][keep].copy()
    for row in d:
        for b in rng.integers(0, 256, nflip): row[b >> 3] ^= np.uint8(1 << (b & 7))
    k = kf["keys_un"][keep].copy()
    k["angle"] = np.mod(k["angle"] + rng.normal(0, 4, len(k)).astype(np.float32) + np.float32(10), np.float32(360)).astype(np.float32)
    k["x"] += rng.normal(0, 2, len(k)).astype(np.float32)
    perm = rng.permutation(len(k))
    return dict(keys_un=k[perm], desc=d[perm], has_map_point=(rng.uniform(size=len(k)) < 0.6).astype(np.uint8),
                u_right=np.where(rng.uniform(size=len(k)) < 0.3, rng.uniform(0, 600, len(k)), -1).astype(np.float32))


@pytest.mark.parametrize("seed,ratio,ori", [(0, 0.7, True), (1, 0.9, False), (2, 0.75, True)])
def test_oracle_bow_kf_frame_equals_python_restatement(seed, ratio, ori):
    rng = np.random.default_rng(seed)
    kf = random_kf(rng, 300)
    f = perturbed_copy(rng, kf)
    kf["feat_vec"] = make_featvec(kf["desc"], shuffle_rng=rng); ffv = make_featvec(f["desc"], shuffle_rng=rng)
    n, out = oracle.search_by_bow_kf_frame(kf, f["keys_un"], f["desc"], ffv, ratio, ori)
    pn, pout = py_bow_kf_frame(kf, f["keys_un"], f["desc"], ffv, ratio, ori)
    assert n == pn and np.array_equal(out, pout)
    assert n > 10 and (out >= 0).sum() == n


def test_oracle_bow_edge_cases():
    rng = np.random.default_rng(5)
    kf = random_kf(rng, 50); f = perturbed_copy(rng, kf)
    kf["feat_vec"] = {1: list(range(50))}
    assert oracle.search_by_bow_kf_frame(kf, f["keys_un"], f["desc"], {2: list(range(len(f["desc"])))}, 0.7)[0] == 0   # no common node
    assert oracle.search_by_bow_kf_frame(kf, f["keys_un"][:0], f["desc"][:0], {}, 0.7)[0] == 0
    kf0 = dict(kf, has_map_point=np.zeros(50, np.uint8))
    assert oracle.search_by_bow_kf_frame(kf0, f["keys_un"], f["desc"], make_featvec(f["desc"]), 0.7)[0] == 0
    f["feat_vec"] = make_featvec(f["desc"])
    n, m12 = oracle.search_by_bow_kf_kf(kf, f, 0.8)
    assert n == (m12 >= 0).sum() and len(set(m12[m12 >= 0])) == n        # vbMatched2: a KF2 feature is used once


def _epipolar_setup(rng, kf2):
    F12 = rng.normal(0, 1e-3, (3, 3)).astype(np.float32); F12[2, 2] = 1e-1
    sf = (np.float32(1.2) ** np.arange(8)).astype(np.float32)
    kf2["scale_factors"] = sf; kf2["level_sigma2"] = (sf * sf * np.float32(400.0)).astype(np.float32)
    return F12, (300.0, 200.0)


def test_oracle_triangulation_properties():
    rng = np.random.default_rng(7)
    kf1 = random_kf(rng, 400, p_mp=0.3); kf2 = perturbed_copy(rng, kf1, nflip=8)
    kf1["feat_vec"] = make_featvec(kf1["desc"]); kf2["feat_vec"] = make_featvec(kf2["desc"])
    F12, epi = _epipolar_setup(rng, kf2)
    for only_stereo in (False, True):
        n, m12 = oracle.search_for_triangulation(kf1, kf2, F12, epi, only_stereo)
        idx = np.nonzero(m12 >= 0)[0]
        assert n == len(idx) and len(set(m12[idx])) == n
        assert not kf1["has_map_point"][idx].any() and not kf2["has_map_point"][m12[idx]].any()
        if only_stereo:
            assert (kf1["u_right"][idx] >= 0).all() and (kf2["u_right"][m12[idx]] >= 0).all()
    n_strict, _ = oracle.search_for_triangulation(kf1, kf2, F12, epi, False, True, oracle.FP_STRICT)
    assert abs(n_strict - oracle.search_for_triangulation(kf1, kf2, F12, epi, False)[0]) <= 2   # modes agree up to float ties


# ----------------------------------------------------------------------------------------------- GPU parity
@pytest.mark.gpu
@pytest.mark.parametrize("ratio,ori", [(0.7, True), (0.9, False)])
def test_gpu_bow_policies_equal_oracle(ratio, ori):
    from orb_slam2_detailed_comments_amd import ORBextractor, ORBmatcher, synth, _capi
    frames = synth.stream(640, 480, 2, stream_id=31)
    ex = ORBextractor(1000, max_batch=2)
    (k1, d1), (k2, d2) = ex.extract_batch(frames)
    rng = np.random.default_rng(3)
    sf = ex.GetScaleFactors()
    mk = lambda k, d: dict(keys_un=k, desc=d, has_map_point=(rng.uniform(size=len(k)) < 0.6).astype(np.uint8),
                           u_right=np.where(rng.uniform(size=len(k)) < 0.4, k["x"] - 5, -1).astype(np.float32),
                           feat_vec=make_featvec(d, bits=4, shuffle_rng=rng), scale_factors=sf,
                           level_sigma2=(sf * sf).astype(np.float32))
    kf1, kf2 = mk(k1, d1), mk(k2, d2)
    m = ORBmatcher(ratio, ori, extractor=ex)
    # SearchByBoW(KF, F)
    n, out = m.SearchByBoW(kf1, k2, d2, kf2["feat_vec"])
    on, oout = oracle.search_by_bow_kf_frame(kf1, k2, d2, kf2["feat_vec"], ratio, ori)
    assert n == on and np.array_equal(out, oout) and n > 30
    # SearchByBoW(KF, KF)
    n, m12 = m.SearchByBoWKeyFrames(kf1, kf2)
    on, om12 = oracle.search_by_bow_kf_kf(kf1, kf2, ratio, ori)
    assert n == on and np.array_equal(m12, om12) and n > 10
    # SearchForTriangulation: the stream translates by (3, 2) px per frame -> pure-translation fundamental matrix
    tx, ty = 3.0, 2.0
    F12 = np.array([[0, 0, ty], [0, 0, -tx], [-ty, tx, 0]], np.float32)
    for only_stereo in (False, True):
        for fp in (_capi.FP_GCC_FMA, _capi.FP_STRICT):
            exm = ORBextractor(1000, fp_mode=fp)
            mm = ORBmatcher(ratio, ori, extractor=exm)
            n, m12 = mm.SearchForTriangulation(kf1, kf2, F12, (-1000.0, -700.0), only_stereo)
            on, om12 = oracle.search_for_triangulation(kf1, kf2, F12, (-1000.0, -700.0), only_stereo, ori, fp)
            assert n == on and np.array_equal(m12, om12)
            if not only_stereo:
                assert n > 5


@pytest.mark.gpu
@pytest.mark.parametrize("fp", [0, 1])
def test_gpu_triangulation_batch_equals_the_loop_of_single_calls(fp):
    """orbx_triangulation_batch_*: the CreateNewMapPoints loop (src/LocalMapping.cc:375-430) -- SearchForTriangulation of the current
    keyframe against each neighbour, MapPoints created for (some of) the matches before the next neighbour is searched -- with the
    distances of all neighbours from ONE device call.  Neighbour by neighbour the same matches as the single call and as the
    oracle, with kf1's has_map_point flags fed back between the searches exactly as the loop does."""
    from orb_slam2_detailed_comments_amd import ORBextractor, ORBmatcher, OrbxError, synth
    frames = synth.stream(640, 480, 4, stream_id=33)
    ex = ORBextractor(1000, max_batch=4, fp_mode=fp)
    res = ex.extract_batch(frames)
    rng = np.random.default_rng(5)
    sf = ex.GetScaleFactors()
    mk = lambda k, d, p: dict(keys_un=k, desc=d, has_map_point=(rng.uniform(size=len(k)) < p).astype(np.uint8),
                              u_right=np.where(rng.uniform(size=len(k)) < 0.4, k["x"] - 5, -1).astype(np.float32),
                              feat_vec=make_featvec(d, bits=4, shuffle_rng=rng), scale_factors=sf, level_sigma2=(sf * sf).astype(np.float32))
    kf1 = mk(*res[0], 0.5)
    nbrs = [mk(*res[1], 0.5), mk(res[2][0][:400], res[2][1][:400], 0.3), mk(*res[3], 0.6), mk(res[1][0][:0], res[1][1][:0], 0.5)]
    Fs = [np.array([[0, 0, 2.0 * (i + 1)], [0, 0, -3.0 * (i + 1)], [-2.0 * (i + 1), 3.0 * (i + 1), 0]], np.float32) for i in range(len(nbrs))]
    epi = (-1000.0, -700.0)
    m = ORBmatcher(0.6, True, extractor=ex)
    total = 0
    with m.TriangulationBatch(kf1, nbrs) as tb:
        cur = dict(kf1, has_map_point=kf1["has_map_point"].copy())
        for k, (kf2, F12) in enumerate(zip(nbrs, Fs)):
            n, m12 = tb.select(k, cur, kf2, F12, epi, False)
            n1, s12 = m.SearchForTriangulation(cur, kf2, F12, epi, False)
            assert n == n1 and np.array_equal(m12, s12), f"neighbour {k}: batch differs from the single call"
            if len(kf2["keys_un"]):
                on, om12 = oracle.search_for_triangulation(cur, kf2, F12, epi, False, True, fp)
                assert n == on and np.array_equal(m12, om12), f"neighbour {k}: batch differs from the oracle"
            # the loop triangulates the matches and gives (most of) them MapPoints: the next search must not see those features
            got = np.nonzero(m12 >= 0)[0]
            made = got[rng.uniform(size=len(got)) < 0.8]
            cur = dict(cur, has_map_point=cur["has_map_point"].copy()); cur["has_map_point"][made] = 1
            total += n
        with pytest.raises(OrbxError):
            tb.select(1, cur, nbrs[0], Fs[0], epi)          # the views of a selection describe the keyframes of the batch
    assert total > 10


@pytest.mark.gpu
def test_gpu_bow_policies_empty_and_malformed():
    from orb_slam2_detailed_comments_amd import ORBextractor, ORBmatcher, OrbxError
    rng = np.random.default_rng(9)
    kf = random_kf(rng, 40); kf["feat_vec"] = make_featvec(kf["desc"])
    m = ORBmatcher(0.7, True, extractor=ORBextractor(500))
    n, out = m.SearchByBoW(kf, kf["keys_un"][:0], kf["desc"][:0], {})
    assert n == 0 and len(out) == 0
    bad = dict(kf, feat_vec=(np.array([5, 3], np.uint32), np.array([0, 1, 2], np.int32), np.array([0, 1], np.uint32)))  # not ascending
    with pytest.raises(OrbxError):
        m.SearchByBoWKeyFrames(bad, kf)
    oob = dict(kf, feat_vec={1: [0, 99]})                                                                                 # index >= n
    with pytest.raises(OrbxError):
        m.SearchByBoWKeyFrames(kf, oob)
