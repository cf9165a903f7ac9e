"""Known-answer tests of the CPU oracle against the only numeric facts the reference itself holds
(SURVEY.md section 4 / 7.4): umax table quoted at src/ORBextractor.cc:912-913, vmax = vmin = 11 (:879-883),
matcher thresholds (src/ORBmatcher.cc:49-51), and the per-level quotas / level sizes that follow from
:820-845 and :2097-2107."""
import numpy as np
import oracle

UMAX = [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]


def test_umax_table():
    t = oracle.OracleExtractor().tables()
    assert t["umax"].tolist() == UMAX
    # the disc defined by umax has 749 pixels
    assert sum(2 * u + 1 for u in UMAX) * 2 - (2 * UMAX[0] + 1) == 749


def test_features_per_level():
    exp = {1000: [217, 181, 151, 126, 105, 87, 73, 60],
           2000: [434, 362, 302, 251, 209, 175, 145, 122],
           1200: [261, 217, 181, 151, 126, 105, 87, 72],
           4000: [869, 724, 603, 503, 419, 349, 291, 242]}
    for n, q in exp.items():
        t = oracle.OracleExtractor(n).tables()
        assert t["features_per_level"].tolist() == q
        assert sum(q) == n


def test_scale_tables():
    t = oracle.OracleExtractor().tables()
    exp = [1, 1.2000000477, 1.4400000572, 1.7280001640, 2.0736002922, 2.4883203506, 2.9859845638, 3.5831816196]
    assert np.allclose(t["scale"], exp, rtol=0, atol=1e-9 * 10)
    assert np.array_equal(t["inv_scale"], np.float32(1) / t["scale"])
    assert np.array_equal(t["sigma2"], t["scale"] * t["scale"])
    # scaledPatchSize = (int)(31 * scale)   (src/ORBextractor.cc:1573)
    assert [int(np.float32(31) * s) for s in t["scale"]] == [31, 37, 44, 53, 64, 77, 92, 111]


def test_level_sizes_fork_padded():
    """Appendix D: un-padded sizes per level, padded = +38 (fork semantics, F1)"""
    exp = {(640, 480): [(640, 480), (533, 400), (444, 333), (370, 278), (309, 231), (257, 193), (214, 161), (179, 134)],
           (1241, 376): [(1241, 376), (1034, 313), (862, 261), (718, 218), (598, 181), (499, 151), (416, 126), (346, 105)],
           (752, 480): [(752, 480), (627, 400), (522, 333), (435, 278), (363, 231), (302, 193), (252, 161), (210, 134)]}
    for (w, h), sizes in exp.items():
        o = oracle.OracleExtractor(100)
        img = np.full((h, w), 77, np.uint8)
        n, _, _ = o.extract(img)
        assert n == 0  # flat image: no corners at any threshold
        P = 0
        for l, (sw, sh) in enumerate(sizes):
            im = o.level_image(l)
            assert im.shape == (sh + 38, sw + 38)
            assert (im == 77).all()
            P += im.size
        if (w, h) == (640, 480):
            assert P == 1158012
        if (w, h) == (1241, 376):
            assert P == 1738559
        if (w, h) == (752, 480):
            assert P == 1344493


def test_three_maxima_and_thresholds():
    assert oracle.three_maxima([0] * 30) == (-1, -1, -1)
    h = [0] * 30
    h[3], h[7], h[20] = 50, 40, 4
    assert oracle.three_maxima(h) == (3, 7, -1)      # third < 10 % of the first
    h[20] = 5
    assert oracle.three_maxima(h) == (3, 7, 20)
    h[7] = 4
    assert oracle.three_maxima(h) == (3, 20, -1)     # bins are ranked by size: (50, 5, 4) -> third dropped
    h[20] = 4
    assert oracle.three_maxima(h) == (3, -1, -1)     # second < 10 % => both dropped
