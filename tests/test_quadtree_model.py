"""The level-synchronous quadtree formulation used by the HIP kernel (tests/qt_model.py mirrors k_quadtree
step by step) must reproduce the literal std::list restatement of DistributeOctTree in the oracle
(reference src/ORBextractor.cc:1050-1417) -- list order, careful-phase cut and selection included."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st
import oracle
from qt_model import distribute


def _keys(pts, resp):
    k = np.zeros(len(pts), oracle.KP_DTYPE)
    if len(pts):
        k["x"] = [p[0] for p in pts]; k["y"] = [p[1] for p in pts]
    k["response"] = resp
    return k


def _random_case(rng, clustered):
    qt_w = int(rng.integers(40, 700)); qt_h = int(rng.integers(40, 500))
    while round(qt_w / qt_h) == 0:
        qt_h = int(rng.integers(40, 500))
    K = int(rng.integers(0, 900)); N = int(rng.integers(1, 300))
    pts = set()
    tries = 0
    while len(pts) < K and tries < 20 * K + 100:
        tries += 1
        if clustered:
            cx, cy = rng.integers(3, qt_w - 3), rng.integers(3, qt_h - 3)
            x = int(np.clip(cx + rng.normal(0, 6), 3, qt_w - 4)); y = int(np.clip(cy + rng.normal(0, 6), 3, qt_h - 4))
        else:
            x = int(rng.integers(3, qt_w - 3)); y = int(rng.integers(3, qt_h - 3))
        pts.add((x, y))
    pts = list(pts)
    rng.shuffle(pts)
    return qt_w, qt_h, N, pts, rng.integers(7, 30, len(pts))


@pytest.mark.parametrize("seed", range(40))
def test_model_equals_literal_list(seed):
    rng = np.random.default_rng(seed)
    qt_w, qt_h, N, pts, resp = _random_case(rng, clustered=bool(seed & 1))
    keys = _keys(pts, resp)
    n, idx = oracle.distribute_octtree(keys, 16, 16 + qt_w, 16, 16 + qt_h, N)
    sel = distribute(keys["x"].astype(int), keys["y"].astype(int), keys["response"], np.arange(len(pts)), qt_w, qt_h, N)
    assert n == len(sel)
    assert idx.tolist() == sel


@settings(max_examples=60, deadline=None)
@given(st.integers(0, 2 ** 31), st.integers(1, 120))
def test_quadtree_invariants(seed, N):
    rng = np.random.default_rng(seed)
    qt_w, qt_h = int(rng.integers(60, 400)), int(rng.integers(60, 300))
    if round(qt_w / qt_h) == 0:
        qt_w = qt_h
    K = int(rng.integers(0, 400))
    pts = list({(int(rng.integers(3, qt_w - 3)), int(rng.integers(3, qt_h - 3))) for _ in range(K)})
    keys = _keys(pts, rng.integers(7, 60, len(pts)))
    n, idx = oracle.distribute_octtree(keys, 16, 16 + qt_w, 16, 16 + qt_h, N)
    nini = int(np.floor(qt_w / qt_h + 0.5))
    assert 0 <= n <= max(N + 3, 4 * nini)          # the quota can be overshot by at most 3
    assert n <= len(pts)
    assert len(set(idx.tolist())) == n              # one keypoint per node, no duplicates
    # note: n may stay below min(N, #points): a pass that only moves a 2-key node into one child leaves the
    # list size unchanged and the reference stops there (lNodes.size() == prevSize, :1260-1268)


def test_bad_aspect_is_reported():
    keys = _keys([(5, 5)], [10])
    n, _ = oracle.distribute_octtree(keys, 16, 16 + 40, 16, 16 + 100, 10)   # w/h = 0.4 -> nIni = 0
    assert n == -3
