"""Host-side policy pieces that need no GPU: the 64x48 grid through the C ABI equals the oracle's restatement of
Frame::AssignFeaturesToGrid / GetFeaturesInArea (src/Frame.cc:432-460, 633-745), ComputeThreeMaxima equals
src/ORBmatcher.cc:2026-2068."""
import ctypes as C
import numpy as np
import oracle
from orb_slam2_detailed_comments_amd import Frame, ORBmatcher, _capi


def _random_keys(rng, n, w, h):
    k = np.zeros(n, _capi.KP_DTYPE)
    k["x"] = rng.uniform(-5, w + 25, n).astype(np.float32)
    k["y"] = rng.uniform(-5, h + 25, n).astype(np.float32)
    k["octave"] = rng.integers(0, 8, n)
    return k


def test_grid_queries_equal_oracle(built_lib):
    rng = np.random.default_rng(0)
    for w, h in ((640, 480), (1241, 376)):
        keys = _random_keys(rng, 1500, w, h)
        F = Frame(keys, np.zeros((len(keys), 32), np.uint8), w, h)
        for _ in range(200):
            x, y = float(rng.uniform(-50, w + 50)), float(rng.uniform(-50, h + 50))
            r = float(rng.choice([3.0, 15.0, 100.0, 2000.0]))
            lo, hi = [(-1, -1), (0, 0), (2, 4), (3, -1), (0, 7), (-1, 2)][rng.integers(0, 6)]
            got = F.GetFeaturesInArea(x, y, r, lo, hi)
            exp = oracle.grid_query(keys, (0, w, 0, h), x, y, r, lo, hi)
            assert np.array_equal(got, exp)           # same members AND same (cell-walk) order


def test_three_maxima_equals_oracle(built_lib):
    rng = np.random.default_rng(1)
    L = _capi.lib()
    for _ in range(300):
        sizes = rng.integers(0, rng.integers(1, 60), 30).astype(np.int32)
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        L.orbx_three_maxima(_capi.ptr(sizes), 30, C.byref(a), C.byref(b), C.byref(c))
        assert (a.value, b.value, c.value) == oracle.three_maxima(sizes) == ORBmatcher.ComputeThreeMaxima(sizes)
