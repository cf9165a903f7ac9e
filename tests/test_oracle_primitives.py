"""Properties of the oracle's OpenCV-boundary primitives.  The reference holds no
fixtures for them (parity unpinned); these tests pin the restatement to the published definitions."""
import numpy as np
import pytest
import oracle

RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1),
        (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def fast_reference(img, t):
    """slow, definition-level FAST-9/16 + score + strict 3x3 NMS"""
    h, w = img.shape
    score = np.zeros((h, w), np.int32)
    im = img.astype(np.int32)
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            v = im[y, x]
            d = [v - im[y + dy, x + dx] for dx, dy in RING]
            best = 0
            for sign in (1, -1):
                for k in range(16):
                    m = min(sign * d[(k + j) % 16] for j in range(9))
                    best = max(best, m)
            if best > t:
                score[y, x] = best - 1
    out = []
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            s = score[y, x]
            if s > 0 and all(s > score[y + dy, x + dx] for dy in (-1, 0, 1) for dx in (-1, 0, 1) if dx or dy):
                out.append((x, y, s))
    return out


@pytest.mark.parametrize("seed,t", [(0, 20), (1, 7), (2, 12), (3, 40)])
def test_fast_matches_definition(seed, t):
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (24, 29)).astype(np.uint8)
    img[8:16, 10:20] = rng.integers(0, 40)      # a dark block makes real corners
    k = oracle.fast9_16(img, t)
    got = sorted(zip(k["x"].astype(int), k["y"].astype(int), k["response"].astype(int)))
    assert got == sorted(fast_reference(img, t))
    # emission order is row-major
    order = list(zip(k["y"].astype(int), k["x"].astype(int)))
    assert order == sorted(order)
    assert (k["size"] == 7).all() and (k["angle"] == -1).all() and (k["class_id"] == -1).all()


def test_fast_small_and_flat():
    assert len(oracle.fast9_16(np.zeros((6, 50), np.uint8), 10)) == 0      # a dimension < 7 yields nothing
    assert len(oracle.fast9_16(np.full((40, 40), 200, np.uint8), 1)) == 0


def test_border_reflect101():
    img = np.arange(5 * 7, dtype=np.uint8).reshape(5, 7)
    out = oracle.border101(img, 3)
    assert np.array_equal(out, np.pad(img, 3, mode="reflect"))             # numpy 'reflect' == BORDER_REFLECT_101


def test_resize_constant_and_monotone():
    for c in (0, 1, 77, 255):
        img = np.full((60, 80), c, np.uint8)
        assert (oracle.resize_linear(img, 67, 50) == c).all()              # weights sum to 2048 exactly
    ramp = np.tile(np.arange(0, 240, 2, dtype=np.uint8), (30, 1))
    out = oracle.resize_linear(ramp, 100, 25)
    assert (np.diff(out.astype(int), axis=1) >= 0).all()
    # identical rows stay within 1 LSB (b0 and b1 products are truncated separately)
    assert np.abs(out.astype(int) - out[0].astype(int)).max() <= 1


def test_resize_bilinear_formula():
    """spot-check against an independent evaluation of the fixed-point formula"""
    rng = np.random.default_rng(5)
    src = rng.integers(0, 256, (37, 53)).astype(np.uint8)
    dw, dh = 44, 31
    out = oracle.resize_linear(src, dw, dh)
    sx_scale = 1.0 / (dw / src.shape[1]); sy_scale = 1.0 / (dh / src.shape[0])
    for (dx, dy) in [(0, 0), (43, 30), (10, 7), (21, 15), (43, 0), (0, 30)]:
        fx = np.float32((dx + 0.5) * sx_scale - 0.5); sx = int(np.floor(fx)); fx = np.float32(fx - sx)
        fy = np.float32((dy + 0.5) * sy_scale - 0.5); sy = int(np.floor(fy)); fy = np.float32(fy - sy)
        if sx < 0: sx, fx = 0, np.float32(0)
        if sx >= src.shape[1] - 1: sx, fx = src.shape[1] - 1, np.float32(0)
        a0 = int(np.rint(np.float32(1 - fx) * 2048)); a1 = int(np.rint(fx * np.float32(2048)))
        b0 = int(np.rint(np.float32(1 - fy) * 2048)); b1 = int(np.rint(fy * np.float32(2048)))
        sx1 = min(sx + 1, src.shape[1] - 1)
        r0 = min(max(sy, 0), src.shape[0] - 1); r1 = min(max(sy + 1, 0), src.shape[0] - 1)
        T0 = int(src[r0, sx]) * a0 + int(src[r0, sx1]) * a1
        T1 = int(src[r1, sx]) * a0 + int(src[r1, sx1]) * a1
        exp = (((b0 * (T0 >> 4)) >> 16) + ((b1 * (T1 >> 4)) >> 16) + 2) >> 2
        assert out[dy, dx] == exp


def test_gaussian_blur_kernel_and_paths():
    # the 8-bit kernel {18,34,49,55,49,34,18} sums to 257 (not renormalised): an impulse shows it
    img = np.zeros((21, 24), np.uint8)
    img[10, 10] = 255
    out = oracle.gaussian_blur7(img).astype(int)
    k = np.array([18, 34, 49, 55, 49, 34, 18])
    exp = (np.outer(k, k) * 255 + 32768) >> 16
    assert np.array_equal(out[7:14, 7:14], exp)
    # constant image: 257*257/65536 gain (then saturation)
    for c, e in ((0, 0), (100, 101), (128, 129), (255, 255)):
        assert (oracle.gaussian_blur7(np.full((16, 19), c, np.uint8)) == e).all()
    # float (x < w&~3) and integer tail paths agree except for rounding ties: never more than 1 LSB apart
    rng = np.random.default_rng(2)
    a = rng.integers(0, 256, (40, 43)).astype(np.uint8)
    b = np.pad(a, ((0, 0), (0, 1)), mode="reflect")          # width 44: every column of `a` is on the float path
    oa, ob = oracle.gaussian_blur7(a).astype(int), oracle.gaussian_blur7(b).astype(int)
    assert np.abs(oa[:, :36] - ob[:, :36]).max() <= 1


def test_fast_atan2_and_round():
    L = oracle.lib()
    assert L.orc_fast_atan2(0.0, 0.0) == 0.0
    for y, x in [(1, 1), (1, -1), (-1, -1), (-1, 1), (0, 5), (5, 0), (-3, 0), (0, -2), (123, -457)]:
        a = L.orc_fast_atan2(float(y), float(x))
        assert 0 <= a < 360
        assert abs(a - (np.degrees(np.arctan2(y, x)) % 360)) < 0.02    # documented accuracy ~0.3 deg worst case
    assert [L.orc_cv_round_f(v) for v in (0.5, 1.5, 2.5, -0.5, -1.5, 2.4999, 2.5001)] == [0, 2, 2, 0, -2, 2, 3]


def test_descriptor_distance_is_popcount():
    rng = np.random.default_rng(0)
    for _ in range(200):
        a = rng.integers(0, 256, 32).astype(np.uint8); b = rng.integers(0, 256, 32).astype(np.uint8)
        assert oracle.descriptor_distance(a, b) == int(np.unpackbits(a ^ b).sum())
    z = np.zeros(32, np.uint8); f = np.full(32, 255, np.uint8)
    assert oracle.descriptor_distance(z, z) == 0 and oracle.descriptor_distance(z, f) == 256


def test_match_bruteforce_bookkeeping():
    rng = np.random.default_rng(1)
    q = rng.integers(0, 256, (37, 32)).astype(np.uint8)
    t = rng.integers(0, 256, (53, 32)).astype(np.uint8)
    t[10] = t[40] = q[5]                       # exact tie: the first (lowest index) wins, second == best
    bi, bd, sd = oracle.match_bruteforce(q, t)
    D = np.unpackbits(q[:, None, :] ^ t[None, :, :], axis=2).sum(axis=2)
    assert np.array_equal(bi, D.argmin(axis=1))
    assert np.array_equal(bd, D.min(axis=1))
    assert np.array_equal(sd, np.sort(D, axis=1)[:, 1])
    assert bi[5] == 10 and bd[5] == 0 and sd[5] == 0
    bi, bd, sd = oracle.match_bruteforce(q, t[:0])
    assert (bi == -1).all() and (bd == 0x7fffffff).all()


def test_cvt_gray_formula():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (7, 9, 3)).astype(np.uint8)
    r, g, b = (img[..., i].astype(np.int64) for i in range(3))
    exp = ((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14).astype(np.uint8)
    assert np.array_equal(oracle.cvt_gray(img, True), exp)
    assert np.array_equal(oracle.cvt_gray(img[..., ::-1], False), exp)
    white = np.full((2, 2, 4), 255, np.uint8)
    assert (oracle.cvt_gray(white, True) == 255).all()          # the three weights sum to 16384
