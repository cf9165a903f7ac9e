"""CPU model of the one-pass FAST corner test of k_fast_rows (csrc/orbx_kernels.hip, fr_round) against the oracle's cv::FAST.

The kernel does not run the reference's ring-mask test (src/ORBextractor.cc:1510-1527 -> cv::FAST) followed by
cv::cornerScore: it evaluates ONE min / max network per candidate -- the score's own -- for ONE polarity, chosen by the larger
compass margin, and re-runs the candidates that pass both compass pre-tests and fail the polarity taken.  This file restates that
scheme in numpy (whole image at once) and checks the identities it rests on, on the CPU, against oracle.fast9_16:
  * corner  <=>  network(+-(ring - v)) > th      for the corner's polarity
  * score   ==   network - 1
  * every corner is found by (polarity with the larger margin) or (both pre-tests pass -> other polarity)
  * strict 3x3 non-maximum suppression over those scores gives the reference's keypoints and responses.
The GPU kernel itself is compared with the oracle in tests/test_gpu_parity.py; this test needs no GPU.
"""
import numpy as np
import pytest

import oracle

RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1),
        (-2, 2), (-1, 3)]   # (dx, dy), ring index k as in fr_round's ro[] table


def _network(d, th):
    """max(th, max over the 16 arcs of the min over the 9 pixels of the arc): m2 -> m4 -> m8 -> + the ninth pixel, as the kernel"""
    m2 = [np.minimum(d[k], d[(k + 1) & 15]) for k in range(16)]
    m4 = [np.minimum(m2[k], m2[(k + 2) & 15]) for k in range(16)]
    a0 = np.full(d[0].shape, th, np.int32)
    for k in range(16):
        a0 = np.maximum(a0, np.minimum(np.minimum(m4[k], m4[(k + 4) & 15]), d[(k + 8) & 15]))
    return a0


def onepass_fast(img, th):
    """(score map before NMS, number of pre-test candidates, number of re-run candidates) by the kernel's scheme"""
    h, w = img.shape
    im = img.astype(np.int32)
    v = im[3:h - 3, 3:w - 3]
    d = [im[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx] - v for dx, dy in RING]
    s1 = np.minimum(np.maximum(d[0], d[8]), np.maximum(d[4], d[12]))     # brighter margin of the compass pixels
    s2 = np.maximum(np.minimum(d[0], d[8]), np.minimum(d[4], d[12]))     # -(darker margin)
    pre = np.maximum(s1, -s2) > th                                        # the walk's pre-test
    sgn = np.where(s1 + s2 < 0, -1, 1)
    r1 = _network([sgn * x for x in d], th)
    c1 = pre & (r1 > th)
    redo = pre & (np.minimum(s1, -s2) > th) & ~c1                         # both pre-tests pass, the polarity taken failed
    r2 = _network([-sgn * x for x in d], th)
    c2 = redo & (r2 > th)
    score = np.zeros((h, w), np.int32)
    score[3:h - 3, 3:w - 3] = np.where(c1, r1 - 1, np.where(c2, r2 - 1, 0))
    return score, int(pre.sum()), int(redo.sum())


def nms3x3(score):
    h, w = score.shape
    keep = score > 0
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            if dx or dy:
                sh = np.zeros_like(score)
                sh[max(0, -dy):h - max(0, dy), max(0, -dx):w - max(0, dx)] = score[max(0, dy):h - max(0, -dy), max(0, dx):w - max(0, -dx)]
                keep &= score > sh
    ys, xs = np.nonzero(keep)
    return [(int(x), int(y), int(score[y, x])) for y, x in zip(ys, xs)]   # row-major, as cv::FAST emits


def _images():
    rng = np.random.default_rng(5)
    yield "noise", rng.integers(0, 256, (90, 120)).astype(np.uint8)
    yield "plateaus", (rng.integers(0, 256, (80, 100)) // 48 * 48).astype(np.uint8)
    yy, xx = np.mgrid[0:96, 0:128]
    ph = (xx + yy) % 12
    st = np.where(ph == 0, 128, np.where(ph < 6, 40, 216)) + rng.integers(-20, 21, (96, 128))
    yield "stripes", np.clip(st, 0, 255).astype(np.uint8)          # many candidates that pass both pre-tests
    sq = np.full((64, 64), 30, np.uint8); sq[20:44, 20:44] = 220
    yield "square", sq
    yield "flat", np.full((32, 40), 99, np.uint8)


@pytest.mark.parametrize("th", [7, 20, 40])
def test_onepass_scheme_equals_cv_fast(th):
    seen_redo = 0
    for name, img in _images():
        score, npre, nredo = onepass_fast(img, th)
        seen_redo += nredo
        ref = oracle.fast9_16(img, th)
        got = nms3x3(score)
        want = [(int(round(float(k["x"]))), int(round(float(k["y"]))), int(k["response"])) for k in ref]
        assert got == want, f"{name} th={th}: {len(got)} vs {len(want)} keypoints"
    assert seen_redo > 0     # the re-run path was exercised (stripes)


def test_polarity_is_unique_and_the_rerun_finds_the_other_one():
    """a corner of the polarity NOT taken is always a re-run candidate: the scheme cannot lose it"""
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (120, 160)).astype(np.uint8)
    h, w = img.shape
    im = img.astype(np.int32)
    v = im[3:h - 3, 3:w - 3]
    d = [im[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx] - v for dx, dy in RING]
    for th in (5, 15, 30):
        bright = _network(d, th) > th
        dark = _network([-x for x in d], th) > th
        assert not (bright & dark).any()                       # 9 of 16 brighter AND 9 of 16 darker cannot both hold
        s1 = np.minimum(np.maximum(d[0], d[8]), np.maximum(d[4], d[12]))
        s2 = np.maximum(np.minimum(d[0], d[8]), np.minimum(d[4], d[12]))
        assert ((s1 > th) | ~bright).all() and ((-s2 > th) | ~dark).all()   # a corner passes its own pre-test
        taken_bright = s1 + s2 >= 0
        lost = (bright & ~taken_bright) | (dark & taken_bright)            # corners whose polarity was not taken first
        assert (np.minimum(s1, -s2)[lost] > th).all()                      # ... all pass BOTH pre-tests: they are re-run
