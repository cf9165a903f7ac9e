"""EuRoC rectification: cv::remap(im, imRect, M1, M2, cv::INTER_LINEAR) of
Examples/Stereo/stereo_euroc.cc:183-194 fused into pyramid level 0.  OpenCV 3.2 arithmetic restated (parity unpinned).
CPU: the oracle's remap against exact cases (identity, half-pixel shift, out-of-range taps).  GPU: extraction of the RAW image
with the maps set equals the oracle extraction of the oracle-rectified image, stage by stage."""
import numpy as np
import pytest
import oracle


def euroc_like_maps(w, h, seed=0):
    """radial distortion + a small rotation, as initUndistortRectifyMap produces them (float32, arbitrary fractions)"""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    cx, cy, f = w / 2 + 3.3, h / 2 - 2.1, 0.6 * w
    x, y = (xx - cx) / f, (yy - cy) / f
    a = 0.01
    xr, yr = x * np.cos(a) - y * np.sin(a), x * np.sin(a) + y * np.cos(a)
    r2 = xr * xr + yr * yr
    k = 1 - 0.28 * r2 + 0.07 * r2 * r2
    return (xr * k * f + cx).astype(np.float32), (yr * k * f + cy).astype(np.float32)


def test_oracle_remap_exact_cases():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (40, 50), dtype=np.uint8)
    yy, xx = np.mgrid[0:40, 0:50].astype(np.float32)
    assert np.array_equal(oracle.remap_linear(img, xx, yy), img)                       # zero fractions: the {32767,0,0,1} entry
    half = oracle.remap_linear(img, xx + 0.5, yy + 0.5)
    exp = (img[:-1, :-1].astype(int) + img[:-1, 1:] + img[1:, :-1] + img[1:, 1:]) * 8192 + 16384 >> 15
    assert np.array_equal(half[:-1, :-1], exp)
    out = oracle.remap_linear(img, xx - 100, yy)                                       # everything outside: border value 0
    assert not out.any()
    edge = oracle.remap_linear(img, xx + 0.25, yy)                                     # last column blends with the border 0
    assert np.array_equal(edge[:, -1], (img[:, -1].astype(int) * 24576 + 16384) >> 15)
    # rounding of the coordinates: 1/64 rounds to even (0), 3/64 rounds to 2/32 -> 1/16 ... cvRound semantics
    assert np.array_equal(oracle.remap_linear(img, xx + 1 / 64, yy), img)


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,nf", [(752, 480, 1200), (320, 240, 400)])
def test_gpu_rectified_extraction_equals_oracle(w, h, nf):
    from orb_slam2_detailed_comments_amd import ORBextractor, OrbxError, synth
    mx, my = euroc_like_maps(w, h)
    frames = synth.stream(w, h, 2, stream_id=91)
    ex = ORBextractor(nf, max_batch=2)
    ex.set_rectification(mx, my)
    res = ex.extract_batch(frames)
    orc = oracle.OracleExtractor(nf)
    for f in range(2):
        rect = oracle.remap_linear(frames[f], mx, my)
        on, ok, od = orc.extract(rect, cap=ex.max_keypoints(w, h))
        assert np.array_equal(ex.pyramid_level(0, f)[19:-19, 19:-19], rect)
        assert np.array_equal(ex.pyramid_level(0, f), orc.level_image(0))
        assert on == len(res[f][0]) and ok.tobytes() == res[f][0].tobytes() and np.array_equal(od, res[f][1])
        assert on > 0.5 * nf
    ex.set_rectification(None, None)                                                    # off again: plain extraction
    k, d = ex(frames[0])
    on, ok, od = orc.extract(frames[0])
    assert ok.tobytes() == k.tobytes()
    ex.set_rectification(mx, my)
    with pytest.raises(OrbxError):
        ex(np.zeros((h + 8, w), np.uint8))                                              # maps of another size
