"""Replay of tools/soak.py's random stream (seed argv[1]) up to the configuration tagged argv[2]; prints where the level images differ."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from orb_slam2_detailed_comments_amd import ORBextractor, OrbxError, synth, _capi
rng = np.random.default_rng(int(sys.argv[1]))
want = sys.argv[2]
n = 0
while True:
    w, h = int(rng.integers(48, 1000)), int(rng.integers(48, 700))
    nf = int(rng.integers(50, 3000)); sf = float(rng.choice([1.1, 1.2, 1.2, 1.2, 1.3, 1.5, 2.0])); nl = int(rng.integers(2, 10))
    ini = int(rng.integers(8, 40)); mn = int(rng.integers(2, ini + 1)); fp = int(rng.integers(0, 2))
    kind = rng.integers(0, 4)
    if kind == 0: img = synth.Scene(w, h, int(rng.integers(0, 1 << 20))).frame(int(rng.integers(0, 50)))
    elif kind == 1: img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    elif kind == 2: img = (synth.Scene(w, h, int(rng.integers(0, 1 << 20))).frame(0).astype(np.int32) // 32 * 32).astype(np.uint8)
    else:
        img = np.full((h, w), int(rng.integers(0, 256)), np.uint8)
        for _ in range(int(rng.integers(1, 200))):
            x, y = int(rng.integers(0, w)), int(rng.integers(0, h)); img[y:y + int(rng.integers(1, 9)), x:x + int(rng.integers(1, 9))] = int(rng.integers(0, 256))
    tag = f"{w}x{h} nf={nf} sf={sf} nl={nl} th={ini}/{mn} fp={fp} kind={kind}"
    try:
        ex = ORBextractor(nf, sf, nl, ini, mn, fp_mode=fp)
        k, d = ex(img)
    except OrbxError as e:
        if e.status in (_capi.BAD_ASPECT, _capi.UNSUPPORTED): continue
        raise
    n += 1
    if tag != want:
        if n > 3000: print("not found"); break
        continue
    print("found at n =", n, tag, "img flags", img.flags['C_CONTIGUOUS'], img.dtype, img.shape)
    orc = oracle.OracleExtractor(nf, sf, nl, ini, mn, fp_mode=fp)
    on, ok, od = orc.extract(img, cap=ex.max_keypoints(w, h))
    print("keypoints equal:", on == len(k) and ok.tobytes() == k.tobytes(), "descriptors equal:", np.array_equal(od, d))
    for rep in range(3):
        k2, d2 = ex(img)
        for l in range(nl):
            a, b = ex.pyramid_level(l), orc.level_image(l)
            if not np.array_equal(a, b):
                ys, xs = np.nonzero(a != b)
                print("rep", rep, "level", l, "shape", a.shape, "ndiff", len(ys), "rows", ys.min(), ys.max(), "cols", xs.min(), xs.max())
                print(" first diffs (y, x, gpu, oracle)", [(int(y), int(x), int(a[y, x]), int(b[y, x])) for y, x in list(zip(ys, xs))[:10]])
            else:
                print("rep", rep, "level", l, "equal")
    break
