#!/usr/bin/env python3
"""Randomised parity soak: extractor geometries / parameters / image statistics drawn at random, HIP path against the CPU
oracle, bit for bit (keypoints, descriptors, and every intermediate stage for a subset).  tools/soak.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from orb_slam2_detailed_comments_amd import ORBextractor, OrbxError, synth, _capi

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time(); n = 0; nkp = 0; skipped = 0
while time.time() - t0 < budget:
    w, h = int(rng.integers(48, 1000)), int(rng.integers(48, 700))
    nf = int(rng.integers(50, 3000)); sf = float(rng.choice([1.1, 1.2, 1.2, 1.2, 1.3, 1.5, 2.0])); nl = int(rng.integers(2, 10))
    ini = int(rng.integers(8, 40)); mn = int(rng.integers(2, ini + 1)); fp = int(rng.integers(0, 2))
    kind = rng.integers(0, 5)
    if kind == 0: img = synth.Scene(w, h, int(rng.integers(0, 1 << 20))).frame(int(rng.integers(0, 50)))
    elif kind == 1: img = rng.integers(0, 256, (h, w), dtype=np.uint8)                                   # dense corners
    elif kind == 2: img = (synth.Scene(w, h, int(rng.integers(0, 1 << 20))).frame(0).astype(np.int32) // 32 * 32).astype(np.uint8)   # plateaus: score ties
    elif kind == 4:   # diagonal stripes with mid-gray lines: candidates that pass both pre-tests of the one-pass corner test (re-run stack)
        per, amp, nz = int(rng.integers(6, 24)), int(rng.integers(30, 120)), int(rng.integers(0, 30))
        yy, xx = np.mgrid[0:h, 0:w]; ph = (xx + yy * int(rng.choice([-1, 1]))) % per
        t = np.where(ph == 0, 128, np.where(ph < per // 2, 128 - amp, 128 + amp)) + rng.integers(-nz, nz + 1, (h, w))
        img = np.clip(t, 0, 255).astype(np.uint8)
    else:
        img = np.full((h, w), int(rng.integers(0, 256)), np.uint8)
        for _ in range(int(rng.integers(1, 200))):
            x, y = int(rng.integers(0, w)), int(rng.integers(0, h)); img[y:y + int(rng.integers(1, 9)), x:x + int(rng.integers(1, 9))] = int(rng.integers(0, 256))
    try:
        ex = ORBextractor(nf, sf, nl, ini, mn, fp_mode=fp)
        k, d = ex(img)
    except OrbxError as e:
        if e.status in (_capi.BAD_ASPECT, _capi.UNSUPPORTED): skipped += 1; continue
        raise
    orc = oracle.OracleExtractor(nf, sf, nl, ini, mn, fp_mode=fp)
    on, ok, od = orc.extract(img, cap=ex.max_keypoints(w, h))
    tag = f"{w}x{h} nf={nf} sf={sf} nl={nl} th={ini}/{mn} fp={fp} kind={kind}"
    assert on == len(k), f"count {on} vs {len(k)}: {tag}"
    assert ok.tobytes() == k.tobytes(), f"keypoints differ: {tag}"
    assert np.array_equal(od, d), f"descriptors differ: {tag}"
    if n % 5 == 0:
        for l in range(nl):
            assert np.array_equal(ex.pyramid_level(l), orc.level_image(l)), f"pyramid level {l}: {tag}"
            oc, gc = orc.level_candidates(l), ex.debug_candidates(l, cap=1 << 21)
            assert sorted(zip(oc["x"], oc["y"], oc["response"])) == sorted(zip(gc["x"], gc["y"], gc["response"])), f"candidates level {l}: {tag}"
    n += 1; nkp += on
    if n % 200 == 0: print(f"  .. {n} configurations, {nkp} keypoints, {time.time() - t0:.0f} s", flush=True)
print(f"soak ok: {n} random configurations ({skipped} unsupported geometries skipped), {nkp} keypoints compared bit for bit in {time.time() - t0:.0f} s")
