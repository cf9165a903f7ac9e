#!/usr/bin/env python3
"""Randomised parity soak of the matching side: ComputeStereoMatches on random stereo geometries, brute-force best / second
best and the full Hamming matrix on random descriptor sets (sizes 0 .. 3000, duplicates and near-duplicates included), HIP
path against the CPU oracle, bit for bit.  tools/soak_policies.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from orb_slam2_detailed_comments_amd import ORBextractor, ORBmatcher, Frame, OrbxError, synth, _capi

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time(); ns = nm = 0; skipped = 0
m = ORBmatcher(0.6, True)
while time.time() - t0 < budget:
    # ---- stereo
    w, h = int(rng.integers(160, 900)), int(rng.integers(120, 600))
    nf = int(rng.integers(100, 2500)); nl = int(rng.integers(3, 9)); sf = float(rng.choice([1.2, 1.2, 1.3, 1.5]))
    mb = float(rng.uniform(0.05, 0.6)); mbf = float(rng.uniform(10.0, 400.0))
    L, R = synth.stereo_pair(w, h, stream_id=int(rng.integers(0, 1 << 20)), t=int(rng.integers(0, 30)))
    tag = f"stereo {w}x{h} nf={nf} sf={sf} nl={nl} mb={mb:.3f} mbf={mbf:.2f}"
    try:
        exL, exR = ORBextractor(nf, sf, nl), ORBextractor(nf, sf, nl)
        kL, dL = exL(L); kR, dR = exR(R)
    except OrbxError as e:
        if e.status in (_capi.BAD_ASPECT, _capi.UNSUPPORTED): skipped += 1; continue
        raise
    FL, FR = Frame(kL, dL, w, h), Frame(kR, dR, w, h)
    n = FL.ComputeStereoMatches(FR, exL, exR, mb, mbf)
    pyrL = [exL.pyramid_level(l) for l in range(nl)]; pyrR = [exR.pyramid_level(l) for l in range(nl)]
    on, ou, od = oracle.stereo_matches(kL, dL, kR, dR, exL.GetScaleFactors(), exL.GetInverseScaleFactors(), pyrL, pyrR, mb, mbf)
    assert n == on, f"count {n} vs {on}: {tag}"
    assert np.array_equal(FL.mvuRight.view(np.uint32), ou.view(np.uint32)), f"uRight: {tag}"
    assert np.array_equal(FL.mvDepth.view(np.uint32), od.view(np.uint32)), f"depth: {tag}"
    ns += 1
    # ---- brute-force match / Hamming matrix on random descriptor sets (real descriptors mixed with noise and duplicates)
    for _ in range(3):
        nq, nt = int(rng.integers(0, 3000)), int(rng.integers(0, 3000))
        pool = np.concatenate([dL, dR, rng.integers(0, 256, (64, 32), dtype=np.uint8)])
        q = pool[rng.integers(0, len(pool), nq)].copy(); t = pool[rng.integers(0, len(pool), nt)].copy()
        if nq: q[rng.integers(0, nq, nq // 3), rng.integers(0, 32, nq // 3)] ^= np.uint8(1 << int(rng.integers(0, 8)))   # near-duplicates
        bi, bd, sd = m.match_bruteforce(q, t)
        obi, obd, osd = oracle.match_bruteforce(q, t)
        assert np.array_equal(bi, obi) and np.array_equal(bd, obd) and np.array_equal(sd, osd), f"match nq={nq} nt={nt} after {tag}"
        if nq * nt <= 400000:
            D = m.distance_matrix(q, t)
            ref = np.unpackbits(q[:, None, :] ^ t[None, :, :], axis=2).sum(2).astype(np.uint16) if nq and nt else np.zeros((nq, nt), np.uint16)
            assert np.array_equal(D, ref), f"matrix nq={nq} nt={nt}"
        nm += 1
    # ---- device grid + gated candidate lists (the primitive behind every projection-guided policy) against the oracle's
    # GetFeaturesInArea + DescriptorDistance: random bounds (keypoints partly outside), radii, level-argument forms
    import ctypes as C
    kk = kL.copy()
    nk = len(kk)
    if nk:
        bx0, by0 = float(rng.uniform(-20, 60)), float(rng.uniform(-20, 60))
        bounds = np.array([bx0, bx0 + float(rng.uniform(0.5, 1.2)) * w, by0, by0 + float(rng.uniform(0.5, 1.2)) * h], np.float32)
        nq = int(rng.integers(1, 400))
        xyr = np.zeros((nq, 3), np.float32); lv = np.zeros((nq, 2), np.int32)
        xyr[:, 0] = rng.uniform(-50, w + 50, nq); xyr[:, 1] = rng.uniform(-50, h + 50, nq)
        xyr[:, 2] = rng.choice([-1.0, 0.5, 4.0, 9.9, 25.0, 80.0, 300.0], nq)
        lv[:, 0] = rng.integers(-1, nl + 1, nq); lv[:, 1] = rng.integers(-1, nl + 1, nq)
        qd = dR[rng.integers(0, max(len(dR), 1), nq)].copy() if len(dR) else np.zeros((nq, 32), np.uint8)
        begin = np.zeros(nq + 1, np.uint32); items = np.zeros(max(nq * nk, 1), np.uint32); tot = C.c_int(0)
        _capi.check(_capi.lib().orbx_gated_candidates(exL.handle, _capi.ptr(kk), _capi.ptr(dL), nk, _capi.ptr(bounds), _capi.ptr(xyr),
                                                      _capi.ptr(lv), _capi.ptr(qd), nq, _capi.ptr(begin), _capi.ptr(items), len(items), C.byref(tot)))
        for i in range(nq):
            got = items[begin[i]:begin[i + 1]]
            want = oracle.grid_query(kk, tuple(bounds), float(xyr[i, 0]), float(xyr[i, 1]), float(xyr[i, 2]), int(lv[i, 0]), int(lv[i, 1])) if xyr[i, 2] >= 0 else np.zeros(0, np.int32)
            assert np.array_equal(got & 0xffff, want.astype(np.uint32)), f"gate query {i} ({xyr[i]}, {lv[i]}) bounds {bounds}: {tag}"
            if len(want):
                j = int(rng.integers(0, len(want)))
                assert int(got[j] >> 16) == oracle.descriptor_distance(qd[i], dL[want[j]]), f"gate distance: {tag}"
        ng = globals().get("ng", 0) + nq
    if ns % 20 == 0: print(f"  .. {ns} stereo pairs, {nm} descriptor sets, {time.time() - t0:.0f} s", flush=True)
print(f"gated candidate queries compared: {globals().get('ng', 0)}")
print(f"policy soak ok: {ns} random stereo pairs ({skipped} unsupported geometries skipped), {nm} random descriptor-set matches in {time.time() - t0:.0f} s")
