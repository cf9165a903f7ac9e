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
    # ---- batched Fuse (orbx_fuse_batch / orbx_fuse_sim3_batch): K random (keyframe, projected points) problems in one call, per
    # problem against the oracle's single Fuse; keyframes = the two eyes' features cut to random sizes (some empty), point sets
    # shared between problems (SearchInNeighbors) or not, both fp modes of the chi2 gate through the extractor's own mode
    if len(kL) and len(kR):
        sfac = exL.GetScaleFactors(); isig = exL.GetInverseScaleSigmaSquares()
        def target(k, d):
            nkeep = int(rng.integers(0, len(k) + 1))
            return dict(keys_un=k[:nkeep], desc=d[:nkeep], bounds=(0.0, float(w + 38), 0.0, float(h + 38)), scale_factors=sfac, inv_level_sigma2=isig,
                        u_right=np.where(rng.uniform(size=nkeep) < 0.4, k["x"][:nkeep] - rng.uniform(2, 30, nkeep), -1).astype(np.float32))
        def points(k, d):
            npt = int(rng.integers(0, 600))
            idx = rng.integers(0, len(k), npt)
            uv = np.stack([k["x"][idx], k["y"][idx]], 1).astype(np.float32) + rng.normal(0, 2.0, (npt, 2)).astype(np.float32)
            dd = d[idx].copy()
            if npt: dd[rng.integers(0, npt, npt // 2), rng.integers(0, 32, npt // 2)] ^= np.uint8(1 << int(rng.integers(0, 8)))
            return dict(valid=(rng.uniform(size=npt) < 0.85).astype(np.uint8), uv=uv, level=np.clip(k["octave"][idx] + rng.integers(0, 2, npt), 0, nl - 1).astype(np.int32),
                        desc=dd, u_right=(uv[:, 0] - rng.uniform(2, 30, npt)).astype(np.float32) if npt else np.zeros(0, np.float32))
        K = int(rng.integers(1, 7))
        tgs = [target(*((kL, dL) if rng.uniform() < 0.5 else (kR, dR))) for _ in range(K)]
        shared = points(kL, dL)
        pts_l = [shared if rng.uniform() < 0.6 else points(kR, dR) for _ in range(K)]
        mm = ORBmatcher(0.6, True, extractor=exL)
        for sim3, th in ((False, float(rng.choice([2.0, 3.0, 5.0]))), (True, float(rng.choice([3.0, 4.0, 8.0])))):
            cnts, bests = mm.FuseBatch(tgs, pts_l, th, sim3=sim3)
            for kq in range(K):
                if len(tgs[kq]["keys_un"]) == 0 or len(pts_l[kq]["valid"]) == 0:
                    assert cnts[kq] == 0 and (bests[kq] == -1).all(), f"empty problem {kq}: {tag}"
                    continue
                on_, ob_ = oracle.fuse_sim3(tgs[kq], pts_l[kq], th) if sim3 else oracle.fuse(tgs[kq], pts_l[kq], th, 0)
                assert cnts[kq] == on_ and np.array_equal(bests[kq], ob_), f"fuse batch problem {kq} of {K} sim3={sim3} th={th}: {tag}"
            globals()["nfb"] = globals().get("nfb", 0) + K
    if ns % 20 == 0: print(f"  .. {ns} stereo pairs, {nm} descriptor sets, {time.time() - t0:.0f} s", flush=True)
print(f"gated candidate queries compared: {globals().get('ng', 0)}; batched Fuse problems compared: {globals().get('nfb', 0)}")
print(f"policy soak ok: {ns} random stereo pairs ({skipped} unsupported geometries skipped), {nm} random descriptor-set matches in {time.time() - t0:.0f} s")
