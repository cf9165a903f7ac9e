#!/usr/bin/env python3
"""us per call of every ORBmatcher / Frame entry point behind the C ABI (host buffers in, host buffers out: what the drop-in
classes of compat/ pay per call), next to the single-thread CPU oracle on the SAME inputs.  Inputs as in the GPU parity tests:
two consecutive 640x480 / 1000-feature frames of the synthetic stream (2000 features for SearchForInitialization, the
reference's mpIniORBextractor, src/Tracking.cc:181-182), MapPoints synthesised from the first frame.

    python tools/policy_rates.py [--json out.json]
"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle
from orb_slam2_detailed_comments_amd import ORBextractor, ORBmatcher, Frame, synth, _capi
from test_projection_policies import make_target, points_from
from test_bow_policies import make_featvec


def bench(fn, reps):
    """median of the per-call wall times (a mean over 30 calls moves by hundreds of us with one scheduling hiccup of the host)"""
    fn()
    ts = []
    for _ in range(reps):
        t = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t)
    return float(np.median(ts)) * 1e6


def main():
    rows = []
    def add(name, ref, gpu_fn, cpu_fn, reps=30, creps=5):
        g, c = bench(gpu_fn, reps), bench(cpu_fn, creps)
        rows.append(dict(entry_point=name, reference=ref, gpu_us=round(g, 1), cpu_oracle_us=round(c, 1), ratio=round(c / g, 2)))
        print(f"{name:40s} {ref:38s} GPU {g:9.1f} us   CPU oracle {c:10.1f} us   x{c / g:6.2f}", flush=True)

    frames = synth.stream(640, 480, 2, stream_id=41)
    ex = ORBextractor(1000, max_batch=2)
    (k1, d1), (k2, d2) = ex.extract_batch(frames)
    rng = np.random.default_rng(11)
    sf = ex.GetScaleFactors()
    m = ORBmatcher(0.7, True, extractor=ex)
    bounds = (0.0, 640.0, 0.0, 480.0)

    # ---- SearchForInitialization (2 x nFeatures extractor)
    ex2 = ORBextractor(2000, max_batch=2)
    (a1, b1), (a2, b2) = ex2.extract_batch(frames)
    F1, F2 = Frame(a1, b1, 640, 480), Frame(a2, b2, 640, 480)
    mi = ORBmatcher(0.9, True, extractor=ex2)
    pm0 = np.stack([a1["x"], a1["y"]], 1).astype(np.float32)
    add("orbx_search_for_initialization", "ORBmatcher.cc:570-712 (window 100)",
        lambda: mi.SearchForInitialization(F1, F2, pm0.copy(), 100),
        lambda: oracle.search_for_initialization(a1, b1, a2, b2, bounds, pm0.copy(), 100, 0.9, True))

    # ---- SearchByProjection(Frame, Frame)
    last, cur = Frame(k1, d1, 640, 480), Frame(k2, d2, 640, 480)
    fx = fy = 500.0; cx, cy = 339.0, 259.0; Z = 2.0
    n0 = len(k1)
    has_mp = (rng.uniform(size=n0) < 0.8).astype(np.uint8); obs = rng.integers(0, 3, n0).astype(np.int32)
    xw = np.stack([(k1["x"] - cx) / fx * Z, (k1["y"] - cy) / fy * Z, np.full(n0, Z)], 1).astype(np.float32)
    Tlw = np.eye(4, dtype=np.float32); Tcw = np.eye(4, dtype=np.float32)
    Tcw[0, 3], Tcw[1, 3] = 3.0 / fx * Z, 2.0 / fy * Z
    add("orbx_search_by_projection_frame", "ORBmatcher.cc:1702-1871 (th 15)",
        lambda: m.SearchByProjection(cur, last, 15.0, True, Tcw=Tcw, Tlw=Tlw, K=(fx, fy, cx, cy), mb=0.1, mbf=40.0,
                                     has_map_point=has_mp, world_pos=xw, mp_desc=d1, observations=obs),
        lambda: oracle.search_by_projection_ff(k2, d2, cur.mvuRight, Tcw, (fx, fy, cx, cy), bounds, 0.1, 40.0, sf, k1, has_mp, xw,
                                               d1, obs, Tlw, 15.0, True))

    # ---- SearchByProjection(Frame, MapPoints)
    nmp = len(k1)
    proj = np.stack([k1["x"] - 3.0 + rng.normal(0, 1.0, nmp), k1["y"] - 2.0 + rng.normal(0, 1.0, nmp), k1["x"] - 23.0], 1).astype(np.float32)
    in_view = (rng.uniform(size=nmp) < 0.85).astype(np.uint8)
    level = np.clip(k1["octave"] + rng.integers(-1, 2, nmp), 0, 7).astype(np.int32)
    view_cos = rng.uniform(0.99, 1.0, nmp).astype(np.float32); mp_obs = rng.integers(0, 4, nmp).astype(np.int32)
    frame_obs = np.where(rng.uniform(size=cur.N) < 0.2, rng.integers(0, 3, cur.N), -1).astype(np.int32)
    add("orbx_search_by_projection_mappoints", "ORBmatcher.cc:69-184 (th 3)",
        lambda: m.SearchByProjectionMapPoints(cur, 3.0, frame_observations=frame_obs, in_view=in_view, proj=proj, level=level,
                                              view_cos=view_cos, mp_desc=d1, observations=mp_obs),
        lambda: oracle.search_by_projection_mp(k2, d2, cur.mvuRight, frame_obs, bounds, sf, in_view, proj, level, view_cos, d1,
                                               mp_obs, 3.0, 0.7))

    # ---- projection-guided back-end policies
    mk = lambda k, d: dict(make_target(k, d, rng, 678, 518), scale_factors=sf, inv_level_sigma2=ex.GetInverseScaleSigmaSquares())
    t1, t2 = mk(k1, d1), mk(k2, d2)
    pts, _ = points_from(k1, d1, rng, shift=(-3.0, -2.0))
    add("orbx_fuse", "ORBmatcher.cc:1100-1280 (th 3)", lambda: m.Fuse(t2, pts, 3.0), lambda: oracle.fuse(t2, pts, 3.0, 0))
    add("orbx_fuse_sim3", "ORBmatcher.cc:1282-1430 (th 4)", lambda: m.FuseSim3(t2, pts, 4.0), lambda: oracle.fuse_sim3(t2, pts, 4.0))
    # the Fuse loops of LocalMapping::SearchInNeighbors / LoopClosing::SearchAndFuse as one batched call: K neighbour keyframes
    # against the current keyframe's points; us PER PROBLEM next to K single CPU calls
    # (timed at the C ABI with the views marshalled once -- what compat/ORBmatcher.h's FuseBatch pays; the Python wrapper's
    # ctypes marshalling of K view structs per call, 10 us per problem, is the test harness's cost, not the library's)
    import ctypes as C
    Lc = _capi.lib()
    for K in (1, 8, 32):
        keep = []
        tvs = [m._target(t2, keep) for _ in range(K)]; pvs = [m._points(pts, keep) for _ in range(K)]
        outs = [np.full(max(pv.n, 1), -1, np.int32) for pv in pvs]
        tarr = (C.c_void_p * K)(*[C.addressof(t) for t in tvs]); parr = (C.c_void_p * K)(*[C.addressof(p_) for p_ in pvs])
        oarr = (C.c_void_p * K)(*[o.ctypes.data for o in outs]); nf = (C.c_int * K)()
        for name, fn, th, orc in (("orbx_fuse_batch", Lc.orbx_fuse_batch, 3.0, lambda: [oracle.fuse(t2, pts, 3.0, 0) for _ in range(K)]),
                                  ("orbx_fuse_sim3_batch", Lc.orbx_fuse_sim3_batch, 4.0, lambda: [oracle.fuse_sim3(t2, pts, 4.0) for _ in range(K)])):
            g, c = bench(lambda: _capi.check(fn(ex.handle, K, tarr, parr, C.c_float(th), oarr, nf)), 30), bench(orc, 3)
            rows.append(dict(entry_point=f"{name} K={K}", reference="LocalMapping.cc:750-768 loop" if "sim3" not in name else "LoopClosing SearchAndFuse loop",
                             gpu_us=round(g / K, 1), cpu_oracle_us=round(c / K, 1), ratio=round(c / g, 2), problems=K))
            print(f"{name + f' K={K}':40s} {'us per problem (C ABI)':38s} GPU {g / K:9.1f} us   CPU oracle {c / K:10.1f} us   x{c / g:6.2f}", flush=True)
    matched = (rng.uniform(size=len(k2)) < 0.2).astype(np.uint8)
    add("orbx_search_by_projection_sim3", "ORBmatcher.cc:415-560 (th 10)",
        lambda: m.SearchByProjectionSim3(t2, pts, matched.copy(), 10), lambda: oracle.search_by_projection_sim3(t2, pts, matched.copy(), 10))
    p12 = dict(valid=(rng.uniform(size=len(k1)) < 0.8).astype(np.uint8), uv=np.stack([k1["x"] - 3, k1["y"] - 2], 1).astype(np.float32),
               level=k1["octave"].astype(np.int32), desc=d1)
    p21 = dict(valid=(rng.uniform(size=len(k2)) < 0.8).astype(np.uint8), uv=np.stack([k2["x"] + 3, k2["y"] + 2], 1).astype(np.float32),
               level=k2["octave"].astype(np.int32), desc=d2)
    add("orbx_search_by_sim3", "ORBmatcher.cc:1433-1690 (th 7.5)",
        lambda: m.SearchBySim3(t1, t2, p12, p21, 7.5), lambda: oracle.search_by_sim3(t1, t2, p12, p21, 7.5))
    has = (rng.uniform(size=len(k2)) < 0.3).astype(np.uint8)
    add("orbx_search_by_projection_keyframe", "ORBmatcher.cc:1873-2020 (th 10)",
        lambda: m.SearchByProjectionKeyFrame(t2, pts, has.copy(), 10.0, 100), lambda: oracle.search_by_projection_kf(t2, pts, has.copy(), 10.0, 100, True))

    # ---- BoW-guided policies (feature vectors from the descriptors' leading bits: 16 nodes, ~60 features per node)
    mkf = lambda k, d: dict(keys_un=k, desc=d, has_map_point=(rng.uniform(size=len(k)) < 0.6).astype(np.uint8),
                            u_right=np.where(rng.uniform(size=len(k)) < 0.4, k["x"] - 5, -1).astype(np.float32),
                            feat_vec=make_featvec(d, bits=4, shuffle_rng=rng), scale_factors=sf, level_sigma2=(sf * sf).astype(np.float32))
    kf1, kf2 = mkf(k1, d1), mkf(k2, d2)
    add("orbx_search_by_bow_keyframe_frame", "ORBmatcher.cc:248-410",
        lambda: m.SearchByBoW(kf1, k2, d2, kf2["feat_vec"]), lambda: oracle.search_by_bow_kf_frame(kf1, k2, d2, kf2["feat_vec"], 0.7, True))
    add("orbx_search_by_bow_keyframes", "ORBmatcher.cc:722-866",
        lambda: m.SearchByBoWKeyFrames(kf1, kf2), lambda: oracle.search_by_bow_kf_kf(kf1, kf2, 0.7, True))
    F12 = np.array([[0, 0, 2.0], [0, 0, -3.0], [-2.0, 3.0, 0]], np.float32)
    add("orbx_search_for_triangulation", "ORBmatcher.cc:879-1087",
        lambda: m.SearchForTriangulation(kf1, kf2, F12, (-1000.0, -700.0), False),
        lambda: oracle.search_for_triangulation(kf1, kf2, F12, (-1000.0, -700.0), False, True, 0))

    # the CreateNewMapPoints loop: K neighbours, distances from one device call, selection per neighbour (us per neighbour)
    # (C ABI with the views marshalled once, as for the Fuse batches)
    for K in (8, 20):
        keep = []
        v1 = m._kf_view(kf1, keep); v2s = [m._kf_view(kf2, keep) for _ in range(K)]
        arr = (C.c_void_p * K)(*[C.addressof(v) for v in v2s])
        Fc = np.ascontiguousarray(F12, np.float32).reshape(9); out = np.full(max(v1.n, 1), -1, np.int32); nn = C.c_int(0)
        def tri_batch():
            b = C.c_void_p(0)
            _capi.check(Lc.orbx_triangulation_batch_create(ex.handle, C.byref(v1), K, arr, C.byref(b)))
            for k in range(K):
                _capi.check(Lc.orbx_triangulation_batch_select(b, k, C.byref(v1), C.byref(v2s[k]), _capi.ptr(Fc), -1000.0, -700.0, 0, 1, _capi.ptr(out), C.byref(nn)))
            Lc.orbx_triangulation_batch_destroy(b)
        def tri_single():
            for k in range(K):
                _capi.check(Lc.orbx_search_for_triangulation(ex.handle, C.byref(v1), C.byref(v2s[k]), _capi.ptr(Fc), -1000.0, -700.0, 0, 1, _capi.ptr(out), C.byref(nn)))
        g, g1 = bench(tri_batch, 10), bench(tri_single, 10)
        c = bench(lambda: [oracle.search_for_triangulation(kf1, kf2, F12, (-1000.0, -700.0), False, True, 0) for _ in range(K)], 3)
        rows.append(dict(entry_point=f"orbx_triangulation_batch K={K}", reference="LocalMapping.cc:375-430 loop", gpu_us=round(g / K, 1),
                         single_calls_us=round(g1 / K, 1), cpu_oracle_us=round(c / K, 1), ratio=round(c / g, 2), problems=K))
        print(f"{'orbx_triangulation_batch K=' + str(K):40s} {'us per neighbour (C ABI)':38s} GPU {g / K:9.1f} us   CPU oracle {c / K:10.1f} us   x{c / g:6.2f}   (K single calls at the C ABI: {g1 / K:.1f} us each)", flush=True)

    # ---- ComputeStereoMatches (KITTI size, host-buffer entry point) and brute-force best / second
    L, R = synth.stereo_pair(1241, 376, stream_id=6)
    exL, exR = ORBextractor(2000), ORBextractor(2000)
    kL, dL = exL(L); kR, dR = exR(R)
    FL, FR = Frame(kL, dL, 1241, 376), Frame(kR, dR, 1241, 376)
    oL, oR = oracle.OracleExtractor(2000), oracle.OracleExtractor(2000)
    oL.extract(L); oR.extract(R)
    pl = [oL.level_image(l) for l in range(8)]; pr = [oR.level_image(l) for l in range(8)]
    tabs = oL.tables()
    add("orbx_stereo_match", "Frame.cc:880-1176 (1241x376, 2000 kp)",
        lambda: FL.ComputeStereoMatches(FR, exL, exR, 0.537, 386.1448),
        lambda: oracle.stereo_matches(kL, dL, kR, dR, tabs["scale"], tabs["inv_scale"], pl, pr, 0.537, 386.1448))
    add("orbx_match_bruteforce", "DescriptorDistance x 1000 x 1000", lambda: m.match_bruteforce(d2, d1), lambda: oracle.match_bruteforce(d2, d1))
    if "--json" in sys.argv:
        json.dump(dict(host_cores=os.cpu_count(), note="host buffers in/out, synchronous; CPU = single-thread oracle (-O2 scalar C)", rows=rows),
                  open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)


if __name__ == "__main__":
    main()
