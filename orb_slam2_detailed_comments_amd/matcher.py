"""Host-side mirror of ORB_SLAM2::ORBmatcher (reference include/ORBmatcher.h:54-225) over the C ABI.

The Hamming distances come from the HIP kernels; the order-dependent bookkeeping of each policy stays on the
host exactly as SURVEY.md Appendix E prescribes.
"""
from __future__ import annotations
import numpy as np
from . import _capi
from ._capi import check, ptr, lib


class ORBmatcher:
    TH_HIGH, TH_LOW, HISTO_LENGTH = 100, 50, 30  # src/ORBmatcher.cc:49-51

    def __init__(self, nnratio=0.6, checkOri=True, *, extractor=None, device=-1):
        from .extractor import ORBextractor
        self.mfNNratio = np.float32(nnratio)
        self.mbCheckOrientation = bool(checkOri)
        self._own = extractor is None
        self._ex = extractor if extractor is not None else ORBextractor(device=device)
        self._L = lib()

    # ---- DescriptorDistance (src/ORBmatcher.cc:2073-2093), batched: full matrix on the GPU
    def distance_matrix(self, q, t):
        q = np.ascontiguousarray(q, np.uint8); t = np.ascontiguousarray(t, np.uint8)
        out = np.zeros((len(q), len(t)), np.uint16)
        if len(q) and len(t):
            check(self._L.orbx_hamming_matrix(self._ex.handle, ptr(q), len(q), ptr(t), len(t), ptr(out)))
        return out

    def DescriptorDistance(self, a, b):
        return int(self.distance_matrix(np.asarray(a).reshape(1, 32), np.asarray(b).reshape(1, 32))[0, 0])

    # ---- brute-force best / second best (the superset primitive named in SURVEY section 8d)
    def match_bruteforce(self, q, t):
        q = np.ascontiguousarray(q, np.uint8); t = np.ascontiguousarray(t, np.uint8)
        nq, nt = len(q), len(t)
        bi = np.full(nq, -1, np.int32); bd = np.full(nq, 0x7fffffff, np.int32); sd = np.full(nq, 0x7fffffff, np.int32)
        if nq:
            check(self._L.orbx_match_bruteforce(self._ex.handle, ptr(q), nq, ptr(t), nt, ptr(bi), ptr(bd), ptr(sd)))
        return bi, bd, sd

    def match_ratio(self, q, t, th=None):
        """best <= th and best < ratio * second  ->  train index, else -1"""
        bi, bd, sd = self.match_bruteforce(q, t)
        th = self.TH_LOW if th is None else th
        ok = (bd <= th) & (bd.astype(np.float32) < sd.astype(np.float32) * self.mfNNratio)
        return np.where(ok, bi, -1).astype(np.int32)

    # ---- SearchForInitialization (src/ORBmatcher.cc:570-712; caller src/Tracking.cc:950-962)
    def SearchForInitialization(self, F1, F2, vbPrevMatched, windowSize=10):
        """F1, F2: frame.Frame; vbPrevMatched: [N1,2] float32, updated in place.  Returns (nmatches, vnMatches12)."""
        import ctypes as C
        pm = np.ascontiguousarray(vbPrevMatched, np.float32)
        m12 = np.full(max(F1.N, 1), -1, np.int32)
        bounds = np.asarray(F2.bounds, np.float32)
        n = C.c_int(0)
        check(self._L.orbx_search_for_initialization(
            self._ex.handle, ptr(F1.mvKeysUn), ptr(F1.mDescriptors), F1.N, ptr(F2.mvKeysUn), ptr(F2.mDescriptors), F2.N,
            ptr(bounds), ptr(pm), int(windowSize), float(self.mfNNratio), int(self.mbCheckOrientation), ptr(m12),
            C.byref(n)))
        if pm is not vbPrevMatched:
            vbPrevMatched[...] = pm
        return n.value, m12[:F1.N].copy()

    # ---- SearchByProjection(CurrentFrame, LastFrame, th, bMono) (src/ORBmatcher.cc:1702-1871)
    def SearchByProjection(self, cur, last, th, bMono, *, Tcw, Tlw, K, mb, mbf, has_map_point, world_pos, mp_desc,
                           observations):
        """cur, last: frame.Frame; MapPoint state of the last frame as arrays.  Returns (nmatches, matched_last[cur.N])."""
        import ctypes as C
        keep = [np.ascontiguousarray(a, t) for a, t in ((cur.mvuRight, np.float32), (has_map_point, np.uint8),
                                                         (world_pos, np.float32), (mp_desc, np.uint8),
                                                         (observations, np.int32))]
        cv, lv = _capi.FrameView(), _capi.LastFrameView()
        cv.keys_un, cv.desc, cv.u_right, cv.n = cur.mvKeysUn.ctypes.data, cur.mDescriptors.ctypes.data, keep[0].ctypes.data, cur.N
        cv.Tcw[:] = [float(v) for v in np.asarray(Tcw, np.float32).reshape(16)]
        cv.fx, cv.fy, cv.cx, cv.cy = [float(v) for v in K]
        cv.min_x, cv.max_x, cv.min_y, cv.max_y = cur.bounds
        cv.mb, cv.mbf = float(mb), float(mbf)
        lv.keys_un, lv.n = last.mvKeysUn.ctypes.data, last.N
        lv.has_map_point, lv.world_pos, lv.mp_desc, lv.observations = (k.ctypes.data for k in keep[1:])
        lv.Tcw[:] = [float(v) for v in np.asarray(Tlw, np.float32).reshape(16)]
        out = np.full(max(cur.N, 1), -1, np.int32)
        n = C.c_int(0)
        check(self._L.orbx_search_by_projection_frame(self._ex.handle, C.byref(cv), C.byref(lv), float(th), int(bMono),
                                                      int(self.mbCheckOrientation), ptr(out), C.byref(n)))
        return n.value, out[:cur.N].copy()

    # ---- SearchByProjection(F, vpMapPoints, th) (src/ORBmatcher.cc:69-184): local-map tracking
    def SearchByProjectionMapPoints(self, F, th, *, frame_observations, in_view, proj, level, view_cos, mp_desc,
                                    observations):
        """F: frame.Frame; MapPoint fields as arrays.  Returns (nmatches, assigned[F.N])."""
        import ctypes as C
        keep = [np.ascontiguousarray(a, t) for a, t in ((F.mvuRight, np.float32), (frame_observations, np.int32),
                                                         (in_view, np.uint8), (proj, np.float32), (level, np.int32),
                                                         (view_cos, np.float32), (mp_desc, np.uint8), (observations, np.int32))]
        fv, mv = _capi.FrameView(), _capi.MapPointView()
        fv.keys_un, fv.desc, fv.u_right, fv.n = F.mvKeysUn.ctypes.data, F.mDescriptors.ctypes.data, keep[0].ctypes.data, F.N
        fv.min_x, fv.max_x, fv.min_y, fv.max_y = F.bounds
        mv.n = len(keep[2])
        mv.in_view, mv.proj, mv.level, mv.view_cos, mv.desc, mv.observations = (k.ctypes.data for k in keep[2:])
        out = np.full(max(F.N, 1), -1, np.int32)
        n = C.c_int(0)
        check(self._L.orbx_search_by_projection_mappoints(self._ex.handle, C.byref(fv), ptr(keep[1]), C.byref(mv), float(th),
                                                          float(self.mfNNratio), ptr(out), C.byref(n)))
        return n.value, out[:F.N].copy()

    # ---- BoW-guided policies (src/ORBmatcher.cc:248-410, 722-866, 879-1087)
    @staticmethod
    def _featvec(fv, keep):
        """fv: dict {node_id: [feature indices]} (DBoW2::FeatureVector) or (node_id, begin, index) arrays"""
        if isinstance(fv, dict):
            nodes = sorted(fv)
            begin = np.zeros(len(nodes) + 1, np.int32)
            for i, k in enumerate(nodes):
                begin[i + 1] = begin[i] + len(fv[k])
            index = np.array([j for k in nodes for j in fv[k]], np.uint32)
            node_id = np.array(nodes, np.uint32)
        else:
            node_id, begin, index = (np.ascontiguousarray(a, t) for a, t in zip(fv, (np.uint32, np.int32, np.uint32)))
        keep += [node_id, begin, index]
        v = _capi.FeatVecView()
        v.n_nodes = len(node_id)
        v.node_id, v.begin, v.index = node_id.ctypes.data, begin.ctypes.data, index.ctypes.data
        return v

    def _kf_view(self, kf, keep):
        """kf: dict with keys_un, desc, has_map_point, feat_vec and optionally u_right, scale_factors, level_sigma2"""
        a = lambda x, t: np.ascontiguousarray(x, t)
        keys, desc, has = a(kf["keys_un"], _capi.KP_DTYPE), a(kf["desc"], np.uint8), a(kf["has_map_point"], np.uint8)
        keep += [keys, desc, has]
        v = _capi.KeyFrameView()
        v.keys_un, v.desc, v.n, v.has_map_point = keys.ctypes.data, desc.ctypes.data, len(keys), has.ctypes.data
        for name, t in (("u_right", np.float32), ("scale_factors", np.float32), ("level_sigma2", np.float32)):
            if kf.get(name) is not None:
                arr = a(kf[name], t); keep.append(arr); setattr(v, name, arr.ctypes.data)
        v.feat_vec = self._featvec(kf["feat_vec"], keep)
        return v

    def SearchByBoW(self, kf, f_keys, f_desc, f_feat_vec):
        """SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches).  Returns (nmatches, matched_kf[F.N])."""
        import ctypes as C
        keep = []
        kv = self._kf_view(kf, keep)
        fk, fd = np.ascontiguousarray(f_keys, _capi.KP_DTYPE), np.ascontiguousarray(f_desc, np.uint8)
        fv = self._featvec(f_feat_vec, keep)
        out = np.full(max(len(fk), 1), -1, np.int32)
        n = C.c_int(0)
        check(self._L.orbx_search_by_bow_keyframe_frame(self._ex.handle, C.byref(kv), ptr(fk), ptr(fd), len(fk), C.byref(fv),
                                                        float(self.mfNNratio), int(self.mbCheckOrientation), ptr(out), C.byref(n)))
        return n.value, out[:len(fk)].copy()

    def SearchByBoWKeyFrames(self, kf1, kf2):
        """SearchByBoW(KeyFrame*, KeyFrame*, vpMatches12).  Returns (nmatches, matches12[KF1.N])."""
        import ctypes as C
        keep = []
        v1, v2 = self._kf_view(kf1, keep), self._kf_view(kf2, keep)
        out = np.full(max(v1.n, 1), -1, np.int32)
        n = C.c_int(0)
        check(self._L.orbx_search_by_bow_keyframes(self._ex.handle, C.byref(v1), C.byref(v2), float(self.mfNNratio),
                                                   int(self.mbCheckOrientation), ptr(out), C.byref(n)))
        return n.value, out[:v1.n].copy()

    def SearchForTriangulation(self, kf1, kf2, F12, epipole, bOnlyStereo=False):
        """Returns (nmatches, matches12[KF1.N]); vMatchedPairs = [(i, m) for i, m in enumerate(matches12) if m >= 0]."""
        import ctypes as C
        keep = []
        v1, v2 = self._kf_view(kf1, keep), self._kf_view(kf2, keep)
        F = np.ascontiguousarray(F12, np.float32).reshape(9)
        out = np.full(max(v1.n, 1), -1, np.int32)
        n = C.c_int(0)
        check(self._L.orbx_search_for_triangulation(self._ex.handle, C.byref(v1), C.byref(v2), ptr(F), float(epipole[0]),
                                                    float(epipole[1]), int(bOnlyStereo), int(self.mbCheckOrientation),
                                                    ptr(out), C.byref(n)))
        return n.value, out[:v1.n].copy()

    def TriangulationBatch(self, kf1, kf2_list):
        """the loop of LocalMapping::CreateNewMapPoints (src/LocalMapping.cc:375-430) as one device round trip: the distances of
        kf1 against every neighbour now, the selection per neighbour later (.select(k, ...)) with the has_map_point flags of that
        moment.  Use as a context manager or call .close()."""
        return _TriangulationBatch(self, kf1, kf2_list)

    # ---- projection-guided back-end policies (src/ORBmatcher.cc:1100-1280, 1282-1430, 415-560, 1433-1690, 1873-2020)
    @staticmethod
    def _target(t, keep):
        """t: dict keys_un, desc, bounds (minx, maxx, miny, maxy), scale_factors [, u_right, inv_level_sigma2]"""
        a = lambda x, ty: np.ascontiguousarray(x, ty)
        keys, desc, sf = a(t["keys_un"], _capi.KP_DTYPE), a(t["desc"], np.uint8), a(t["scale_factors"], np.float32)
        keep += [keys, desc, sf]
        v = _capi.TargetView()
        v.keys_un, v.desc, v.n, v.scale_factors = keys.ctypes.data, desc.ctypes.data, len(keys), sf.ctypes.data
        v.min_x, v.max_x, v.min_y, v.max_y = [float(b) for b in t["bounds"]]
        for name in ("u_right", "inv_level_sigma2"):
            if t.get(name) is not None:
                arr = a(t[name], np.float32); keep.append(arr); setattr(v, name, arr.ctypes.data)
        return v

    @staticmethod
    def _points(p, keep):
        """p: dict valid, uv [n,2], level, desc [, u_right, angle]"""
        a = lambda x, ty: np.ascontiguousarray(x, ty)
        valid, uv, level, desc = a(p["valid"], np.uint8), a(p["uv"], np.float32), a(p["level"], np.int32), a(p["desc"], np.uint8)
        keep += [valid, uv, level, desc]
        v = _capi.ProjectedPoints()
        v.n, v.valid, v.uv, v.level, v.desc = len(valid), valid.ctypes.data, uv.ctypes.data, level.ctypes.data, desc.ctypes.data
        for name in ("u_right", "angle"):
            if p.get(name) is not None:
                arr = a(p[name], np.float32); keep.append(arr); setattr(v, name, arr.ctypes.data)
        return v

    def Fuse(self, kf, pts, th=3.0):
        """Fuse(KeyFrame*, vpMapPoints, th), selection part.  Returns (nFused, best_idx[n_points])."""
        import ctypes as C
        keep = []; tv, pv = self._target(kf, keep), self._points(pts, keep)
        out = np.full(max(pv.n, 1), -1, np.int32); n = C.c_int(0)
        check(self._L.orbx_fuse(self._ex.handle, C.byref(tv), C.byref(pv), float(th), ptr(out), C.byref(n)))
        return n.value, out[:pv.n].copy()

    def FuseBatch(self, kfs, pts_list, th=3.0, sim3=False):
        """the loop `for pKFi in vpTargetKFs: matcher.Fuse(pKFi, ...)` of LocalMapping::SearchInNeighbors (src/LocalMapping.cc:750-768)
        / LoopClosing::SearchAndFuse as ONE call: K (keyframe, projected points) problems through one upload, one grid-build + gate
        launch pair, one download.  Returns ([nFused_k], [best_idx_k]) -- per problem exactly what Fuse / FuseSim3 return."""
        import ctypes as C
        K = len(kfs)
        assert K == len(pts_list)
        keep = []
        tvs = [self._target(kf, keep) for kf in kfs]; pvs = [self._points(p, keep) for p in pts_list]
        outs = [np.full(max(pv.n, 1), -1, np.int32) for pv in pvs]
        tarr = (C.c_void_p * max(K, 1))(*[C.addressof(t) for t in tvs]); parr = (C.c_void_p * max(K, 1))(*[C.addressof(p_) for p_ in pvs])
        oarr = (C.c_void_p * max(K, 1))(*[o.ctypes.data for o in outs])
        n = (C.c_int * max(K, 1))()
        fn = self._L.orbx_fuse_sim3_batch if sim3 else self._L.orbx_fuse_batch
        check(fn(self._ex.handle, K, tarr, parr, C.c_float(th), oarr, n))
        return [int(n[k]) for k in range(K)], [outs[k][:pvs[k].n].copy() for k in range(K)]

    def FuseSim3(self, kf, pts, th):
        """Fuse(KeyFrame*, Scw, vpPoints, th, vpReplacePoint), selection part."""
        import ctypes as C
        keep = []; tv, pv = self._target(kf, keep), self._points(pts, keep)
        out = np.full(max(pv.n, 1), -1, np.int32); n = C.c_int(0)
        check(self._L.orbx_fuse_sim3(self._ex.handle, C.byref(tv), C.byref(pv), float(th), ptr(out), C.byref(n)))
        return n.value, out[:pv.n].copy()

    def SearchByProjectionSim3(self, kf, pts, matched, th):
        """SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th); `matched` (uint8 per feature) is updated in place."""
        import ctypes as C
        keep = []; tv, pv = self._target(kf, keep), self._points(pts, keep)
        assert matched.dtype == np.uint8 and matched.flags.c_contiguous and len(matched) == tv.n
        out = np.full(max(pv.n, 1), -1, np.int32); n = C.c_int(0)
        check(self._L.orbx_search_by_projection_sim3(self._ex.handle, C.byref(tv), C.byref(pv), int(th), ptr(matched), ptr(out),
                                                     C.byref(n)))
        return n.value, out[:pv.n].copy()

    def SearchBySim3(self, kf1, kf2, pts1_in_2, pts2_in_1, th):
        import ctypes as C
        keep = []
        t1, t2 = self._target(kf1, keep), self._target(kf2, keep)
        p12, p21 = self._points(pts1_in_2, keep), self._points(pts2_in_1, keep)
        out = np.full(max(t1.n, 1), -1, np.int32); n = C.c_int(0)
        check(self._L.orbx_search_by_sim3(self._ex.handle, C.byref(t1), C.byref(t2), C.byref(p12), C.byref(p21), float(th),
                                          ptr(out), C.byref(n)))
        return n.value, out[:t1.n].copy()

    def SearchByProjectionKeyFrame(self, cur, pts, cur_has_map_point, th, ORBdist):
        """SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist); cur_has_map_point updated in place."""
        import ctypes as C
        keep = []; tv, pv = self._target(cur, keep), self._points(pts, keep)
        assert cur_has_map_point.dtype == np.uint8 and len(cur_has_map_point) == tv.n
        out = np.full(max(tv.n, 1), -1, np.int32); n = C.c_int(0)
        check(self._L.orbx_search_by_projection_keyframe(self._ex.handle, C.byref(tv), C.byref(pv), float(th), int(ORBdist),
                                                         int(self.mbCheckOrientation), ptr(cur_has_map_point), ptr(out), C.byref(n)))
        return n.value, out[:tv.n].copy()

    # ---- ComputeThreeMaxima (src/ORBmatcher.cc:2026-2068): 30 numbers, host side
    @staticmethod
    def ComputeThreeMaxima(sizes):
        max1 = max2 = max3 = 0
        ind1 = ind2 = ind3 = -1
        for i, s in enumerate(sizes):
            s = int(s)
            if s > max1:
                max3, max2, max1 = max2, max1, s
                ind3, ind2, ind1 = ind2, ind1, i
            elif s > max2:
                max3, max2 = max2, s
                ind3, ind2 = ind2, i
            elif s > max3:
                max3, ind3 = s, i
        if max2 < np.float32(0.1) * np.float32(max1):
            ind2 = ind3 = -1
        elif max3 < np.float32(0.1) * np.float32(max1):
            ind3 = -1
        return ind1, ind2, ind3


class _TriangulationBatch:
    def __init__(self, m, kf1, kf2_list):
        import ctypes as C
        self._m, self._keep = m, []
        v1 = m._kf_view(kf1, self._keep)
        v2 = [m._kf_view(k, self._keep) for k in kf2_list]
        arr = (C.c_void_p * max(len(v2), 1))(*[C.addressof(v) for v in v2])
        self._b = C.c_void_p(0)
        check(m._L.orbx_triangulation_batch_create(m._ex.handle, C.byref(v1), len(v2), arr, C.byref(self._b)))
        self.n = len(v2)

    def select(self, k, kf1_now, kf2_now, F12, epipole, bOnlyStereo=False):
        """SearchForTriangulation(kf1, kf2[k]) on the batch's distances; kf1_now / kf2_now carry the CURRENT has_map_point flags"""
        import ctypes as C
        keep = []
        v1, v2 = self._m._kf_view(kf1_now, keep), self._m._kf_view(kf2_now, keep)
        F = np.ascontiguousarray(F12, np.float32).reshape(9)
        out = np.full(max(v1.n, 1), -1, np.int32); n = C.c_int(0)
        check(self._m._L.orbx_triangulation_batch_select(self._b, int(k), C.byref(v1), C.byref(v2), ptr(F), float(epipole[0]), float(epipole[1]),
                                                         int(bOnlyStereo), int(self._m.mbCheckOrientation), ptr(out), C.byref(n)))
        return n.value, out[:v1.n].copy()

    def close(self):
        if self._b:
            self._m._L.orbx_triangulation_batch_destroy(self._b); self._b = None

    def __enter__(self): return self
    def __exit__(self, *a): self.close()
    def __del__(self): self.close()
