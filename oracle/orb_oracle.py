"""ctypes binding of the CPU oracle (oracle/orb_oracle.h).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28

FP_GCC_FMA, FP_STRICT = 0, 1


def build(force=False):
    if os.environ.get("ORB_ORACLE_LIB"):   # another build of the same sources (sanitizer run, tests/test_sanitizers.py)
        return os.environ["ORB_ORACLE_LIB"]
    so = os.path.join(_HERE, "liborb_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("orb_oracle.c", "orb_oracle_match.c", "orb_oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liborb_oracle.so"], stdout=subprocess.DEVNULL)
    return so


NATIVE_FLAGS = "-O3 -march=native -ffp-contract=off"


def build_native():
    """the same sources with the reference's own optimisation flags (CMakeLists.txt:10-11), built on THIS machine (-march=native);
    select it for a process with ORB_ORACLE_LIB=<path> before the first oracle call"""
    subprocess.check_call(["make", "-C", _HERE, "-B", "_native/liborb_oracle_native.so"], stdout=subprocess.DEVNULL)
    return os.path.join(_HERE, "_native", "liborb_oracle_native.so")


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        u8p, i32p, f32p = C.POINTER(C.c_uint8), C.POINTER(C.c_int), C.POINTER(C.c_float)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_extract.restype = C.c_int
        L.orc_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_get_tables.argtypes = [C.c_void_p] + [C.c_void_p] * 6
        L.orc_level_dims.argtypes = [C.c_void_p, C.c_int, i32p, i32p]
        L.orc_level_image.restype = C.c_void_p
        L.orc_level_image.argtypes = [C.c_void_p, C.c_int]
        L.orc_level_blur.restype = C.c_void_p
        L.orc_level_blur.argtypes = [C.c_void_p, C.c_int]
        L.orc_level_candidates.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orc_level_keypoints.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orc_get_stage_times.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_border_reflect101.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
        L.orc_resize_linear_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.orc_fast9_16.restype = C.c_int
        L.orc_fast9_16.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_gaussian_blur7.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_fast_atan2.restype = C.c_float
        L.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.orc_cv_round_f.restype = C.c_int
        L.orc_cv_round_f.argtypes = [C.c_float]
        L.orc_distribute_octtree.restype = C.c_int
        L.orc_distribute_octtree.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_ic_angle.restype = C.c_float
        L.orc_ic_angle.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.orc_descriptor.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]
        L.orc_descriptor_distance.restype = C.c_int
        L.orc_descriptor_distance.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_match_bruteforce.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_three_maxima.argtypes = [C.c_void_p, C.c_int, i32p, i32p, i32p]
        L.orc_grid_build.restype = C.c_void_p
        L.orc_grid_build.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float]
        L.orc_grid_free.argtypes = [C.c_void_p]
        L.orc_grid_query.restype = C.c_int
        L.orc_grid_query.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_search_for_initialization.restype = C.c_int
        L.orc_search_for_initialization.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                                    C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_int,
                                                    C.c_float, C.c_int, C.c_void_p]
        L.orc_stereo_matches.restype = C.c_int
        L.orc_stereo_matches.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.orc_search_by_projection_ff.restype = C.c_int
        L.orc_search_by_projection_ff.argtypes = ([C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p] + [C.c_float] * 10 +
                                                  [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                                   C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_int, C.c_void_p])
        L.orc_search_by_projection_mp.restype = C.c_int
        L.orc_search_by_projection_mp.argtypes = ([C.c_void_p] * 4 + [C.c_int] + [C.c_float] * 4 + [C.c_void_p, C.c_int] +
                                                  [C.c_void_p] * 6 + [C.c_float, C.c_float, C.c_void_p])
        L.orc_search_by_bow_kf_frame.restype = C.c_int
        L.orc_search_by_bow_kf_frame.argtypes = ([C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 3 +
                                                 [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 3 +
                                                 [C.c_float, C.c_int, C.c_void_p])
        L.orc_search_by_bow_kf_kf.restype = C.c_int
        L.orc_search_by_bow_kf_kf.argtypes = ([C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 3) * 2 + \
                                             [C.c_float, C.c_int, C.c_void_p]
        L.orc_search_for_triangulation.restype = C.c_int
        L.orc_search_for_triangulation.argtypes = ([C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int] +
                                                   [C.c_void_p] * 3) * 2 + [C.c_void_p, C.c_float, C.c_float, C.c_void_p,
                                                                            C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        vp_, ci, cf = C.c_void_p, C.c_int, C.c_float
        L.orc_fuse.restype = ci
        L.orc_fuse.argtypes = [vp_, vp_, vp_, ci, cf, cf, cf, cf, vp_, vp_, ci, vp_, vp_, vp_, vp_, vp_, cf, ci, vp_]
        L.orc_fuse_sim3.restype = ci
        L.orc_fuse_sim3.argtypes = [vp_, vp_, ci, cf, cf, cf, cf, vp_, ci, vp_, vp_, vp_, vp_, cf, vp_]
        L.orc_search_by_projection_sim3.restype = ci
        L.orc_search_by_projection_sim3.argtypes = [vp_, vp_, ci, cf, cf, cf, cf, vp_, ci, vp_, vp_, vp_, vp_, ci, vp_, vp_]
        L.orc_search_by_sim3.restype = ci
        L.orc_search_by_sim3.argtypes = [vp_, vp_, ci, vp_, vp_] * 2 + [vp_] * 8 + [cf, vp_]
        L.orc_search_by_projection_kf.restype = ci
        L.orc_search_by_projection_kf.argtypes = [vp_, vp_, ci, cf, cf, cf, cf, vp_, ci, vp_, vp_, vp_, vp_, vp_, cf, ci, ci, vp_, vp_]
        L.orc_bow_transform.restype = None
        L.orc_bow_transform.argtypes = [ci, ci] + [vp_] * 6 + [ci, ci] + [vp_] * 3
        L.orc_bow_vectors.restype = ci
        L.orc_bow_vectors.argtypes = [ci, ci, vp_, vp_, vp_, ci, vp_, vp_, vp_, vp_, vp_, vp_, vp_]
        L.orc_undistort_points.restype = None
        L.orc_undistort_points.argtypes = [vp_, ci, cf, cf, cf, cf, vp_, ci, vp_]
        L.orc_undistort_keypoints.restype = None
        L.orc_undistort_keypoints.argtypes = [vp_, ci, cf, cf, cf, cf, vp_, ci, vp_]
        L.orc_image_bounds.restype = None
        L.orc_image_bounds.argtypes = [ci, ci, cf, cf, cf, cf, vp_, ci, vp_]
        L.orc_remap_linear_u8.restype = None
        L.orc_remap_linear_u8.argtypes = [vp_, ci, ci, ci, vp_, vp_, ci, ci, vp_, ci]
        L.orc_cvt_gray.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleExtractor:
    """CPU oracle of ORBextractor (src/ORBextractor.cc:776-925, 1961-2084)."""

    def __init__(self, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7, fp_mode=FP_GCC_FMA):
        self.L = lib()
        self.nlevels = nlevels
        self.nfeatures = nfeatures
        self.h = self.L.orc_create(nfeatures, scale_factor, nlevels, ini_th, min_th, fp_mode)
        if not self.h:
            raise ValueError("orc_create failed")

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_destroy(self.h)
            self.h = None

    def tables(self):
        n = self.nlevels
        sc, inv, s2, is2 = (np.zeros(n, np.float32) for _ in range(4))
        fpl = np.zeros(n, np.int32); umax = np.zeros(16, np.int32)
        self.L.orc_get_tables(self.h, _p(sc), _p(inv), _p(s2), _p(is2), _p(fpl), _p(umax))
        return dict(scale=sc, inv_scale=inv, sigma2=s2, inv_sigma2=is2, features_per_level=fpl, umax=umax)

    def extract(self, img, cap=None):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        if cap is None:
            cap = self.nfeatures + 64 * self.nlevels
        kps = np.zeros(cap, KP_DTYPE); desc = np.zeros((cap, 32), np.uint8)
        n = self.L.orc_extract(self.h, _p(img), w, h, img.strides[0], _p(kps), _p(desc), cap)
        if n < 0:
            return n, None, None
        return n, kps[:n].copy(), desc[:n].copy()

    def level_image(self, level, blur=False):
        w, h = C.c_int(), C.c_int()
        if self.L.orc_level_dims(self.h, level, C.byref(w), C.byref(h)) != 0:
            return None
        ptr = (self.L.orc_level_blur if blur else self.L.orc_level_image)(self.h, level)
        if not ptr:
            return None
        buf = (C.c_uint8 * (w.value * h.value)).from_address(ptr)
        return np.frombuffer(buf, np.uint8).reshape(h.value, w.value).copy()

    def level_candidates(self, level):
        n = self.L.orc_level_candidates(self.h, level, None, 0)
        out = np.zeros(max(n, 1), KP_DTYPE)
        self.L.orc_level_candidates(self.h, level, _p(out), n)
        return out[:n]

    def level_keypoints(self, level):
        n = self.L.orc_level_keypoints(self.h, level, None, 0)
        out = np.zeros(max(n, 1), KP_DTYPE)
        self.L.orc_level_keypoints(self.h, level, _p(out), n)
        return out[:n]

    def stage_times(self):
        t = np.zeros(6, np.float64)
        self.L.orc_get_stage_times(self.h, _p(t))
        return dict(zip(("pyramid", "fast", "quadtree", "orient", "blur", "desc"), t))


def fast9_16(img, threshold):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.zeros(w * h, KP_DTYPE)
    n = lib().orc_fast9_16(_p(img), w, h, img.strides[0], threshold, _p(out), out.size)
    return out[:n].copy()


def resize_linear(img, dw, dh):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.zeros((dh, dw), np.uint8)
    lib().orc_resize_linear_u8(_p(img), w, h, img.strides[0], _p(out), dw, dh, dw)
    return out


def border101(img, border):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.zeros((h + 2 * border, w + 2 * border), np.uint8)
    lib().orc_border_reflect101(_p(img), w, h, img.strides[0], _p(out), out.shape[1], border)
    return out


def gaussian_blur7(img):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.zeros_like(img)
    lib().orc_gaussian_blur7(_p(img), w, h, img.strides[0], _p(out), w)
    return out


def distribute_octtree(keys, minX, maxX, minY, maxY, N):
    keys = np.ascontiguousarray(keys, KP_DTYPE)
    cap = len(keys) + 16
    idx = np.zeros(cap, np.int32)
    n = lib().orc_distribute_octtree(_p(keys), len(keys), minX, maxX, minY, maxY, N, _p(idx), cap)
    return n, idx[:max(n, 0)].copy()


def descriptor_distance(a, b):
    a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
    return lib().orc_descriptor_distance(_p(a), _p(b))


def match_bruteforce(q, t):
    q = np.ascontiguousarray(q, np.uint8); t = np.ascontiguousarray(t, np.uint8)
    nq, nt = len(q), len(t)
    bi = np.zeros(nq, np.int32); bd = np.zeros(nq, np.int32); sd = np.zeros(nq, np.int32)
    lib().orc_match_bruteforce(_p(q), nq, _p(t), nt, _p(bi), _p(bd), _p(sd))
    return bi, bd, sd


def three_maxima(sizes):
    sizes = np.ascontiguousarray(sizes, np.int32)
    a, b, c = C.c_int(), C.c_int(), C.c_int()
    lib().orc_three_maxima(_p(sizes), len(sizes), C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


def grid_query(kps, bounds, x, y, r, min_level=-1, max_level=-1):
    """Frame::GetFeaturesInArea over a freshly assigned 64x48 grid; bounds = (minx, maxx, miny, maxy)"""
    kps = np.ascontiguousarray(kps, KP_DTYPE)
    g = lib().orc_grid_build(_p(kps), len(kps), *[float(b) for b in bounds])
    out = np.zeros(max(len(kps), 1), np.int32)
    n = lib().orc_grid_query(g, x, y, r, min_level, max_level, _p(out), len(out))
    lib().orc_grid_free(g)
    return out[:n].copy()


def search_for_initialization(k1, d1, k2, d2, bounds, prev_matched, window=100, nnratio=0.9, check_ori=True):
    k1 = np.ascontiguousarray(k1, KP_DTYPE); k2 = np.ascontiguousarray(k2, KP_DTYPE)
    d1 = np.ascontiguousarray(d1, np.uint8); d2 = np.ascontiguousarray(d2, np.uint8)
    pm = np.ascontiguousarray(prev_matched, np.float32).copy()
    m12 = np.zeros(max(len(k1), 1), np.int32)
    n = lib().orc_search_for_initialization(_p(k1), _p(d1), len(k1), _p(k2), _p(d2), len(k2),
                                            *[float(b) for b in bounds], _p(pm), window, nnratio, int(check_ori), _p(m12))
    return n, m12[:len(k1)].copy(), pm


def stereo_matches(kL, dL, kR, dR, scale, inv_scale, pyrL, pyrR, mb, mbf):
    """pyrL / pyrR: lists of contiguous padded level images (uint8 2-D arrays)"""
    kL = np.ascontiguousarray(kL, KP_DTYPE); kR = np.ascontiguousarray(kR, KP_DTYPE)
    dL = np.ascontiguousarray(dL, np.uint8); dR = np.ascontiguousarray(dR, np.uint8)
    nl = len(pyrL)
    pyrL = [np.ascontiguousarray(a) for a in pyrL]; pyrR = [np.ascontiguousarray(a) for a in pyrR]
    PL = (C.c_void_p * nl)(*[a.ctypes.data for a in pyrL]); PR = (C.c_void_p * nl)(*[a.ctypes.data for a in pyrR])
    pw = np.array([a.shape[1] for a in pyrL], np.int32); ph = np.array([a.shape[0] for a in pyrL], np.int32)
    sc = np.ascontiguousarray(scale, np.float32); isc = np.ascontiguousarray(inv_scale, np.float32)
    uR = np.zeros(max(len(kL), 1), np.float32); dep = np.zeros(max(len(kL), 1), np.float32)
    n = lib().orc_stereo_matches(_p(kL), _p(dL), len(kL), _p(kR), _p(dR), len(kR), nl, _p(sc), _p(isc), PL, PR, _p(pw),
                                 _p(ph), mb, mbf, _p(uR), _p(dep))
    return n, uR[:len(kL)].copy(), dep[:len(kL)].copy()


def search_by_projection_ff(kc, dc, u_right, Tcw, K, bounds, mb, mbf, scale, kl, has_mp, xw, mpdesc, obs, Tlw, th, mono,
                            check_ori=True, fp_mode=FP_GCC_FMA):
    """K = (fx, fy, cx, cy); bounds = (minx, maxx, miny, maxy)"""
    kc = np.ascontiguousarray(kc, KP_DTYPE); kl = np.ascontiguousarray(kl, KP_DTYPE)
    arr = lambda a, t: np.ascontiguousarray(a, t)
    dc, mpdesc = arr(dc, np.uint8), arr(mpdesc, np.uint8)
    u_right, xw, Tcw, Tlw, scale = (arr(a, np.float32) for a in (u_right, xw, Tcw, Tlw, scale))
    has_mp, obs = arr(has_mp, np.uint8), arr(obs, np.int32)
    out = np.full(max(len(kc), 1), -1, np.int32)
    n = lib().orc_search_by_projection_ff(_p(kc), _p(dc), _p(u_right), len(kc), _p(Tcw), *[float(v) for v in K],
                                          *[float(b) for b in bounds], mb, mbf, _p(scale), _p(kl), len(kl), _p(has_mp),
                                          _p(xw), _p(mpdesc), _p(obs), _p(Tlw), th, int(mono), int(check_ori), fp_mode, _p(out))
    return n, out[:len(kc)].copy()


def search_by_projection_mp(kf, df, u_right, frame_obs, bounds, scale, in_view, proj, level, view_cos, mpdesc, mp_obs, th,
                            nnratio):
    kf = np.ascontiguousarray(kf, KP_DTYPE)
    a = lambda x, t: np.ascontiguousarray(x, t)
    df, mpdesc, in_view = a(df, np.uint8), a(mpdesc, np.uint8), a(in_view, np.uint8)
    u_right, proj, view_cos, scale = (a(x, np.float32) for x in (u_right, proj, view_cos, scale))
    frame_obs, level, mp_obs = (a(x, np.int32) for x in (frame_obs, level, mp_obs))
    out = np.full(max(len(kf), 1), -1, np.int32)
    n = lib().orc_search_by_projection_mp(_p(kf), _p(df), _p(u_right), _p(frame_obs), len(kf), *[float(b) for b in bounds],
                                          _p(scale), len(in_view), _p(in_view), _p(proj), _p(level), _p(view_cos), _p(mpdesc),
                                          _p(mp_obs), th, nnratio, _p(out))
    return n, out[:len(kf)].copy()


def cvt_gray(img, rgb_order=True):
    """cv::cvtColor to gray for HxWx3 / HxWx4 uint8 images (rgb_order False = BGR / BGRA)"""
    img = np.ascontiguousarray(img, np.uint8)
    h, w, c = img.shape
    out = np.zeros((h, w), np.uint8)
    lib().orc_cvt_gray(_p(img), w, h, img.strides[0], c, int(rgb_order), _p(out), w)
    return out


def _flat_featvec(fv):
    """dict {node: [indices]} -> (node_id u32 ascending, begin i32, index u32)"""
    nodes = sorted(fv)
    begin = np.zeros(len(nodes) + 1, np.int32)
    for i, k in enumerate(nodes):
        begin[i + 1] = begin[i] + len(fv[k])
    return (np.array(nodes, np.uint32), begin, np.array([j for k in nodes for j in fv[k]], np.uint32))


def search_by_bow_kf_frame(kf, f_keys, f_desc, f_fv, nnratio, check_ori=True):
    kk, dk, has = (np.ascontiguousarray(kf["keys_un"], KP_DTYPE), np.ascontiguousarray(kf["desc"], np.uint8),
                   np.ascontiguousarray(kf["has_map_point"], np.uint8))
    n1, b1, i1 = _flat_featvec(kf["feat_vec"]); n2, b2, i2 = _flat_featvec(f_fv)
    fk, fd = np.ascontiguousarray(f_keys, KP_DTYPE), np.ascontiguousarray(f_desc, np.uint8)
    out = np.full(max(len(fk), 1), -1, np.int32)
    n = lib().orc_search_by_bow_kf_frame(_p(kk), _p(dk), len(kk), _p(has), len(n1), _p(n1), _p(b1), _p(i1), _p(fk), _p(fd),
                                         len(fk), len(n2), _p(n2), _p(b2), _p(i2), nnratio, int(check_ori), _p(out))
    return n, out[:len(fk)].copy()


def search_by_bow_kf_kf(kf1, kf2, nnratio, check_ori=True):
    args = []
    for kf in (kf1, kf2):
        kk, dk, has = (np.ascontiguousarray(kf["keys_un"], KP_DTYPE), np.ascontiguousarray(kf["desc"], np.uint8),
                       np.ascontiguousarray(kf["has_map_point"], np.uint8))
        nn, bb, ii = _flat_featvec(kf["feat_vec"])
        args.append((kk, dk, has, nn, bb, ii))
    out = np.full(max(len(args[0][0]), 1), -1, np.int32)
    flat = []
    for (kk, dk, has, nn, bb, ii) in args:
        flat += [_p(kk), _p(dk), len(kk), _p(has), len(nn), _p(nn), _p(bb), _p(ii)]
    n = lib().orc_search_by_bow_kf_kf(*flat, nnratio, int(check_ori), _p(out))
    return n, out[:len(args[0][0])].copy()


def search_for_triangulation(kf1, kf2, F12, epipole, only_stereo=False, check_ori=True, fp_mode=FP_GCC_FMA):
    args = []
    for kf in (kf1, kf2):
        kk, dk, has, ur = (np.ascontiguousarray(kf["keys_un"], KP_DTYPE), np.ascontiguousarray(kf["desc"], np.uint8),
                           np.ascontiguousarray(kf["has_map_point"], np.uint8), np.ascontiguousarray(kf["u_right"], np.float32))
        nn, bb, ii = _flat_featvec(kf["feat_vec"])
        args.append((kk, dk, has, ur, nn, bb, ii))
    out = np.full(max(len(args[0][0]), 1), -1, np.int32)
    flat = []
    for (kk, dk, has, ur, nn, bb, ii) in args:
        flat += [_p(kk), _p(dk), len(kk), _p(has), _p(ur), len(nn), _p(nn), _p(bb), _p(ii)]
    F = np.ascontiguousarray(F12, np.float32).reshape(9)
    sf = np.ascontiguousarray(kf2["scale_factors"], np.float32); s2 = np.ascontiguousarray(kf2["level_sigma2"], np.float32)
    n = lib().orc_search_for_triangulation(*flat, _p(F), float(epipole[0]), float(epipole[1]), _p(sf), _p(s2),
                                           int(only_stereo), int(check_ori), fp_mode, _p(out))
    return n, out[:len(args[0][0])].copy()


def _tgt(t):
    return (np.ascontiguousarray(t["keys_un"], KP_DTYPE), np.ascontiguousarray(t["desc"], np.uint8),
            [float(b) for b in t["bounds"]], np.ascontiguousarray(t["scale_factors"], np.float32))


def _pts(p):
    return (np.ascontiguousarray(p["valid"], np.uint8), np.ascontiguousarray(p["uv"], np.float32),
            np.ascontiguousarray(p["level"], np.int32), np.ascontiguousarray(p["desc"], np.uint8))


def fuse(kf, pts, th, fp_mode=FP_GCC_FMA):
    k, d, b, sf = _tgt(kf); v, uv, lv, pd = _pts(pts)
    urk = np.ascontiguousarray(kf["u_right"], np.float32); is2 = np.ascontiguousarray(kf["inv_level_sigma2"], np.float32)
    ur = np.ascontiguousarray(pts["u_right"], np.float32)
    out = np.full(max(len(v), 1), -1, np.int32)
    n = lib().orc_fuse(_p(k), _p(d), _p(urk), len(k), *b, _p(sf), _p(is2), len(v), _p(v), _p(uv), _p(ur), _p(lv), _p(pd), th,
                       fp_mode, _p(out))
    return n, out[:len(v)].copy()


def fuse_sim3(kf, pts, th):
    k, d, b, sf = _tgt(kf); v, uv, lv, pd = _pts(pts)
    out = np.full(max(len(v), 1), -1, np.int32)
    n = lib().orc_fuse_sim3(_p(k), _p(d), len(k), *b, _p(sf), len(v), _p(v), _p(uv), _p(lv), _p(pd), th, _p(out))
    return n, out[:len(v)].copy()


def search_by_projection_sim3(kf, pts, matched, th):
    k, d, b, sf = _tgt(kf); v, uv, lv, pd = _pts(pts)
    out = np.full(max(len(v), 1), -1, np.int32)
    n = lib().orc_search_by_projection_sim3(_p(k), _p(d), len(k), *b, _p(sf), len(v), _p(v), _p(uv), _p(lv), _p(pd), int(th),
                                            _p(matched), _p(out))
    return n, out[:len(v)].copy()


def search_by_sim3(kf1, kf2, p12, p21, th):
    k1, d1, b1, sf1 = _tgt(kf1); k2, d2, b2, sf2 = _tgt(kf2)
    v1, uv1, lv1, pd1 = _pts(p12); v2, uv2, lv2, pd2 = _pts(p21)
    bb1, bb2 = np.array(b1, np.float32), np.array(b2, np.float32)
    out = np.full(max(len(k1), 1), -1, np.int32)
    n = lib().orc_search_by_sim3(_p(k1), _p(d1), len(k1), _p(bb1), _p(sf1), _p(k2), _p(d2), len(k2), _p(bb2), _p(sf2),
                                 _p(v1), _p(uv1), _p(lv1), _p(pd1), _p(v2), _p(uv2), _p(lv2), _p(pd2), th, _p(out))
    return n, out[:len(k1)].copy()


def search_by_projection_kf(cur, pts, cur_has_mp, th, orb_dist, check_ori=True):
    k, d, b, sf = _tgt(cur); v, uv, lv, pd = _pts(pts)
    ang = np.ascontiguousarray(pts["angle"], np.float32)
    out = np.full(max(len(k), 1), -1, np.int32)
    n = lib().orc_search_by_projection_kf(_p(k), _p(d), len(k), *b, _p(sf), len(v), _p(v), _p(uv), _p(lv), _p(pd), _p(ang), th,
                                          int(orb_dist), int(check_ori), _p(cur_has_mp), _p(out))
    return n, out[:len(k)].copy()


def bow_transform(voc, desc, levelsup=4):
    """voc: dict n_nodes, L, child_begin, child_ids, desc, weight, word_id (+ weighting, scoring).  Per-descriptor results
    and the flattened BowVector / FeatureVector: (word_id, weight, node_id, (bow_word, bow_value), (fv_node, fv_begin, fv_index))"""
    d = np.ascontiguousarray(desc, np.uint8); n = len(d)
    cb = np.ascontiguousarray(voc["child_begin"], np.int32); ci_ = np.ascontiguousarray(voc["child_ids"], np.uint32)
    nd = np.ascontiguousarray(voc["desc"], np.uint8); nw = np.ascontiguousarray(voc["weight"], np.float64)
    nwid = np.ascontiguousarray(voc["word_id"], np.uint32)
    wid = np.zeros(max(n, 1), np.uint32); w = np.zeros(max(n, 1), np.float64); nid = np.zeros(max(n, 1), np.uint32)
    lib().orc_bow_transform(int(voc["n_nodes"]), int(voc["L"]), _p(cb), _p(ci_), _p(nd), _p(nw), _p(nwid), _p(d), n, levelsup,
                            _p(wid), _p(w), _p(nid))
    bw = np.zeros(max(n, 1), np.uint32); bv = np.zeros(max(n, 1), np.float64)
    fn = np.zeros(max(n, 1), np.uint32); fb = np.zeros(n + 2, np.int32); fi = np.zeros(max(n, 1), np.uint32)
    nb, nn = C.c_int(0), C.c_int(0)
    lib().orc_bow_vectors(int(voc.get("weighting", 0)), int(voc.get("scoring", 0)), _p(wid), _p(w), _p(nid), n, _p(bw), _p(bv),
                          C.byref(nb), _p(fn), _p(fb), _p(fi), C.byref(nn))
    return (wid[:n].copy(), w[:n].copy(), nid[:n].copy(), (bw[:nb.value].copy(), bv[:nb.value].copy()),
            (fn[:nn.value].copy(), fb[:nn.value + 1].copy(), fi[:fb[nn.value]].copy()))


def undistort_keypoints(kps, K, dist):
    """Frame::UndistortKeyPoints; K = (fx, fy, cx, cy), dist = (k1, k2, p1, p2[, k3])"""
    k = np.ascontiguousarray(kps, KP_DTYPE); d = np.ascontiguousarray(dist, np.float32)
    out = np.zeros(max(len(k), 1), KP_DTYPE)
    lib().orc_undistort_keypoints(_p(k), len(k), *[float(v) for v in K], _p(d), len(d), _p(out))
    return out[:len(k)].copy()


def image_bounds(cols, rows, K, dist):
    d = np.ascontiguousarray(dist, np.float32); b = np.zeros(4, np.float32)
    lib().orc_image_bounds(cols, rows, *[float(v) for v in K], _p(d), len(d), _p(b))
    return b


def remap_linear(img, mapx, mapy):
    """cv::remap(img, ., mapx, mapy, INTER_LINEAR) for an 8-bit image and float32 maps of the output size"""
    img = np.ascontiguousarray(img, np.uint8); mx = np.ascontiguousarray(mapx, np.float32); my = np.ascontiguousarray(mapy, np.float32)
    dh, dw = mx.shape
    out = np.zeros((dh, dw), np.uint8)
    lib().orc_remap_linear_u8(_p(img), img.shape[1], img.shape[0], img.strides[0], _p(mx), _p(my), dw, dh, _p(out), dw)
    return out
