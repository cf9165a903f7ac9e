"""Host-side mirror of ORB_SLAM2::ORBextractor (reference include/ORBextractor.h:82-185) over the C ABI.

Same constructor arguments, same getters, `__call__(image)` = operator()(image, mask, keypoints, descriptors).
All compute happens in the HIP library; this module only marshals buffers.
"""
from __future__ import annotations
import ctypes as C
import numpy as np
from . import _capi
from ._capi import KP_DTYPE, Params, check, ptr, lib


class ORBextractor:
    HARRIS_SCORE, FAST_SCORE = 0, 1  # include/ORBextractor.h:99-103

    def __init__(self, nfeatures=1000, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7, *,
                 fp_mode=_capi.FP_GCC_FMA, device=-1, max_batch=1, max_cand_per_cell=0):
        self._L = lib()
        p = Params()
        self._L.orbx_default_params(C.byref(p))
        p.nfeatures, p.scale_factor, p.nlevels = int(nfeatures), float(scaleFactor), int(nlevels)
        p.ini_th_fast, p.min_th_fast = int(iniThFAST), int(minThFAST)
        p.fp_mode, p.device, p.max_batch, p.max_cand_per_cell = fp_mode, device, max_batch, max_cand_per_cell
        self.params = p
        self._h = C.c_void_p()
        check(self._L.orbx_create(C.byref(p), C.byref(self._h)))
        self.nfeatures, self.nlevels, self.max_batch = int(nfeatures), int(nlevels), int(max_batch)

    def close(self):
        if getattr(self, "_h", None):
            self._L.orbx_destroy(self._h)
            self._h = None

    __del__ = close

    @property
    def handle(self):
        return self._h

    # ---- getters (include/ORBextractor.h:120-170)
    def GetLevels(self):
        return self._L.orbx_get_levels(self._h)

    def GetScaleFactor(self):
        return self._L.orbx_get_scale_factor(self._h)

    def _tables(self):
        n = self.nlevels
        t = [np.zeros(n, np.float32) for _ in range(4)]
        check(self._L.orbx_get_scale_tables(self._h, *[ptr(a) for a in t]))
        return t

    def GetScaleFactors(self):
        return self._tables()[0]

    def GetInverseScaleFactors(self):
        return self._tables()[1]

    def GetScaleSigmaSquares(self):
        return self._tables()[2]

    def GetInverseScaleSigmaSquares(self):
        return self._tables()[3]

    def features_per_level(self):
        a = np.zeros(self.nlevels, np.int32)
        check(self._L.orbx_get_features_per_level(self._h, ptr(a)))
        return a

    def umax(self):
        a = np.zeros(16, np.int32)
        check(self._L.orbx_get_umax(self._h, ptr(a)))
        return a

    def max_keypoints(self, width, height):
        n = self._L.orbx_max_keypoints(self._h, width, height)
        if n < 0:
            raise _capi.OrbxError(-n, self._L.orbx_last_error().decode())
        return n

    def set_input_format(self, fmt):
        """_capi.FMT_*: colour frames are converted on the device exactly like cv::cvtColor in Tracking::GrabImage*"""
        check(self._L.orbx_set_input_format(self._h, int(fmt)))
        self._nch = 1 if fmt == _capi.FMT_GRAY8 else 3 if fmt in (_capi.FMT_RGB8, _capi.FMT_BGR8) else 4

    # ---- operator()
    def __call__(self, image, mask=None):
        """image: HxW uint8.  Returns (keypoints[KP_DTYPE], descriptors[N,32] uint8).
        An empty image returns (None, None): the reference returns silently (src/ORBextractor.cc:1966-1967)."""
        if image is None or image.size == 0:
            return None, None
        img = np.ascontiguousarray(image)
        nch = getattr(self, "_nch", 1)
        assert img.dtype == np.uint8 and (img.ndim == 2 if nch == 1 else (img.ndim == 3 and img.shape[2] == nch)), \
            "image.type() == CV_8UC1 (src/ORBextractor.cc:1972), or the colour format set with set_input_format"
        h, w = img.shape[:2]
        cap = self.max_keypoints(w, h)
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int(0)
        check(self._L.orbx_extract(self._h, ptr(img), w, h, img.strides[0], ptr(kps), ptr(desc), cap, C.byref(n)))
        return kps[:n.value].copy(), desc[:n.value].copy()

    def extract_batch(self, images):
        """images: [B,H,W] uint8 host array -> list of (keypoints, descriptors)"""
        imgs = np.ascontiguousarray(images)
        assert imgs.dtype == np.uint8 and imgs.ndim == 3
        b, h, w = imgs.shape
        cap = self.max_keypoints(w, h)
        kps = np.zeros((b, cap), KP_DTYPE)
        desc = np.zeros((b, cap, 32), np.uint8)
        counts = np.zeros(b, np.int32)
        check(self._L.orbx_extract_batch(self._h, b, ptr(imgs), w, h, imgs.strides[1], imgs.strides[0], ptr(kps),
                                         ptr(desc), ptr(counts), cap))
        return [(kps[i, :counts[i]].copy(), desc[i, :counts[i]].copy()) for i in range(b)]

    def set_rectification(self, map_x=None, map_y=None):
        """cv::remap(., M1, M2, INTER_LINEAR) of the EuRoC driver fused into level 0 (float32 maps of the image size);
        None, None switches it off"""
        if map_x is None:
            check(self._L.orbx_set_rectification(self._h, None, None, 0, 0))
            return
        mx, my = np.ascontiguousarray(map_x, np.float32), np.ascontiguousarray(map_y, np.float32)
        assert mx.shape == my.shape and mx.ndim == 2
        check(self._L.orbx_set_rectification(self._h, ptr(mx), ptr(my), mx.shape[1], mx.shape[0]))

    def extract_batch_device(self, d_imgs, nframes, width, height, stride, frame_stride, d_kps, d_desc, d_counts,
                             d_status, cap):
        """device-pointer entry (torch tensors or raw addresses); asynchronous on the handle's stream"""
        check(self._L.orbx_extract_batch_device(self._h, nframes, ptr(d_imgs), width, height, stride, frame_stride,
                                                ptr(d_kps), ptr(d_desc), ptr(d_counts), ptr(d_status), cap))

    # ---- mvImagePyramid (include/ORBextractor.h:185)
    def pyramid_level(self, level, frame=0, blur=False):
        w, h, p = C.c_int(), C.c_int(), C.c_int()
        check(self._L.orbx_pyramid_level_info(self._h, level, C.byref(w), C.byref(h), C.byref(p)))
        out = np.zeros((h.value, w.value), np.uint8)
        fn = self._L.orbx_debug_blur_copy if blur else self._L.orbx_pyramid_level_copy
        check(fn(self._h, frame, level, ptr(out), w.value))
        return out

    # ---- per-stage inspection (parity tests)
    def debug_candidates(self, level, frame=0, cap=1 << 16):
        out = np.zeros(cap, KP_DTYPE)
        n = C.c_int(0)
        check(self._L.orbx_debug_candidates(self._h, frame, level, ptr(out), cap, C.byref(n)))
        return out[:n.value].copy()

    def debug_level_keypoints(self, level, frame=0, cap=1 << 14):
        out = np.zeros(cap, KP_DTYPE)
        n = C.c_int(0)
        check(self._L.orbx_debug_level_keypoints(self._h, frame, level, ptr(out), cap, C.byref(n)))
        return out[:n.value].copy()

    # ---- stream / timing
    def stream(self):
        return self._L.orbx_get_stream(self._h)

    def set_stream(self, s):
        check(self._L.orbx_set_stream(self._h, C.c_void_p(s) if s else None))

    def synchronize(self):
        check(self._L.orbx_synchronize(self._h))

    def profile_enable(self, mask):
        check(self._L.orbx_profile_enable(self._h, mask))

    def profile_read(self, reset=True):
        ms = np.zeros(_capi.K_COUNT, np.float32)
        n = np.zeros(_capi.K_COUNT, np.int32)
        check(self._L.orbx_profile_read(self._h, ptr(ms), ptr(n), 1 if reset else 0))
        return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(_capi.K_NAMES)}
