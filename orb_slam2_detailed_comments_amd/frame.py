"""Host-side mirror of the parts of ORB_SLAM2::Frame that sit on the hot path (reference include/Frame.h,
src/Frame.cc): the 64x48 feature grid (AssignFeaturesToGrid :432-460, GetFeaturesInArea :633-717) and
ComputeStereoMatches (:880-1176).  Plain arrays instead of cv::Mat / std::vector; compute goes through the C ABI."""
from __future__ import annotations
import ctypes as C
import numpy as np
from . import _capi
from ._capi import KP_DTYPE, check, ptr, lib

FRAME_GRID_ROWS, FRAME_GRID_COLS = 48, 64  # include/Frame.h:54,59


class Frame:
    def __init__(self, keys, descriptors, width, height, bounds=None):
        """keys: KP_DTYPE array (mvKeys == mvKeysUn: no distortion), descriptors [N,32] uint8,
        bounds = (mnMinX, mnMaxX, mnMinY, mnMaxY), default (0, width, 0, height) (src/Frame.cc:863-866)"""
        self.mvKeys = np.ascontiguousarray(keys, KP_DTYPE)
        self.mvKeysUn = self.mvKeys
        self.mDescriptors = np.ascontiguousarray(descriptors, np.uint8)
        self.N = len(self.mvKeys)
        self._wh = (int(width), int(height))
        self.bounds = tuple(float(b) for b in (bounds or (0, width, 0, height)))
        self.mvuRight = np.full(self.N, -1.0, np.float32)
        self.mvDepth = np.full(self.N, -1.0, np.float32)
        self._grid = None

    def UndistortKeyPoints(self, extractor, K, dist):
        """Frame::UndistortKeyPoints + ComputeImageBounds (src/Frame.cc:770-865): K = (fx, fy, cx, cy), dist = mDistCoef.
        Sets mvKeysUn and the grid bounds (the reference computes the bounds once per camera, :115-124)."""
        k4 = np.ascontiguousarray(K, np.float32); d = np.ascontiguousarray(dist, np.float32)
        out = np.zeros(max(self.N, 1), KP_DTYPE)
        check(lib().orbx_undistort_keypoints(extractor.handle, ptr(self.mvKeys), self.N, ptr(k4), ptr(d), len(d), ptr(out)))
        self.mvKeysUn = out[:self.N].copy()
        b = np.zeros(4, np.float32)
        check(lib().orbx_image_bounds(extractor.handle, self._wh[0], self._wh[1], ptr(k4), ptr(d), len(d), ptr(b)))
        self.bounds = tuple(float(v) for v in b)
        self._free()
        return self.mvKeysUn

    def AssignFeaturesToGrid(self):
        self._free()
        self._grid = lib().orbx_grid_create(ptr(self.mvKeysUn), self.N, *self.bounds)
        if not self._grid:
            raise _capi.OrbxError(_capi.BAD_ARGUMENT, lib().orbx_last_error().decode())

    def GetFeaturesInArea(self, x, y, r, minLevel=-1, maxLevel=-1):
        if self._grid is None:
            self.AssignFeaturesToGrid()
        out = np.zeros(max(self.N, 1), np.int32)
        n = lib().orbx_grid_query(self._grid, x, y, r, minLevel, maxLevel, ptr(out), len(out))
        return out[:n].copy()

    def ComputeStereoMatches(self, right, extractor_left, extractor_right, mb, mbf, frame_left=0, frame_right=0):
        """right: Frame of the right image; the two extractors must still hold the pyramids of these frames"""
        n = C.c_int(0)
        check(lib().orbx_stereo_match(extractor_left.handle, extractor_right.handle, frame_left, frame_right,
                                      ptr(self.mvKeys), ptr(self.mDescriptors), self.N, ptr(right.mvKeys),
                                      ptr(right.mDescriptors), right.N, mb, mbf, ptr(self.mvuRight), ptr(self.mvDepth),
                                      C.byref(n)))
        return n.value

    def _free(self):
        if getattr(self, "_grid", None):
            lib().orbx_grid_destroy(self._grid)
            self._grid = None

    __del__ = _free
