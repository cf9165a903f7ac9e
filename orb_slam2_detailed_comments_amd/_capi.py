"""ctypes binding of liborbx.so (include/orbx.h).  This is the ONLY way Python reaches the HIP path;
there is no CPU fallback: a missing library or a missing GPU raises."""
from __future__ import annotations
import ctypes as C
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# ORBX_LIB: another build of the same library (kernel-variant sweeps in tools/); never a different implementation
LIB_PATH = os.environ.get("ORBX_LIB") or os.path.join(HERE, "lib", "liborbx.so")

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28

OK, EMPTY_IMAGE, BAD_ARGUMENT, BAD_ASPECT, CAPACITY, HIP_ERROR, NO_DEVICE, UNSUPPORTED = range(8)
FP_GCC_FMA, FP_STRICT = 0, 1
FMT_GRAY8, FMT_RGB8, FMT_BGR8, FMT_RGBA8, FMT_BGRA8 = range(5)
K_NAMES = ("k_pyr_l0", "k_pyr_resize", "k_fast_rows", "k_quadtree", "k_orient", "k_blur", "k_describe",
           "k_match", "misc")
K_COUNT = len(K_NAMES)


class Params(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
                ("ini_th_fast", C.c_int32), ("min_th_fast", C.c_int32), ("pyramid_mode", C.c_int32),
                ("fp_mode", C.c_int32), ("device", C.c_int32), ("max_batch", C.c_int32),
                ("max_cand_per_cell", C.c_int32)]


class FrameView(C.Structure):
    _fields_ = [("keys_un", C.c_void_p), ("desc", C.c_void_p), ("u_right", C.c_void_p), ("n", C.c_int32),
                ("Tcw", C.c_float * 16), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("min_x", C.c_float), ("max_x", C.c_float), ("min_y", C.c_float), ("max_y", C.c_float),
                ("mb", C.c_float), ("mbf", C.c_float)]


class LastFrameView(C.Structure):
    _fields_ = [("keys_un", C.c_void_p), ("n", C.c_int32), ("has_map_point", C.c_void_p), ("world_pos", C.c_void_p),
                ("mp_desc", C.c_void_p), ("observations", C.c_void_p), ("Tcw", C.c_float * 16)]


class MapPointView(C.Structure):
    _fields_ = [("n", C.c_int32), ("in_view", C.c_void_p), ("proj", C.c_void_p), ("level", C.c_void_p),
                ("view_cos", C.c_void_p), ("desc", C.c_void_p), ("observations", C.c_void_p)]


class FeatVecView(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("node_id", C.c_void_p), ("begin", C.c_void_p), ("index", C.c_void_p)]


class KeyFrameView(C.Structure):
    _fields_ = [("keys_un", C.c_void_p), ("desc", C.c_void_p), ("n", C.c_int32), ("has_map_point", C.c_void_p),
                ("u_right", C.c_void_p), ("feat_vec", FeatVecView), ("scale_factors", C.c_void_p),
                ("level_sigma2", C.c_void_p)]


class ProjectedPoints(C.Structure):
    _fields_ = [("n", C.c_int32), ("valid", C.c_void_p), ("uv", C.c_void_p), ("u_right", C.c_void_p), ("level", C.c_void_p),
                ("desc", C.c_void_p), ("angle", C.c_void_p)]


class TargetView(C.Structure):
    _fields_ = [("keys_un", C.c_void_p), ("desc", C.c_void_p), ("u_right", C.c_void_p), ("n", C.c_int32),
                ("min_x", C.c_float), ("max_x", C.c_float), ("min_y", C.c_float), ("max_y", C.c_float),
                ("scale_factors", C.c_void_p), ("inv_level_sigma2", C.c_void_p)]


class VocabularyView(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("k", C.c_int32), ("L", C.c_int32), ("weighting", C.c_int32), ("scoring", C.c_int32),
                ("child_begin", C.c_void_p), ("child_ids", C.c_void_p), ("desc", C.c_void_p), ("weight", C.c_void_p),
                ("word_id", C.c_void_p)]


class OrbxError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"orbx status {status}: {msg}")
        self.status = status


# every symbol include/orbx.h declares (tests/test_abi.py checks the header against this list)
SYMBOLS = [
    "orbx_create", "orbx_destroy", "orbx_default_params", "orbx_last_error", "orbx_status_string",
    "orbx_abi_version", "orbx_get_levels", "orbx_get_scale_factor", "orbx_get_scale_tables",
    "orbx_get_features_per_level", "orbx_get_umax", "orbx_max_keypoints", "orbx_extract", "orbx_extract_batch",
    "orbx_extract_batch_device", "orbx_pyramid_level_info", "orbx_pyramid_level_device",
    "orbx_pyramid_level_copy", "orbx_match_bruteforce_device", "orbx_match_bruteforce", "orbx_hamming_matrix",
    "orbx_get_stream", "orbx_set_stream", "orbx_synchronize", "orbx_profile_enable", "orbx_profile_read",
    "orbx_kernel_name", "orbx_debug_candidates", "orbx_debug_level_keypoints", "orbx_debug_blur_copy",
    "orbx_grid_create", "orbx_grid_destroy", "orbx_grid_query", "orbx_three_maxima",
    "orbx_search_for_initialization", "orbx_stereo_match", "orbx_search_by_projection_frame",
    "orbx_search_by_projection_mappoints", "orbx_set_input_format", "orbx_search_by_bow_keyframe_frame",
    "orbx_search_by_bow_keyframes", "orbx_search_for_triangulation", "orbx_triangulation_batch_create", "orbx_triangulation_batch_select", "orbx_triangulation_batch_destroy", "orbx_fuse", "orbx_fuse_sim3", "orbx_fuse_batch", "orbx_fuse_sim3_batch",
    "orbx_search_by_projection_sim3", "orbx_search_by_sim3", "orbx_search_by_projection_keyframe",
    "orbx_stereo_match_batch_device", "orbx_host_alloc", "orbx_host_free", "orbx_set_rectification", "orbx_undistort_keypoints_device",
    "orbx_grid_build_device", "orbx_gated_candidates",
    "orbx_undistort_keypoints", "orbx_image_bounds", "orbx_vocabulary_create", "orbx_vocabulary_destroy", "orbx_bow_transform", "orbx_bow_transform_device", "orbx_bow_vectors",
]

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OrbxError(NO_DEVICE, f"{LIB_PATH} is missing: build it with "
                        "`python -m orb_slam2_detailed_comments_amd.build` (hipcc, gfx950); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_float
    L.orbx_create.restype = i32; L.orbx_create.argtypes = [C.POINTER(Params), C.POINTER(vp)]
    L.orbx_destroy.restype = None; L.orbx_destroy.argtypes = [vp]
    L.orbx_default_params.restype = None; L.orbx_default_params.argtypes = [C.POINTER(Params)]
    L.orbx_last_error.restype = C.c_char_p; L.orbx_last_error.argtypes = []
    L.orbx_status_string.restype = C.c_char_p; L.orbx_status_string.argtypes = [i32]
    L.orbx_abi_version.restype = i32
    L.orbx_get_levels.restype = i32; L.orbx_get_levels.argtypes = [vp]
    L.orbx_get_scale_factor.restype = f32; L.orbx_get_scale_factor.argtypes = [vp]
    L.orbx_get_scale_tables.restype = i32; L.orbx_get_scale_tables.argtypes = [vp, vp, vp, vp, vp]
    L.orbx_get_features_per_level.restype = i32; L.orbx_get_features_per_level.argtypes = [vp, vp]
    L.orbx_get_umax.restype = i32; L.orbx_get_umax.argtypes = [vp, vp]
    L.orbx_max_keypoints.restype = i32; L.orbx_max_keypoints.argtypes = [vp, i32, i32]
    L.orbx_extract.restype = i32; L.orbx_extract.argtypes = [vp, vp, i32, i32, i32, vp, vp, i32, C.POINTER(i32)]
    L.orbx_extract_batch.restype = i32
    L.orbx_extract_batch.argtypes = [vp, i32, vp, i32, i32, i32, i64, vp, vp, vp, i32]
    L.orbx_extract_batch_device.restype = i32
    L.orbx_extract_batch_device.argtypes = [vp, i32, vp, i32, i32, i32, i64, vp, vp, vp, vp, i32]
    L.orbx_pyramid_level_info.restype = i32
    L.orbx_pyramid_level_info.argtypes = [vp, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    L.orbx_pyramid_level_device.restype = i32; L.orbx_pyramid_level_device.argtypes = [vp, i32, i32, C.POINTER(vp)]
    L.orbx_pyramid_level_copy.restype = i32; L.orbx_pyramid_level_copy.argtypes = [vp, i32, i32, vp, i32]
    L.orbx_match_bruteforce_device.restype = i32
    L.orbx_match_bruteforce_device.argtypes = [vp, i32, vp, vp, i64, vp, vp, i64, vp, vp, vp, i32]
    L.orbx_match_bruteforce.restype = i32; L.orbx_match_bruteforce.argtypes = [vp, vp, i32, vp, i32, vp, vp, vp]
    L.orbx_hamming_matrix.restype = i32; L.orbx_hamming_matrix.argtypes = [vp, vp, i32, vp, i32, vp]
    L.orbx_get_stream.restype = vp; L.orbx_get_stream.argtypes = [vp]
    L.orbx_set_stream.restype = i32; L.orbx_set_stream.argtypes = [vp, vp]
    L.orbx_synchronize.restype = i32; L.orbx_synchronize.argtypes = [vp]
    L.orbx_profile_enable.restype = i32; L.orbx_profile_enable.argtypes = [vp, C.c_uint32]
    L.orbx_profile_read.restype = i32; L.orbx_profile_read.argtypes = [vp, vp, vp, i32]
    L.orbx_kernel_name.restype = C.c_char_p; L.orbx_kernel_name.argtypes = [i32]
    L.orbx_debug_candidates.restype = i32; L.orbx_debug_candidates.argtypes = [vp, i32, i32, vp, i32, C.POINTER(i32)]
    L.orbx_debug_level_keypoints.restype = i32
    L.orbx_debug_level_keypoints.argtypes = [vp, i32, i32, vp, i32, C.POINTER(i32)]
    L.orbx_debug_blur_copy.restype = i32; L.orbx_debug_blur_copy.argtypes = [vp, i32, i32, vp, i32]
    L.orbx_grid_create.restype = vp; L.orbx_grid_create.argtypes = [vp, i32, f32, f32, f32, f32]
    L.orbx_grid_destroy.restype = None; L.orbx_grid_destroy.argtypes = [vp]
    L.orbx_grid_query.restype = i32; L.orbx_grid_query.argtypes = [vp, f32, f32, f32, i32, i32, vp, i32]
    L.orbx_three_maxima.restype = None
    L.orbx_three_maxima.argtypes = [vp, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    L.orbx_search_for_initialization.restype = i32
    L.orbx_search_for_initialization.argtypes = [vp, vp, vp, i32, vp, vp, i32, vp, vp, i32, f32, i32, vp, C.POINTER(i32)]
    L.orbx_stereo_match.restype = i32
    L.orbx_stereo_match.argtypes = [vp, vp, i32, i32, vp, vp, i32, vp, vp, i32, f32, f32, vp, vp, C.POINTER(i32)]
    L.orbx_search_by_projection_frame.restype = i32
    L.orbx_search_by_projection_frame.argtypes = [vp, C.POINTER(FrameView), C.POINTER(LastFrameView), f32, i32, i32, vp,
                                                  C.POINTER(i32)]
    L.orbx_search_by_projection_mappoints.restype = i32
    L.orbx_search_by_projection_mappoints.argtypes = [vp, C.POINTER(FrameView), vp, C.POINTER(MapPointView), f32, f32, vp,
                                                      C.POINTER(i32)]
    L.orbx_set_input_format.restype = i32; L.orbx_set_input_format.argtypes = [vp, i32]
    L.orbx_search_by_bow_keyframe_frame.restype = i32
    L.orbx_search_by_bow_keyframe_frame.argtypes = [vp, C.POINTER(KeyFrameView), vp, vp, i32, C.POINTER(FeatVecView), f32,
                                                    i32, vp, C.POINTER(i32)]
    L.orbx_search_by_bow_keyframes.restype = i32
    L.orbx_search_by_bow_keyframes.argtypes = [vp, C.POINTER(KeyFrameView), C.POINTER(KeyFrameView), f32, i32, vp,
                                               C.POINTER(i32)]
    L.orbx_search_for_triangulation.restype = i32
    L.orbx_search_for_triangulation.argtypes = [vp, C.POINTER(KeyFrameView), C.POINTER(KeyFrameView), vp, f32, f32, i32,
                                                i32, vp, C.POINTER(i32)]
    L.orbx_triangulation_batch_create.restype = i32
    L.orbx_triangulation_batch_create.argtypes = [vp, C.POINTER(KeyFrameView), i32, vp, C.POINTER(vp)]
    L.orbx_triangulation_batch_select.restype = i32
    L.orbx_triangulation_batch_select.argtypes = [vp, i32, C.POINTER(KeyFrameView), C.POINTER(KeyFrameView), vp, f32, f32, i32, i32, vp, C.POINTER(i32)]
    L.orbx_triangulation_batch_destroy.restype = None; L.orbx_triangulation_batch_destroy.argtypes = [vp]
    PT, TV = C.POINTER(ProjectedPoints), C.POINTER(TargetView)
    L.orbx_fuse.restype = i32; L.orbx_fuse.argtypes = [vp, TV, PT, f32, vp, C.POINTER(i32)]
    L.orbx_fuse_sim3.restype = i32; L.orbx_fuse_sim3.argtypes = [vp, TV, PT, f32, vp, C.POINTER(i32)]
    for fn in (L.orbx_fuse_batch, L.orbx_fuse_sim3_batch):   # arrays of pointers to the views / output arrays
        fn.restype = i32; fn.argtypes = [vp, i32, vp, vp, f32, vp, vp]
    L.orbx_search_by_projection_sim3.restype = i32
    L.orbx_search_by_projection_sim3.argtypes = [vp, TV, PT, i32, vp, vp, C.POINTER(i32)]
    L.orbx_search_by_sim3.restype = i32; L.orbx_search_by_sim3.argtypes = [vp, TV, TV, PT, PT, f32, vp, C.POINTER(i32)]
    L.orbx_search_by_projection_keyframe.restype = i32
    L.orbx_search_by_projection_keyframe.argtypes = [vp, TV, PT, f32, i32, i32, vp, vp, C.POINTER(i32)]
    L.orbx_vocabulary_create.restype = i32; L.orbx_vocabulary_create.argtypes = [vp, C.POINTER(VocabularyView), C.POINTER(vp)]
    L.orbx_vocabulary_destroy.restype = None; L.orbx_vocabulary_destroy.argtypes = [vp]
    L.orbx_bow_transform.restype = i32; L.orbx_bow_transform.argtypes = [vp, vp, vp, i32, i32, vp, vp, vp]
    L.orbx_bow_transform_device.restype = i32
    L.orbx_bow_transform_device.argtypes = [vp, vp, i32, vp, vp, i64, i32, i32, vp, vp, i32]
    L.orbx_bow_vectors.restype = i32
    L.orbx_bow_vectors.argtypes = [vp, vp, vp, vp, i32, vp, vp, C.POINTER(i32), vp, vp, vp, C.POINTER(i32)]
    L.orbx_stereo_match_batch_device.restype = i32
    L.orbx_stereo_match_batch_device.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp, vp, i32, f32, f32, vp, vp, vp]
    L.orbx_undistort_keypoints_device.restype = i32
    L.orbx_undistort_keypoints_device.argtypes = [vp, i32, vp, vp, i32, vp, vp, i32, vp]
    L.orbx_grid_build_device.restype = i32
    L.orbx_grid_build_device.argtypes = [vp, i32, vp, vp, i32, vp, vp, vp]
    L.orbx_gated_candidates.restype = i32
    L.orbx_gated_candidates.argtypes = [vp, vp, vp, i32, vp, vp, vp, vp, i32, vp, vp, i32, vp]
    L.orbx_undistort_keypoints.restype = i32; L.orbx_undistort_keypoints.argtypes = [vp, vp, i32, vp, vp, i32, vp]
    L.orbx_image_bounds.restype = i32; L.orbx_image_bounds.argtypes = [vp, i32, i32, vp, vp, i32, vp]
    L.orbx_set_rectification.restype = i32; L.orbx_set_rectification.argtypes = [vp, vp, vp, i32, i32]
    L.orbx_host_alloc.restype = vp; L.orbx_host_alloc.argtypes = [C.c_size_t]
    L.orbx_host_free.restype = None; L.orbx_host_free.argtypes = [vp]
    _lib = L
    return L


def check(status, allow=()):
    if status != OK and status not in allow:
        raise OrbxError(status, lib().orbx_last_error().decode(errors="replace"))
    return status


def ptr(a):
    """numpy array -> void*, torch tensor / int -> device address"""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    if isinstance(a, int):
        return C.c_void_p(a)
    return C.c_void_p(a.data_ptr())


class PinnedArray:
    """numpy view of page-locked host memory (orbx_host_alloc); keep the object alive as long as the array is used"""

    def __init__(self, shape, dtype=np.uint8):
        self.nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self._p = lib().orbx_host_alloc(self.nbytes)
        if not self._p:
            raise MemoryError("orbx_host_alloc failed")
        self.array = np.frombuffer((C.c_uint8 * self.nbytes).from_address(self._p), dtype=dtype).reshape(shape)

    def __del__(self):
        if getattr(self, "_p", None):
            try:
                lib().orbx_host_free(self._p)
            except Exception:
                pass
            self._p = None
