"""ctypes binding of liborbgpu.so and host-side mirrors of the reference classes.

`ORBextractor` and `ORBmatcher` keep the names, argument meaning and return
conventions of the reference's C++ classes (include/ORBextractor.h:44-113,
include/ORBmatcher.h:41-103 under /root/reference/src/ORB_SLAM2/), so parity
tests read like calls into the reference.  All compute happens in the HIP
library; there is no CPU fallback -- constructing either class without the
built library or without a GPU raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liborbgpu.so")

KEYPOINT_DTYPE = np.dtype(
    [("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
     ("octave", "<i4"), ("class_id", "<i4")]
)
WQUERY_DTYPE = np.dtype(
    [("u", "<f4"), ("v", "<f4"), ("r", "<f4"), ("minLevel", "<i4"), ("maxLevel", "<i4"),
     ("angle", "<f4"), ("ur", "<f4"), ("flags", "<i4")]
)

TH_HIGH, TH_LOW, HISTO_LENGTH = 100, 50, 30
MODE_LOCAL_MAP, MODE_LAST_FRAME, MODE_KEYFRAME = 0, 1, 2

# every symbol include/orbgpu.h declares
EXPORTS = [
    "orbgpu_last_error", "orbgpu_device_count", "orbgpu_pinned_alloc", "orbgpu_pinned_free",
    "orbgpu_extractor_create", "orbgpu_extractor_destroy", "orbgpu_extractor_set_stream", "orbgpu_extractor_sync",
    "orbgpu_extractor_status", "orbgpu_extractor_profile", "orbgpu_extractor_stage_times",
    "orbgpu_extractor_tables", "orbgpu_extractor_max_keypoints", "orbgpu_extract", "orbgpu_extract_batch",
    "orbgpu_extract_batch_device", "orbgpu_pyramid_level", "orbgpu_pyramid_level_device",
    "orbgpu_debug_level_candidates", "orbgpu_debug_level_selected",
    "orbgpu_matcher_create", "orbgpu_matcher_destroy", "orbgpu_matcher_set_stream", "orbgpu_matcher_sync", "orbgpu_matcher_set_grid_geometry",
    "orbgpu_descriptor_distance", "orbgpu_hamming_bruteforce", "orbgpu_hamming_bruteforce_device",
    "orbgpu_hamming_bruteforce_device_noalloc", "orbgpu_matcher_set_bruteforce_kernel",
    "orbgpu_search_by_projection", "orbgpu_debug_window_stats", "orbgpu_search_by_projection_batch_device", "orbgpu_make_shift_queries_device",
    "orbgpu_search_by_bow", "orbgpu_search_for_triangulation", "orbgpu_search_in_area", "orbgpu_assign_features_to_grid",
    "orbgpu_distinctive_descriptors", "orbgpu_stereo_from_rgbd", "orbgpu_undistort_keypoints", "orbgpu_search_for_initialization", "orbgpu_stereo_match",
    "orbgpu_stereo_match_batch_device", "orbgpu_search_for_initialization_batch_device",
    "orbgpu_frame_create", "orbgpu_frame_destroy", "orbgpu_frame_build_mono", "orbgpu_frame_build_rgbd", "orbgpu_frame_build_stereo", "orbgpu_frame_upload",
    "orbgpu_frame_size", "orbgpu_frame_download", "orbgpu_frame_grid", "orbgpu_frame_device",
    "orbgpu_search_by_projection_frame", "orbgpu_search_in_area_frame", "orbgpu_search_by_bow_frame",
    "orbgpu_search_for_triangulation_frame", "orbgpu_search_for_initialization_frames",
    "orbgpu_vocabulary_create", "orbgpu_vocabulary_load_text", "orbgpu_vocabulary_destroy", "orbgpu_vocabulary_set_stream",
    "orbgpu_vocabulary_sync", "orbgpu_vocabulary_info", "orbgpu_vocabulary_transform", "orbgpu_vocabulary_transform_batch_device",
]


class OrbGpuError(RuntimeError):
    pass


class Params(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scaleFactor", C.c_float), ("nlevels", C.c_int32),
                ("iniThFAST", C.c_int32), ("minThFAST", C.c_int32), ("gauss_variant", C.c_int32),
                ("device", C.c_int32), ("max_width", C.c_int32), ("max_height", C.c_int32),
                ("max_batch", C.c_int32)]


_lib = None


def load_library():
    """Load liborbgpu.so (fails loudly when it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OrbGpuError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(or make -C orb_slam2_amd/csrc)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.orbgpu_last_error.restype = C.c_char_p
        vp, i32, f32 = C.c_void_p, C.c_int32, C.c_float
        L.orbgpu_pinned_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
        L.orbgpu_pinned_free.argtypes = [vp]
        L.orbgpu_pinned_free.restype = None
        L.orbgpu_extractor_create.argtypes = [C.POINTER(Params), C.POINTER(vp)]
        L.orbgpu_extractor_destroy.argtypes = [vp]
        L.orbgpu_extractor_destroy.restype = None
        L.orbgpu_extractor_set_stream.argtypes = [vp, vp]
        L.orbgpu_extractor_sync.argtypes = [vp]
        L.orbgpu_extractor_status.argtypes = [vp, C.POINTER(i32)]
        L.orbgpu_extractor_profile.argtypes = [vp, i32]
        L.orbgpu_extractor_stage_times.argtypes = [vp, vp, C.POINTER(i32)]
        L.orbgpu_extractor_tables.argtypes = [vp, C.POINTER(i32), vp, vp, vp, vp, vp]
        L.orbgpu_extractor_max_keypoints.argtypes = [vp, C.POINTER(i32)]
        L.orbgpu_extract.argtypes = [vp, vp, i32, i32, i32, vp, vp, i32, C.POINTER(i32)]
        L.orbgpu_extract_batch.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, i32, vp]
        L.orbgpu_extract_batch_device.argtypes = [vp, vp, i32, i32, i32, i32, C.c_size_t, vp, vp, i32, vp]
        L.orbgpu_pyramid_level.argtypes = [vp, i32, i32, i32, i32, vp, i32, C.POINTER(i32), C.POINTER(i32)]
        L.orbgpu_pyramid_level_device.argtypes = [vp, i32, i32, i32, C.POINTER(vp), C.POINTER(i32), C.POINTER(i32),
                                                  C.POINTER(i32)]
        L.orbgpu_debug_level_candidates.argtypes = [vp, i32, i32, vp, vp, vp, i32, C.POINTER(i32)]
        L.orbgpu_debug_level_selected.argtypes = [vp, i32, i32, vp, vp, vp, i32, C.POINTER(i32)]
        L.orbgpu_matcher_create.argtypes = [i32, C.POINTER(vp)]
        L.orbgpu_matcher_destroy.argtypes = [vp]
        L.orbgpu_matcher_destroy.restype = None
        L.orbgpu_matcher_set_stream.argtypes = [vp, vp]
        L.orbgpu_matcher_sync.argtypes = [vp]
        L.orbgpu_matcher_set_grid_geometry.argtypes = [vp, vp]
        L.orbgpu_descriptor_distance.argtypes = [vp, vp]
        L.orbgpu_hamming_bruteforce.argtypes = [vp, vp, i32, vp, i32, vp, vp, vp]
        L.orbgpu_hamming_bruteforce_device.argtypes = [vp, vp, i32, vp, i32, vp, vp, vp]
        L.orbgpu_hamming_bruteforce_device_noalloc.argtypes = [vp, vp, i32, vp, i32, vp, vp, vp]
        L.orbgpu_matcher_set_bruteforce_kernel.argtypes = [vp, i32]
        L.orbgpu_search_by_projection.argtypes = [vp, i32, vp, vp, i32, vp, vp, vp, i32, vp, vp, vp, i32, f32, i32,
                                                  C.POINTER(i32)]
        L.orbgpu_search_by_projection_batch_device.argtypes = [vp, i32, i32, vp, vp, vp, i32, vp, vp, vp, vp, i32, vp,
                                                               vp, vp, i32, f32, i32, vp]
        L.orbgpu_debug_window_stats.argtypes = [vp, i32, C.POINTER(i32), C.POINTER(i32)]
        L.orbgpu_make_shift_queries_device.argtypes = [vp, i32, vp, vp, i32, f32, f32, f32, vp, i32, vp, vp]
        L.orbgpu_search_by_bow.argtypes = [vp, vp, vp, i32, vp, vp, i32, vp, vp, i32, vp, i32, f32, i32, vp, vp, C.POINTER(i32)]
        L.orbgpu_search_for_triangulation.argtypes = [vp, vp, vp, i32, vp, vp, vp, i32, vp, vp, i32, vp, vp, f32, f32, vp, vp, i32,
                                                      i32, i32, vp, C.POINTER(i32)]
        L.orbgpu_search_in_area.argtypes = [vp, vp, vp, i32, vp, vp, vp, i32, vp, vp, i32, i32, i32, vp, C.POINTER(i32)]
        L.orbgpu_assign_features_to_grid.argtypes = [vp, vp, i32, vp, vp, vp, C.POINTER(i32)]
        L.orbgpu_distinctive_descriptors.argtypes = [vp, vp, vp, i32, vp, vp]
        L.orbgpu_stereo_from_rgbd.argtypes = [vp, vp, vp, i32, vp, i32, i32, i32, f32, vp, vp]
        L.orbgpu_undistort_keypoints.argtypes = [vp, vp, i32, vp, vp, vp]
        L.orbgpu_search_for_initialization.argtypes = [vp, vp, vp, i32, vp, vp, i32, vp, vp, vp, i32, f32, i32,
                                                       C.POINTER(i32)]
        L.orbgpu_stereo_match.argtypes = [vp, vp, vp, i32, vp, vp, i32, vp, vp, i32, f32, f32, vp, vp, C.POINTER(i32)]
        L.orbgpu_search_for_initialization_batch_device.argtypes = [vp, i32, vp, vp, vp, i32, vp, vp, vp, i32, vp, i32, f32, i32, vp, vp]
        L.orbgpu_stereo_match_batch_device.argtypes = [vp, vp, vp, i32, vp, vp, vp, i32, vp, vp, vp, i32, f32, f32, vp, vp, vp]
        L.orbgpu_frame_create.argtypes = [i32, i32, C.POINTER(vp)]
        L.orbgpu_frame_destroy.argtypes = [vp]
        L.orbgpu_frame_destroy.restype = None
        L.orbgpu_frame_build_mono.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp, vp, C.POINTER(i32)]
        L.orbgpu_frame_build_rgbd.argtypes = [vp, vp, vp, i32, i32, i32, vp, i32, f32, vp, vp, vp, C.POINTER(i32)]
        L.orbgpu_frame_build_stereo.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, f32, f32, vp, vp, vp, C.POINTER(i32),
                                                C.POINTER(i32)]
        L.orbgpu_frame_upload.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, vp]
        L.orbgpu_frame_size.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
        L.orbgpu_frame_download.argtypes = [vp, vp, vp, vp, vp, vp]
        L.orbgpu_frame_grid.argtypes = [vp, vp, vp, C.POINTER(i32)]
        L.orbgpu_frame_device.argtypes = [vp] + [C.POINTER(vp)] * 6
        L.orbgpu_search_by_projection_frame.argtypes = [vp, i32, vp, vp, i32, vp, vp, vp, i32, f32, i32, C.POINTER(i32)]
        L.orbgpu_search_in_area_frame.argtypes = [vp, vp, vp, i32, vp, vp, i32, i32, i32, vp, C.POINTER(i32)]
        L.orbgpu_search_by_bow_frame.argtypes = [vp, vp, vp, i32, vp, vp, vp, i32, vp, i32, f32, i32, vp, vp, C.POINTER(i32)]
        L.orbgpu_search_for_triangulation_frame.argtypes = [vp, vp, vp, i32, vp, i32, vp, vp, i32, vp, vp, f32, f32, vp, vp, i32,
                                                            i32, i32, vp, C.POINTER(i32)]
        L.orbgpu_search_for_initialization_frames.argtypes = [vp, vp, vp, vp, vp, i32, f32, i32, C.POINTER(i32)]
        L.orbgpu_vocabulary_create.argtypes = [i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, C.POINTER(vp)]
        L.orbgpu_vocabulary_load_text.argtypes = [i32, C.c_char_p, C.POINTER(vp)]
        L.orbgpu_vocabulary_destroy.argtypes = [vp]
        L.orbgpu_vocabulary_destroy.restype = None
        L.orbgpu_vocabulary_set_stream.argtypes = [vp, vp]
        L.orbgpu_vocabulary_sync.argtypes = [vp]
        L.orbgpu_vocabulary_info.argtypes = [vp] + [C.POINTER(i32)] * 6
        L.orbgpu_vocabulary_transform.argtypes = [vp, vp, i32, i32, vp, vp, vp, vp, C.POINTER(i32), vp, vp, vp, C.POINTER(i32)]
        L.orbgpu_vocabulary_transform_batch_device.argtypes = [vp, vp, vp, i32, i32, i32] + [vp] * 9
        _lib = L
    return _lib


def _check(rc, what):
    if rc != 0:
        msg = load_library().orbgpu_last_error().decode("utf-8", "replace")
        raise OrbGpuError("%s failed (%d): %s" % (what, rc, msg))


def _p(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


class _Pinned:
    def __init__(self, ptr):
        self.ptr = ptr

    def __del__(self):
        try:
            load_library().orbgpu_pinned_free(self.ptr)
        except Exception:
            pass


def pinned_empty(shape, dtype):
    """A numpy array in page-locked host memory (orbgpu_pinned_alloc); freed with the array."""
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) * dt.itemsize
    p = C.c_void_p()
    _check(load_library().orbgpu_pinned_alloc(max(n, 1), C.byref(p)), "orbgpu_pinned_alloc")
    buf = (C.c_uint8 * max(n, 1)).from_address(p.value)
    buf._owner = _Pinned(p)  # the array keeps the ctypes buffer alive, the buffer keeps the allocation
    return np.frombuffer(buf, dtype=dt, count=int(np.prod(shape))).reshape(shape)


def device_count():
    n = C.c_int(0)
    load_library().orbgpu_device_count(C.byref(n))
    return n.value


class ORBextractor:
    """Mirror of ORB_SLAM2::ORBextractor (include/ORBextractor.h:44-113)."""

    def __init__(self, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, device=0, max_width=640,
                 max_height=480, max_batch=1, gauss_variant=0):
        self._lib = load_library()
        self._h = C.c_void_p()
        self.params = Params(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, gauss_variant, device, max_width,
                             max_height, max_batch)
        _check(self._lib.orbgpu_extractor_create(C.byref(self.params), C.byref(self._h)), "orbgpu_extractor_create")
        L = nlevels
        self._scale = np.zeros(L, np.float32); self._inv = np.zeros(L, np.float32)
        self._s2 = np.zeros(L, np.float32); self._is2 = np.zeros(L, np.float32)
        self._quota = np.zeros(L, np.int32)
        n = C.c_int32()
        _check(self._lib.orbgpu_extractor_tables(self._h, C.byref(n), _p(self._scale), _p(self._inv), _p(self._s2),
                                                 _p(self._is2), _p(self._quota)), "orbgpu_extractor_tables")
        mk = C.c_int32()
        _check(self._lib.orbgpu_extractor_max_keypoints(self._h, C.byref(mk)), "orbgpu_extractor_max_keypoints")
        self.max_keypoints = mk.value
        self.nlevels = L
        self.mvImagePyramid = _PyramidView(self)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.orbgpu_extractor_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- operator() ------------------------------------------------------
    def __call__(self, image, mask=None):
        """(keypoints, descriptors) of one uint8 image; `mask` is ignored like in the reference."""
        if image is None or image.size == 0:
            return np.zeros(0, KEYPOINT_DTYPE), np.zeros((0, 32), np.uint8)
        if image.dtype != np.uint8 or image.ndim != 2:
            raise ValueError("image must be a 2-D uint8 array (CV_8UC1)")
        if image.strides[1] != 1:
            image = np.ascontiguousarray(image)
        h, w = image.shape
        cap = self.max_keypoints
        kps = np.zeros(cap, KEYPOINT_DTYPE); desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int32(0)
        _check(self._lib.orbgpu_extract(self._h, _p(image), w, h, image.strides[0], _p(kps), _p(desc), cap,
                                        C.byref(n)), "orbgpu_extract")
        return kps[:n.value].copy(), desc[:n.value].copy()

    def extract_batch(self, images):
        """images: uint8 [B,h,w] (contiguous). Returns list of (keypoints, descriptors)."""
        images = np.ascontiguousarray(images, np.uint8)
        B, h, w = images.shape
        cap = self.max_keypoints
        kps = np.zeros((B, cap), KEYPOINT_DTYPE); desc = np.zeros((B, cap, 32), np.uint8)
        counts = np.zeros(B, np.int32)
        ptrs = (C.c_void_p * B)(*[images[i].ctypes.data for i in range(B)])
        _check(self._lib.orbgpu_extract_batch(self._h, ptrs, B, w, h, images.strides[1], _p(kps), _p(desc), cap,
                                              _p(counts)), "orbgpu_extract_batch")
        return [(kps[i, :counts[i]].copy(), desc[i, :counts[i]].copy()) for i in range(B)]

    def extract_batch_into(self, images, kps, desc, counts):
        """orbgpu_extract_batch into caller-owned arrays: images uint8 [B,h,w] contiguous, kps [B,cap] of KEYPOINT_DTYPE,
        desc uint8 [B,cap,32], counts int32 [B].  Nothing is allocated or copied on the Python side; page-locked
        arrays (pinned_empty) are used by the GPU in place."""
        B, h, w = images.shape
        cap = kps.shape[1]
        assert images.dtype == np.uint8 and images.flags.c_contiguous and kps.flags.c_contiguous and desc.flags.c_contiguous
        assert kps.dtype == KEYPOINT_DTYPE and desc.shape == (B, cap, 32) and counts.dtype == np.int32 and len(counts) == B
        ptrs = (C.c_void_p * B)(*[images.ctypes.data + i * h * w for i in range(B)])
        _check(self._lib.orbgpu_extract_batch(self._h, ptrs, B, w, h, w, _p(kps), _p(desc), cap, _p(counts)), "orbgpu_extract_batch")

    def extract_batch_device(self, d_imgs_ptr, nimg, w, h, stride, frame_stride, d_kps_ptr, d_desc_ptr, capacity,
                             d_counts_ptr):
        _check(self._lib.orbgpu_extract_batch_device(self._h, C.c_void_p(d_imgs_ptr), nimg, w, h, stride,
                                                     C.c_size_t(frame_stride), C.c_void_p(d_kps_ptr),
                                                     C.c_void_p(d_desc_ptr), capacity, C.c_void_p(d_counts_ptr)),
               "orbgpu_extract_batch_device")

    def set_stream(self, hip_stream):
        _check(self._lib.orbgpu_extractor_set_stream(self._h, C.c_void_p(hip_stream)), "orbgpu_extractor_set_stream")

    def sync(self):
        _check(self._lib.orbgpu_extractor_sync(self._h), "orbgpu_extractor_sync")

    def status(self):
        """Overflow flags of the last extract call (bit 0: quadtree node storage, bit 1: keypoints beyond capacity)."""
        f = C.c_int32()
        _check(self._lib.orbgpu_extractor_status(self._h, C.byref(f)), "orbgpu_extractor_status")
        return f.value

    def profile(self, enable=True):
        _check(self._lib.orbgpu_extractor_profile(self._h, int(enable)), "orbgpu_extractor_profile")

    def stage_times(self):
        """Mean ms of (pyramid, fast_cells, octree, blur, describe) and the number of calls averaged."""
        ms = np.zeros(5, np.float32); n = C.c_int32()
        _check(self._lib.orbgpu_extractor_stage_times(self._h, _p(ms), C.byref(n)), "orbgpu_extractor_stage_times")
        return ms, n.value

    # ---- getters (include/ORBextractor.h:63-83) -----------------------------
    def GetLevels(self):
        return self.nlevels

    def GetScaleFactor(self):
        return float(self.params.scaleFactor)

    def GetScaleFactors(self):
        return self._scale.copy()

    def GetInverseScaleFactors(self):
        return self._inv.copy()

    def GetScaleSigmaSquares(self):
        return self._s2.copy()

    def GetInverseScaleSigmaSquares(self):
        return self._is2.copy()

    @property
    def mnFeaturesPerLevel(self):
        return self._quota.copy()

    # ---- pyramid / stage access -------------------------------------------------
    def pyramid_level(self, level, frame=0, blurred=False, border=0):
        w, h = C.c_int32(), C.c_int32()
        _check(self._lib.orbgpu_pyramid_level(self._h, frame, level, int(blurred), border, None, 0, C.byref(w),
                                              C.byref(h)), "orbgpu_pyramid_level")
        out = np.zeros((h.value + 2 * border, w.value + 2 * border), np.uint8)
        _check(self._lib.orbgpu_pyramid_level(self._h, frame, level, int(blurred), border, _p(out), out.strides[0],
                                              C.byref(w), C.byref(h)), "orbgpu_pyramid_level")
        return out

    def _debug(self, fn, level, frame, cap):
        xs = np.zeros(cap, np.int32); ys = np.zeros(cap, np.int32); rs = np.zeros(cap, np.int32)
        n = C.c_int32()
        _check(fn(self._h, frame, level, _p(xs), _p(ys), _p(rs), cap, C.byref(n)), "orbgpu_debug_level")
        return xs[:n.value].copy(), ys[:n.value].copy(), rs[:n.value].copy()

    def debug_level_candidates(self, level, frame=0, cap=1 << 20):
        return self._debug(self._lib.orbgpu_debug_level_candidates, level, frame, cap)

    def debug_level_selected(self, level, frame=0):
        return self._debug(self._lib.orbgpu_debug_level_selected, level, frame, 8192)


class _PyramidView:
    """`extractor.mvImagePyramid[level]` -> host copy of that level of the last frame."""

    def __init__(self, ex):
        self._ex = ex

    def __len__(self):
        return self._ex.nlevels

    def __getitem__(self, level):
        return self._ex.pyramid_level(level)


class Frame:
    """The device-resident state of a Frame (orbgpu_frame in include/orbgpu.h): mvKeys, mvKeysUn, mDescriptors,
    mvuRight, mvDepth and mGrid as the constructors of src/Frame.cc:58-226 leave them, kept on the GPU.  The host
    mirrors below are fetched on first use."""

    def __init__(self, capacity, device=0):
        self._lib = load_library()
        self._h = C.c_void_p()
        _check(self._lib.orbgpu_frame_create(device, int(capacity), C.byref(self._h)), "orbgpu_frame_create")
        self.capacity = int(capacity)
        self._cache = {}
        self.N = 0
        self.nvalid = 0

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.orbgpu_frame_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _cam(K4, dist5, bounds):
        return (np.ascontiguousarray(K4, np.float32), np.ascontiguousarray(dist5, np.float32),
                np.ascontiguousarray(bounds, np.float32))

    def _built(self, n):
        self._cache = {}
        self.N = n.value
        nv = C.c_int32()
        _check(self._lib.orbgpu_frame_size(self._h, None, C.byref(nv), None), "orbgpu_frame_size")
        self.nvalid = nv.value
        return self.N

    def build_mono(self, ex, image, K4, dist5, bounds):
        """Frame::Frame, monocular (src/Frame.cc:175-226).  Returns N."""
        img = np.ascontiguousarray(image, np.uint8)
        K, D, b = self._cam(K4, dist5, bounds)
        n = C.c_int32()
        _check(self._lib.orbgpu_frame_build_mono(self._h, ex._h, _p(img), img.shape[1], img.shape[0], img.shape[1], _p(K), _p(D),
                                                 _p(b), C.byref(n)), "orbgpu_frame_build_mono")
        return self._built(n)

    def build_rgbd(self, ex, image, depth, bf, K4, dist5, bounds):
        """Frame::Frame, RGB-D (src/Frame.cc:119-172).  Returns N."""
        img = np.ascontiguousarray(image, np.uint8)
        dep = np.ascontiguousarray(depth, np.float32)
        K, D, b = self._cam(K4, dist5, bounds)
        n = C.c_int32()
        _check(self._lib.orbgpu_frame_build_rgbd(self._h, ex._h, _p(img), img.shape[1], img.shape[0], img.shape[1], _p(dep),
                                                 dep.shape[1], bf, _p(K), _p(D), _p(b), C.byref(n)), "orbgpu_frame_build_rgbd")
        return self._built(n)

    def build_stereo(self, left, right, imL, imR, bf, b_line, K4, dist5, bounds):
        """Frame::Frame, stereo (src/Frame.cc:58-116).  Returns (N, keypoints with a depth)."""
        iL = np.ascontiguousarray(imL, np.uint8); iR = np.ascontiguousarray(imR, np.uint8)
        assert iL.shape == iR.shape
        K, D, b = self._cam(K4, dist5, bounds)
        n, nv = C.c_int32(), C.c_int32()
        _check(self._lib.orbgpu_frame_build_stereo(self._h, left._h, right._h, _p(iL), _p(iR), iL.shape[1], iL.shape[0], iL.shape[1],
                                                   bf, b_line, _p(K), _p(D), _p(b), C.byref(n), C.byref(nv)),
               "orbgpu_frame_build_stereo")
        self._built(n)
        return self.N, nv.value

    def upload(self, kps_un, desc, bounds, kps=None, uright=None, depth=None, nlevels=8):
        """A frame from host arrays (orbgpu_frame_upload).  Returns N."""
        ku = np.ascontiguousarray(kps_un); assert ku.dtype == KEYPOINT_DTYPE
        d = np.ascontiguousarray(desc, np.uint8)
        k = None if kps is None else np.ascontiguousarray(kps)
        ur = None if uright is None else np.ascontiguousarray(uright, np.float32)
        dp = None if depth is None else np.ascontiguousarray(depth, np.float32)
        b = np.ascontiguousarray(bounds, np.float32)
        _check(self._lib.orbgpu_frame_upload(self._h, _p(k), _p(ku), _p(d), _p(ur), _p(dp), len(ku), nlevels, _p(b)),
               "orbgpu_frame_upload")
        n = C.c_int32(len(ku))
        return self._built(n)

    def _fetch(self, which):
        if which not in self._cache:
            n = max(self.N, 1)
            bufs = {"mvKeys": np.zeros(n, KEYPOINT_DTYPE), "mvKeysUn": np.zeros(n, KEYPOINT_DTYPE),
                    "mDescriptors": np.zeros((n, 32), np.uint8), "mvuRight": np.zeros(n, np.float32),
                    "mvDepth": np.zeros(n, np.float32)}
            order = ["mvKeys", "mvKeysUn", "mDescriptors", "mvuRight", "mvDepth"]
            args = [_p(bufs[k]) if k == which else None for k in order]
            _check(self._lib.orbgpu_frame_download(self._h, *args), "orbgpu_frame_download")
            self._cache[which] = bufs[which][:self.N]
        return self._cache[which]

    mvKeys = property(lambda self: self._fetch("mvKeys"))
    mvKeysUn = property(lambda self: self._fetch("mvKeysUn"))
    mDescriptors = property(lambda self: self._fetch("mDescriptors"))
    mvuRight = property(lambda self: self._fetch("mvuRight"))
    mvDepth = property(lambda self: self._fetch("mvDepth"))

    def grid(self):
        """mGrid as (cell_start[64*48+1], cell_index), cell = ix*48 + iy (src/Frame.cc:270-293)."""
        cs = np.zeros(64 * 48 + 1, np.int32); ci = np.zeros(max(self.N, 1), np.int32)
        n = C.c_int32()
        _check(self._lib.orbgpu_frame_grid(self._h, _p(cs), _p(ci), C.byref(n)), "orbgpu_frame_grid")
        return cs, ci[:n.value].copy()


def bow_segments(featvec1, featvec2, valid1):
    """Intersect two FeatureVectors (src/ORBmatcher.cc:175-264): for every common node, ascending,
    the side-1 features with a good map point become queries (in the node's index order) and the
    side-2 indices one segment.  Returns (q_index, q_seg, seg_start, seg_index) as int32 arrays."""
    q_index, q_seg, seg_start, seg_index = [], [], [0], []
    for node in sorted(set(featvec1) & set(featvec2)):
        s = len(seg_start) - 1
        seg_index.extend(int(t) for t in featvec2[node])
        seg_start.append(len(seg_index))
        for i in featvec1[node]:
            if valid1 is None or valid1[i]:
                q_index.append(int(i)); q_seg.append(s)
    return (np.asarray(q_index, np.int32), np.asarray(q_seg, np.int32), np.asarray(seg_start, np.int32),
            np.asarray(seg_index if seg_index else [0], np.int32))  # never an empty buffer; seg_start bounds its use


class ORBmatcher:
    """Mirror of ORB_SLAM2::ORBmatcher (include/ORBmatcher.h:41-103): the distance
    primitive, the brute-force scan and the windowed searches."""

    TH_HIGH, TH_LOW, HISTO_LENGTH = TH_HIGH, TH_LOW, HISTO_LENGTH

    def __init__(self, nnratio=0.6, checkOri=True, device=0):
        self._lib = load_library()
        self._h = C.c_void_p()
        self.mfNNratio = float(nnratio)
        self.mbCheckOrientation = bool(checkOri)
        _check(self._lib.orbgpu_matcher_create(device, C.byref(self._h)), "orbgpu_matcher_create")

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.orbgpu_matcher_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream):
        _check(self._lib.orbgpu_matcher_set_stream(self._h, C.c_void_p(hip_stream)), "orbgpu_matcher_set_stream")

    def sync(self):
        _check(self._lib.orbgpu_matcher_sync(self._h), "orbgpu_matcher_sync")

    def set_grid_geometry(self, geom6):
        """Grid geometry of the next windowed search only: (build origin x, y, query origin x, y, 1/cell w, 1/cell h)."""
        g = None if geom6 is None else np.ascontiguousarray(geom6, np.float32)
        _check(self._lib.orbgpu_matcher_set_grid_geometry(self._h, _p(g)), "orbgpu_matcher_set_grid_geometry")

    def set_bruteforce_kernel(self, which):
        """0: by size, 1: vector-unit kernel, 2: matrix-core kernel (same results)."""
        _check(self._lib.orbgpu_matcher_set_bruteforce_kernel(self._h, int(which)), "orbgpu_matcher_set_bruteforce_kernel")

    @staticmethod
    def DescriptorDistance(a, b):
        a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
        assert a.size == 32 and b.size == 32
        return load_library().orbgpu_descriptor_distance(_p(a), _p(b))

    def BruteForce(self, Q, T):
        Q = np.ascontiguousarray(Q, np.uint8).reshape(-1, 32); T = np.ascontiguousarray(T, np.uint8).reshape(-1, 32)
        nq, nt = len(Q), len(T)
        bi = np.zeros(nq, np.int32); b1 = np.zeros(nq, np.int32); b2 = np.zeros(nq, np.int32)
        _check(self._lib.orbgpu_hamming_bruteforce(self._h, _p(Q), nq, _p(T) if nt else None, nt, _p(bi), _p(b1),
                                                   _p(b2)), "orbgpu_hamming_bruteforce")
        return bi, b1, b2

    def bruteforce_device(self, dQ, nq, dT, nt, d_idx, d_d1, d_d2):
        _check(self._lib.orbgpu_hamming_bruteforce_device(self._h, C.c_void_p(dQ), nq, C.c_void_p(dT), nt,
                                                          C.c_void_p(d_idx), C.c_void_p(d_d1), C.c_void_p(d_d2)),
               "orbgpu_hamming_bruteforce_device")

    def bruteforce_device_noalloc(self, dQ, nq, dT, nt, d_idx, d_d1, d_d2):
        _check(self._lib.orbgpu_hamming_bruteforce_device_noalloc(self._h, C.c_void_p(dQ), nq, C.c_void_p(dT), nt,
                                                                  C.c_void_p(d_idx), C.c_void_p(d_d1), C.c_void_p(d_d2)),
               "orbgpu_hamming_bruteforce_device_noalloc")

    def window_stats(self, pair=0):
        it, rs = C.c_int32(), C.c_int32()
        _check(self._lib.orbgpu_debug_window_stats(self._h, pair, C.byref(it), C.byref(rs)), "orbgpu_debug_window_stats")
        return it.value, rs.value

    def SearchByProjection(self, mode, queries, qdesc, kps2, desc2, bounds, uright2=None, occupied=None,
                           th_dist=TH_HIGH):
        """Returns (nmatches, assign[n2]); see orbgpu_search_by_projection in include/orbgpu.h."""
        q = np.ascontiguousarray(queries); assert q.dtype == WQUERY_DTYPE
        qdesc = np.ascontiguousarray(qdesc, np.uint8)
        kps2 = np.ascontiguousarray(kps2); desc2 = np.ascontiguousarray(desc2, np.uint8)
        b = np.asarray(bounds, np.float32)
        ur = None if uright2 is None else np.ascontiguousarray(uright2, np.float32)
        oc = None if occupied is None else np.ascontiguousarray(occupied, np.uint8)
        assign = np.zeros(max(len(kps2), 1), np.int32)
        n = C.c_int32()
        _check(self._lib.orbgpu_search_by_projection(self._h, mode, _p(q), _p(qdesc), len(q), _p(kps2), _p(desc2),
                                                     _p(ur), len(kps2), _p(b), _p(oc), _p(assign), th_dist,
                                                     self.mfNNratio, int(self.mbCheckOrientation), C.byref(n)),
               "orbgpu_search_by_projection")
        return n.value, assign[:len(kps2)].copy()

    def SearchByBoW(self, featvec1, kps1, desc1, valid1, featvec2, kps2, desc2, excluded2=None, strict=False):
        """ORBmatcher::SearchByBoW (src/ORBmatcher.cc:159-288; strict=True: the KeyFrame x KeyFrame
        overload :573-706 with `bestDist1 < TH_LOW`).  featvec1 / featvec2: DBoW2 FeatureVectors as
        {node id: [feature indices]}; valid1[i]: feature i of side 1 owns a good map point.
        Returns (nmatches, assign[n2] = side-1 feature per side-2 feature or -1,
        matches12[n1] = side-2 feature per side-1 feature or -1)."""
        kps1 = np.ascontiguousarray(kps1); kps2 = np.ascontiguousarray(kps2)
        desc1 = np.ascontiguousarray(desc1, np.uint8); desc2 = np.ascontiguousarray(desc2, np.uint8)
        q_index, q_seg, seg_start, seg_index = bow_segments(featvec1, featvec2, valid1)
        q = np.zeros(max(len(q_index), 1), WQUERY_DTYPE)[:len(q_index)]
        q["angle"] = kps1["angle"][q_index]; q["minLevel"] = q_seg; q["flags"] = 1
        qd = np.ascontiguousarray(desc1[q_index]) if len(q_index) else np.zeros((0, 32), np.uint8)
        ex = None if excluded2 is None else np.ascontiguousarray(excluded2, np.uint8)
        assign = np.zeros(max(len(kps2), 1), np.int32); qm = np.zeros(max(len(q), 1), np.int32)
        n = C.c_int32()
        _check(self._lib.orbgpu_search_by_bow(self._h, _p(q), _p(qd), len(q), _p(kps2), _p(desc2), len(kps2), _p(seg_start),
                                              _p(seg_index), len(seg_start) - 1, _p(ex), TH_LOW - 1 if strict else TH_LOW,
                                              self.mfNNratio, int(self.mbCheckOrientation), _p(assign), _p(qm), C.byref(n)),
               "orbgpu_search_by_bow")
        assign = assign[:len(kps2)]
        a1 = np.where(assign >= 0, q_index[np.clip(assign, 0, None)] if len(q_index) else -1, -1).astype(np.int32)
        m12 = np.full(len(kps1), -1, np.int32)
        qm = qm[:len(q)]
        if len(q_index):
            m12[q_index[qm >= 0]] = qm[qm >= 0]
        return n.value, a1, m12

    def SearchForTriangulation(self, featvec1, kps1, desc1, has_mp1, uright1, featvec2, kps2, desc2, has_mp2, uright2, F12,
                               epipole, scale2, sigma2, bOnlyStereo=False):
        """ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:708-874).  kps = mvKeysUn, uright = mvuRight (None:
        monocular), has_mp = feature already holds a map point, epipole = (ex, ey) of :714-721, scale2 / sigma2 =
        mvScaleFactors / mvLevelSigma2 of keyframe 2.  Returns (nmatches, matches12[n1])."""
        kps1 = np.ascontiguousarray(kps1); kps2 = np.ascontiguousarray(kps2)
        desc1 = np.ascontiguousarray(desc1, np.uint8); desc2 = np.ascontiguousarray(desc2, np.uint8)
        stereo1 = np.zeros(len(kps1), bool) if uright1 is None else np.asarray(uright1) >= 0
        use1 = ~np.asarray(has_mp1, bool) & (stereo1 if bOnlyStereo else True)
        q_index, q_seg, seg_start, seg_index = bow_segments(featvec1, featvec2, use1)
        q = np.zeros(max(len(q_index), 1), WQUERY_DTYPE)[:len(q_index)]
        q["u"] = kps1["x"][q_index]; q["v"] = kps1["y"][q_index]; q["angle"] = kps1["angle"][q_index]
        q["minLevel"] = q_seg; q["flags"] = 1 + 4 * stereo1[q_index]
        qd = np.ascontiguousarray(desc1[q_index]) if len(q_index) else np.zeros((0, 32), np.uint8)
        ex2 = np.ascontiguousarray(has_mp2, np.uint8)
        ur2 = None if uright2 is None else np.ascontiguousarray(uright2, np.float32)
        F = np.ascontiguousarray(F12, np.float32).reshape(9)
        sc = np.ascontiguousarray(scale2, np.float32); sg = np.ascontiguousarray(sigma2, np.float32)
        qm = np.zeros(max(len(q), 1), np.int32)
        n = C.c_int32()
        _check(self._lib.orbgpu_search_for_triangulation(self._h, _p(q), _p(qd), len(q), _p(kps2), _p(desc2), _p(ur2), len(kps2),
                                                         _p(seg_start), _p(seg_index), len(seg_start) - 1, _p(ex2), _p(F),
                                                         float(epipole[0]), float(epipole[1]), _p(sc), _p(sg), len(sc),
                                                         int(bOnlyStereo), int(self.mbCheckOrientation), _p(qm), C.byref(n)),
               "orbgpu_search_for_triangulation")
        m12 = np.full(len(kps1), -1, np.int32)
        qm = qm[:len(q)]
        if len(q_index):
            m12[q_index[qm >= 0]] = qm[qm >= 0]
        return n.value, m12

    def SearchInArea(self, queries, qdesc, kps2, desc2, bounds, uright2=None, inv_sigma2=None, chi2_gate=False, th_dist=TH_LOW):
        """The keypoint search inside ORBmatcher::Fuse (src/ORBmatcher.cc:876-1151; chi2_gate=True for the first
        overload) and ORBmatcher::SearchBySim3 (:1153-1377, th_dist=TH_HIGH, once per direction): every query
        independently takes the nearest keypoint of its window.  Returns (nmatches, qmatch[nq])."""
        q = np.ascontiguousarray(queries); qdesc = np.ascontiguousarray(qdesc, np.uint8)
        kps2 = np.ascontiguousarray(kps2); desc2 = np.ascontiguousarray(desc2, np.uint8)
        b = np.asarray(bounds, np.float32)
        ur = None if uright2 is None else np.ascontiguousarray(uright2, np.float32)
        sg = None if inv_sigma2 is None else np.ascontiguousarray(inv_sigma2, np.float32)
        qm = np.zeros(max(len(q), 1), np.int32)
        n = C.c_int32()
        _check(self._lib.orbgpu_search_in_area(self._h, _p(q), _p(qdesc), len(q), _p(kps2), _p(desc2), _p(ur), len(kps2), _p(b),
                                               _p(sg), 0 if sg is None else len(sg), int(chi2_gate), th_dist, _p(qm), C.byref(n)),
               "orbgpu_search_in_area")
        return n.value, qm[:len(q)].copy()

    def AssignFeaturesToGrid(self, kps_un, bounds):
        """Frame::AssignFeaturesToGrid (src/Frame.cc:270-293): (cell_start[64*48+1], cell_index) with cell = ix*48 + iy."""
        kps = np.ascontiguousarray(kps_un); b = np.asarray(bounds, np.float32)
        cs = np.zeros(64 * 48 + 1, np.int32); ci = np.zeros(max(len(kps), 1), np.int32)
        n = C.c_int32()
        _check(self._lib.orbgpu_assign_features_to_grid(self._h, _p(kps), len(kps), _p(b), _p(cs), _p(ci), C.byref(n)),
               "orbgpu_assign_features_to_grid")
        return cs, ci[:n.value].copy()

    def UndistortKeyPoints(self, kps, K4, dist5):
        """Frame::UndistortKeyPoints (src/Frame.cc:517-558); K4 = (fx, fy, cx, cy), dist5 = (k1, k2, p1, p2, k3)."""
        kps = np.ascontiguousarray(kps); out = np.zeros_like(kps)
        K = np.ascontiguousarray(K4, np.float32); D = np.ascontiguousarray(dist5, np.float32)
        _check(self._lib.orbgpu_undistort_keypoints(self._h, _p(kps), len(kps), _p(K), _p(D), _p(out)), "orbgpu_undistort_keypoints")
        return out

    def ComputeDistinctiveDescriptors(self, desc, start):
        """MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:262-340) for many map points at once: observations of
        map point p are desc[start[p]:start[p+1]].  Returns (best index within the range, its median distance)."""
        desc = np.ascontiguousarray(desc, np.uint8); start = np.ascontiguousarray(start, np.int32)
        n = len(start) - 1
        bi = np.zeros(max(n, 1), np.int32); bm = np.zeros(max(n, 1), np.int32)
        _check(self._lib.orbgpu_distinctive_descriptors(self._h, _p(desc), _p(start), n, _p(bi), _p(bm)),
               "orbgpu_distinctive_descriptors")
        return bi[:n].copy(), bm[:n].copy()

    def ComputeStereoFromRGBD(self, kps, kps_un, depth, bf):
        """Frame::ComputeStereoFromRGBD (src/Frame.cc:777-798).  Returns (mvuRight, mvDepth)."""
        kps = np.ascontiguousarray(kps); kps_un = np.ascontiguousarray(kps_un)
        depth = np.ascontiguousarray(depth, np.float32)
        ur = np.zeros(max(len(kps), 1), np.float32); dp = np.zeros(max(len(kps), 1), np.float32)
        _check(self._lib.orbgpu_stereo_from_rgbd(self._h, _p(kps), _p(kps_un), len(kps), _p(depth), depth.shape[1], depth.shape[0],
                                                 depth.shape[1], bf, _p(ur), _p(dp)), "orbgpu_stereo_from_rgbd")
        return ur[:len(kps)].copy(), dp[:len(kps)].copy()

    def SearchForInitialization(self, kps1, desc1, kps2, desc2, bounds, vbPrevMatched, windowSize=10):
        """Returns (nmatches, vnMatches12, updated vbPrevMatched)."""
        kps1 = np.ascontiguousarray(kps1); kps2 = np.ascontiguousarray(kps2)
        desc1 = np.ascontiguousarray(desc1, np.uint8); desc2 = np.ascontiguousarray(desc2, np.uint8)
        b = np.asarray(bounds, np.float32)
        prev = np.ascontiguousarray(vbPrevMatched, np.float32).copy()
        m12 = np.zeros(max(len(kps1), 1), np.int32)
        n = C.c_int32()
        _check(self._lib.orbgpu_search_for_initialization(self._h, _p(kps1), _p(desc1), len(kps1), _p(kps2), _p(desc2),
                                                          len(kps2), _p(b), _p(prev), _p(m12), windowSize,
                                                          self.mfNNratio, int(self.mbCheckOrientation), C.byref(n)),
               "orbgpu_search_for_initialization")
        return n.value, m12[:len(kps1)].copy(), prev

    def ComputeStereoMatches(self, left, right, kpsL, descL, kpsR, descR, bf, b, frame=0):
        """Frame::ComputeStereoMatches on the pyramids held by two extractors. Returns (nvalid, mvuRight, mvDepth)."""
        kpsL = np.ascontiguousarray(kpsL); kpsR = np.ascontiguousarray(kpsR)
        descL = np.ascontiguousarray(descL, np.uint8); descR = np.ascontiguousarray(descR, np.uint8)
        uR = np.zeros(max(len(kpsL), 1), np.float32); dep = np.zeros(max(len(kpsL), 1), np.float32)
        n = C.c_int32()
        _check(self._lib.orbgpu_stereo_match(self._h, left._h, right._h, frame, _p(kpsL), _p(descL), len(kpsL),
                                             _p(kpsR), _p(descR), len(kpsR), bf, b, _p(uR), _p(dep), C.byref(n)),
               "orbgpu_stereo_match")
        return n.value, uR[:len(kpsL)].copy(), dep[:len(kpsL)].copy()

    def search_for_initialization_batch_device(self, npairs, d_kps1, d_desc1, d_n1, cap1, d_kps2, d_desc2, d_n2, cap2, bounds, d_matches12,
                                               d_nmatches, windowSize=100):
        """orbgpu_search_for_initialization_batch_device: device pointers as integers, asynchronous on the matcher's stream."""
        v = C.c_void_p
        b = np.asarray(bounds, np.float32)
        _check(self._lib.orbgpu_search_for_initialization_batch_device(
            self._h, npairs, v(d_kps1), v(d_desc1), v(d_n1), cap1, v(d_kps2), v(d_desc2), v(d_n2), cap2, _p(b), windowSize,
            self.mfNNratio, int(self.mbCheckOrientation), v(d_matches12), v(d_nmatches)), "orbgpu_search_for_initialization_batch_device")

    def stereo_match_batch_device(self, left, right, npairs, d_kpsL, d_descL, d_nL, capL, d_kpsR, d_descR, d_nR, capR, bf, b,
                                  d_uRight, d_depth, d_nvalid):
        """orbgpu_stereo_match_batch_device: device pointers as integers, asynchronous on the matcher's stream."""
        v = C.c_void_p
        _check(self._lib.orbgpu_stereo_match_batch_device(self._h, left._h, right._h, npairs, v(d_kpsL), v(d_descL), v(d_nL), capL,
                                                          v(d_kpsR), v(d_descR), v(d_nR), capR, bf, b, v(d_uRight), v(d_depth),
                                                          v(d_nvalid)), "orbgpu_stereo_match_batch_device")

    # ---- the same searches against a device-resident Frame (nothing of the train side is uploaded) ----

    def SearchByProjectionFrame(self, mode, queries, qdesc, frame, occupied=None, th_dist=TH_HIGH):
        q = np.ascontiguousarray(queries); assert q.dtype == WQUERY_DTYPE
        qdesc = np.ascontiguousarray(qdesc, np.uint8)
        oc = None if occupied is None else np.ascontiguousarray(occupied, np.uint8)
        assign = np.zeros(max(frame.N, 1), np.int32)
        n = C.c_int32()
        _check(self._lib.orbgpu_search_by_projection_frame(self._h, mode, _p(q), _p(qdesc), len(q), frame._h, _p(oc), _p(assign),
                                                           th_dist, self.mfNNratio, int(self.mbCheckOrientation), C.byref(n)),
               "orbgpu_search_by_projection_frame")
        return n.value, assign[:frame.N].copy()

    def SearchInAreaFrame(self, queries, qdesc, frame, inv_sigma2=None, chi2_gate=False, th_dist=TH_LOW):
        q = np.ascontiguousarray(queries); qdesc = np.ascontiguousarray(qdesc, np.uint8)
        sg = None if inv_sigma2 is None else np.ascontiguousarray(inv_sigma2, np.float32)
        qm = np.zeros(max(len(q), 1), np.int32)
        n = C.c_int32()
        _check(self._lib.orbgpu_search_in_area_frame(self._h, _p(q), _p(qdesc), len(q), frame._h, _p(sg), 0 if sg is None else len(sg),
                                                     int(chi2_gate), th_dist, _p(qm), C.byref(n)), "orbgpu_search_in_area_frame")
        return n.value, qm[:len(q)].copy()

    def SearchByBoWFrame(self, featvec1, kps1, desc1, valid1, featvec2, frame2, excluded2=None, strict=False):
        kps1 = np.ascontiguousarray(kps1); desc1 = np.ascontiguousarray(desc1, np.uint8)
        q_index, q_seg, seg_start, seg_index = bow_segments(featvec1, featvec2, valid1)
        q = np.zeros(max(len(q_index), 1), WQUERY_DTYPE)[:len(q_index)]
        q["angle"] = kps1["angle"][q_index]; q["minLevel"] = q_seg; q["flags"] = 1
        qd = np.ascontiguousarray(desc1[q_index]) if len(q_index) else np.zeros((0, 32), np.uint8)
        ex = None if excluded2 is None else np.ascontiguousarray(excluded2, np.uint8)
        n2 = frame2.N
        assign = np.zeros(max(n2, 1), np.int32); qm = np.zeros(max(len(q), 1), np.int32)
        n = C.c_int32()
        _check(self._lib.orbgpu_search_by_bow_frame(self._h, _p(q), _p(qd), len(q), frame2._h, _p(seg_start), _p(seg_index),
                                                    len(seg_start) - 1, _p(ex), TH_LOW - 1 if strict else TH_LOW, self.mfNNratio,
                                                    int(self.mbCheckOrientation), _p(assign), _p(qm), C.byref(n)),
               "orbgpu_search_by_bow_frame")
        assign = assign[:n2]
        a1 = np.where(assign >= 0, q_index[np.clip(assign, 0, None)] if len(q_index) else -1, -1).astype(np.int32)
        m12 = np.full(len(kps1), -1, np.int32)
        qm = qm[:len(q)]
        if len(q_index):
            m12[q_index[qm >= 0]] = qm[qm >= 0]
        return n.value, a1, m12

    def SearchForTriangulationFrame(self, featvec1, kps1, desc1, has_mp1, uright1, featvec2, frame2, has_mp2, use_uright2, F12,
                                    epipole, scale2, sigma2, bOnlyStereo=False):
        kps1 = np.ascontiguousarray(kps1); desc1 = np.ascontiguousarray(desc1, np.uint8)
        stereo1 = np.zeros(len(kps1), bool) if uright1 is None else np.asarray(uright1) >= 0
        use1 = ~np.asarray(has_mp1, bool) & (stereo1 if bOnlyStereo else True)
        q_index, q_seg, seg_start, seg_index = bow_segments(featvec1, featvec2, use1)
        q = np.zeros(max(len(q_index), 1), WQUERY_DTYPE)[:len(q_index)]
        q["u"] = kps1["x"][q_index]; q["v"] = kps1["y"][q_index]; q["angle"] = kps1["angle"][q_index]
        q["minLevel"] = q_seg; q["flags"] = 1 + 4 * stereo1[q_index]
        qd = np.ascontiguousarray(desc1[q_index]) if len(q_index) else np.zeros((0, 32), np.uint8)
        ex2 = np.ascontiguousarray(has_mp2, np.uint8)
        F = np.ascontiguousarray(F12, np.float32).reshape(9)
        sc = np.ascontiguousarray(scale2, np.float32); sg = np.ascontiguousarray(sigma2, np.float32)
        qm = np.zeros(max(len(q), 1), np.int32)
        n = C.c_int32()
        _check(self._lib.orbgpu_search_for_triangulation_frame(self._h, _p(q), _p(qd), len(q), frame2._h, int(use_uright2),
                                                               _p(seg_start), _p(seg_index), len(seg_start) - 1, _p(ex2), _p(F),
                                                               float(epipole[0]), float(epipole[1]), _p(sc), _p(sg), len(sc),
                                                               int(bOnlyStereo), int(self.mbCheckOrientation), _p(qm), C.byref(n)),
               "orbgpu_search_for_triangulation_frame")
        m12 = np.full(len(kps1), -1, np.int32)
        qm = qm[:len(q)]
        if len(q_index):
            m12[q_index[qm >= 0]] = qm[qm >= 0]
        return n.value, m12

    def SearchForInitializationFrames(self, frame1, frame2, vbPrevMatched, windowSize=10):
        """Returns (nmatches, vnMatches12, updated vbPrevMatched)."""
        prev = np.ascontiguousarray(vbPrevMatched, np.float32).copy()
        m12 = np.zeros(max(frame1.N, 1), np.int32)
        n = C.c_int32()
        _check(self._lib.orbgpu_search_for_initialization_frames(self._h, frame1._h, frame2._h, _p(prev), _p(m12), windowSize,
                                                                 self.mfNNratio, int(self.mbCheckOrientation), C.byref(n)),
               "orbgpu_search_for_initialization_frames")
        return n.value, m12[:frame1.N].copy(), prev

    def search_by_projection_batch_device(self, mode, npairs, d_q, d_qdesc, d_nq, q_cap, d_kps2, d_desc2, d_n2, t_cap,
                                          bounds, d_assign, d_nmatches, th_dist=TH_HIGH, d_uright2=0, d_occupied=0):
        b = np.asarray(bounds, np.float32)
        _check(self._lib.orbgpu_search_by_projection_batch_device(
            self._h, mode, npairs, C.c_void_p(d_q), C.c_void_p(d_qdesc), C.c_void_p(d_nq), q_cap, C.c_void_p(d_kps2),
            C.c_void_p(d_desc2), C.c_void_p(d_uright2) if d_uright2 else None, C.c_void_p(d_n2), t_cap, _p(b),
            C.c_void_p(d_occupied) if d_occupied else None, C.c_void_p(d_assign), th_dist, self.mfNNratio,
            int(self.mbCheckOrientation), C.c_void_p(d_nmatches)), "orbgpu_search_by_projection_batch_device")

    def make_shift_queries_device(self, npairs, d_kps_last, d_n_last, capacity, du, dv, th, scale, bounds, d_q):
        sc = np.ascontiguousarray(scale, np.float32); b = np.asarray(bounds, np.float32)
        _check(self._lib.orbgpu_make_shift_queries_device(self._h, npairs, C.c_void_p(d_kps_last), C.c_void_p(d_n_last),
                                                          capacity, du, dv, th, _p(sc), len(sc), _p(b), C.c_void_p(d_q)),
               "orbgpu_make_shift_queries_device")


def shift_queries(kps_last, du, dv, th, scale, bounds):
    """Host twin of orbgpu_make_shift_queries_device (same float arithmetic)."""
    k = np.ascontiguousarray(kps_last)
    q = np.zeros(len(k), WQUERY_DTYPE)
    sc = np.asarray(scale, np.float32)
    q["u"] = k["x"] + np.float32(du)
    q["v"] = k["y"] + np.float32(dv)
    q["r"] = np.float32(th) * sc[k["octave"]]
    q["minLevel"] = k["octave"] - 1
    q["maxLevel"] = k["octave"] + 1
    q["angle"] = k["angle"]
    q["ur"] = -1.0
    inside = ~((q["u"] < bounds[0]) | (q["u"] > bounds[2]) | (q["v"] < bounds[1]) | (q["v"] > bounds[3]))
    q["flags"] = np.where(inside, 3, 0)
    return q


class ORBVocabulary:
    """DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB> as ORB-SLAM2 uses it (include/ORBVocabulary.h:31-32):
    loadFromTextFile (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1338-1424) and
    transform(features, BowVector, FeatureVector, levelsup) (:1127-1194) on the device."""

    def __init__(self, device=0):
        self._lib = load_library()
        self._device = device
        self._h = C.c_void_p(None)

    def _replace(self, h):
        if self._h:
            self._lib.orbgpu_vocabulary_destroy(self._h)
        self._h = h

    def loadFromTextFile(self, filename):
        """Returns True on success like the reference; the reason of a failure is in LastError()."""
        h = C.c_void_p(None)
        rc = self._lib.orbgpu_vocabulary_load_text(self._device, os.fsencode(filename), C.byref(h))
        if rc != 0:
            return False
        self._replace(h)
        return True

    def LastError(self):
        return self._lib.orbgpu_last_error().decode("utf-8", "replace")

    @classmethod
    def from_arrays(cls, k, L, parent, is_leaf, desc, weight, scoring=0, weighting=0, device=0):
        """Vocabulary from node arrays in file order (node i+1 <- parent[i], is_leaf[i], desc[i], weight[i])."""
        self = cls(device)
        parent = np.ascontiguousarray(parent, np.int32); is_leaf = np.ascontiguousarray(is_leaf, np.uint8)
        desc = np.ascontiguousarray(desc, np.uint8); weight = np.ascontiguousarray(weight, np.float64)
        if not (len(parent) == len(is_leaf) == len(desc) == len(weight)) or (len(desc) and desc.shape[1:] != (32,)):
            raise ValueError("parent, is_leaf, desc [n,32] and weight must describe the same nodes")
        h = C.c_void_p(None)
        _check(self._lib.orbgpu_vocabulary_create(device, k, L, scoring, weighting, len(parent), _p(parent), _p(is_leaf), _p(desc),
                                                  _p(weight), C.byref(h)), "orbgpu_vocabulary_create")
        self._h = h
        return self

    def __del__(self):
        try:
            self._replace(None)
        except Exception:
            pass

    def _info(self):
        v = [C.c_int32(0) for _ in range(6)]
        _check(self._lib.orbgpu_vocabulary_info(self._h, *[C.byref(x) for x in v]), "orbgpu_vocabulary_info")
        return [x.value for x in v]

    def empty(self):
        return (not self._h) or self._info()[5] == 0

    def size(self):
        return self._info()[5] if self._h else 0

    def getBranchingFactor(self):
        return self._info()[0]

    def getDepthLevels(self):
        return self._info()[1]

    def getScoringType(self):
        return self._info()[2]

    def getWeightingType(self):
        return self._info()[3]

    def transform_arrays(self, descriptors, levelsup=4):
        """word_id[n], node_id[n], (bow_ids, bow_vals), (fv_nodes, fv_start, fv_feat): the containers as sorted arrays."""
        if not self._h:
            raise OrbGpuError("the vocabulary has not been loaded")
        d = np.ascontiguousarray(descriptors, np.uint8).reshape(-1, 32)
        n = len(d); m = max(n, 1)
        wid = np.zeros(m, np.int32); nid = np.zeros(m, np.int32)
        bi = np.zeros(m, np.int32); bv = np.zeros(m, np.float64); nb = C.c_int32(0)
        fn = np.zeros(m, np.int32); fs = np.zeros(m + 1, np.int32); ff = np.zeros(m, np.int32); nf = C.c_int32(0)
        _check(self._lib.orbgpu_vocabulary_transform(self._h, _p(d), n, levelsup, _p(wid), _p(nid), _p(bi), _p(bv), C.byref(nb),
                                                     _p(fn), _p(fs), _p(ff), C.byref(nf)), "orbgpu_vocabulary_transform")
        nb = nb.value; nf = nf.value
        return wid[:n], nid[:n], (bi[:nb], bv[:nb]), (fn[:nf], fs[:nf + 1], ff[:fs[nf]])

    def transform(self, descriptors, levelsup=4):
        """(BowVector, FeatureVector) as the reference's maps: {WordId: value}, {NodeId: [feature indices]}."""
        _, _, (bi, bv), (fn, fs, ff) = self.transform_arrays(descriptors, levelsup)
        bow = {int(i): float(v) for i, v in zip(bi, bv)}
        fv = {int(nd): [int(x) for x in ff[fs[j]:fs[j + 1]]] for j, nd in enumerate(fn)}
        return bow, fv

    def transform_batch_device(self, d_desc_ptr, d_counts_ptr, nsets, capacity, levelsup, out_ptrs):
        """Asynchronous device variant; out_ptrs = the nine device pointers of orbgpu_vocabulary_transform_batch_device."""
        _check(self._lib.orbgpu_vocabulary_transform_batch_device(self._h, d_desc_ptr, d_counts_ptr, nsets, capacity, levelsup,
                                                                  *out_ptrs), "orbgpu_vocabulary_transform_batch_device")

    def set_stream(self, stream_ptr):
        _check(self._lib.orbgpu_vocabulary_set_stream(self._h, stream_ptr), "orbgpu_vocabulary_set_stream")

    def sync(self):
        _check(self._lib.orbgpu_vocabulary_sync(self._h), "orbgpu_vocabulary_sync")
