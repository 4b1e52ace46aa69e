"""ctypes view of include/mcrt.h."""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("MCRT_LIB") or os.path.join(_HERE, "libmcrt_hip.so")   # MCRT_LIB: tuning builds only
_LIB = None


class McrtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"mcrt error {code}: {msg}")
        self.code = code


class Params(C.Structure):
    _fields_ = [("n_elements", C.c_uint32), ("n_samples", C.c_uint32), ("max_depth", C.c_uint32), ("n_rows", C.c_uint32),
                ("frequency", C.c_float), ("intensity_epsilon", C.c_float), ("initial_intensity", C.c_float),
                ("ray_start_offset", C.c_float), ("speed_of_sound", C.c_uint32), ("depth_cm", C.c_double),
                ("seed", C.c_uint32), ("sanitize_tir", C.c_uint32), ("tex_n", C.c_uint32), ("tex_res", C.c_float)]


class MeshRec(C.Structure):
    _fields_ = [("mat_inside", C.c_uint32), ("mat_outside", C.c_uint32), ("vascular", C.c_uint32), ("_pad", C.c_uint32)]


class BvhNode(C.Structure):
    _fields_ = [("lo0", C.c_float * 3), ("c0", C.c_int32), ("hi0", C.c_float * 3), ("c1", C.c_int32),
                ("lo1", C.c_float * 3), ("pad0", C.c_uint32), ("hi1", C.c_float * 3), ("pad1", C.c_uint32)]


class Bvh(C.Structure):
    _fields_ = [("n_nodes", C.c_uint32), ("n_tri", C.c_uint32), ("max_depth", C.c_uint32), ("pad_abs", C.c_float),
                ("nodes", C.c_void_p), ("tri", C.c_void_p)]


class Bvh4(C.Structure):
    _fields_ = [("n_nodes", C.c_uint32), ("max_stack", C.c_uint32), ("nodes", C.c_void_p)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("queries", "nodes_visited", "tris_tested", "segments", "rf_steps", "hits")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


NODE_DTYPE = np.dtype([("lo0", "<f4", 3), ("c0", "<i4"), ("hi0", "<f4", 3), ("c1", "<i4"),
                       ("lo1", "<f4", 3), ("pad0", "<u4"), ("hi1", "<f4", 3), ("pad1", "<u4")])
SEGMENT_DTYPE = np.dtype([("from", "<f4", 3), ("to", "<f4", 3), ("dir", "<f4", 3),
                          ("reflected_intensity", "<f4"), ("initial_intensity", "<f4"), ("attenuation", "<f4"),
                          ("distance_traveled", "<f8"), ("media", "<i4"), ("tri", "<i4")])
assert NODE_DTYPE.itemsize == 64 and SEGMENT_DTYPE.itemsize == 64 and C.sizeof(BvhNode) == 64

# every symbol include/mcrt.h declares (tests/test_abi.py checks the .so exports each one)
SYMBOLS = ["mcrt_last_error", "mcrt_version", "mcrt_device_count", "mcrt_create", "mcrt_destroy", "mcrt_set_stream",
           "mcrt_synchronize", "mcrt_default_params", "mcrt_set_params", "mcrt_get_params", "mcrt_import_rf", "mcrt_set_bvh_builder", "mcrt_upload_scene", "mcrt_update_triangles", "mcrt_refit_triangles", "mcrt_upload_texture",
           "mcrt_set_transducer", "mcrt_trace_frame", "mcrt_trace_frames", "mcrt_trace_frames_poses", "mcrt_envelope_frames", "mcrt_scan_convert_frames", "mcrt_trace_frame_debug", "mcrt_cast_rays", "mcrt_convolve", "mcrt_convolve_frames",
           "mcrt_envelope", "mcrt_scan_convert", "mcrt_export_rf", "mcrt_alloc", "mcrt_free", "mcrt_memcpy_d2h",
           "mcrt_memcpy_h2d", "mcrt_enable_stats", "mcrt_get_stats", "mcrt_enable_timing", "mcrt_get_kernel_time", "mcrt_get_kernel_times",
           "mcrt_build_bvh", "mcrt_free_bvh", "mcrt_get_bvh", "mcrt_build_bvh4", "mcrt_free_bvh4", "mcrt_get_bvh4", "mcrt_row_thresholds", "mcrt_generate_texture", "mcrt_psf_kernels",
           "mcrt_transducer_elements", "mcrt_debug_math", "mcrt_debug_philox", "mcrt_debug_stamps", "mcrt_debug_tail_histograms", "mcrt_debug_set_error", "mcrt_debug_fast_paths", "mcrt_scan_maps",
           "mcrt_group_create", "mcrt_group_destroy", "mcrt_group_size", "mcrt_group_root", "mcrt_group_member", "mcrt_group_shard", "mcrt_group_set_params",
           "mcrt_group_set_bvh_builder", "mcrt_group_upload_scene", "mcrt_group_update_triangles", "mcrt_group_refit_triangles", "mcrt_group_upload_texture",
           "mcrt_group_set_transducer", "mcrt_group_trace_frames", "mcrt_group_trace_frames_poses", "mcrt_group_synchronize", "mcrt_group_last_pass_ms", "mcrt_group_last_scene_seconds"]


def build_library(force=False):
    """Compile libmcrt_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(_HERE, "csrc", f) for f in os.listdir(os.path.join(_HERE, "csrc"))]
    srcs.append(os.path.join(_HERE, "..", "include", "mcrt.h"))
    stale = (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "libmcrt_hip.so"] + (["-B"] if force else []))
    return _SO


def load_library():
    """Load the HIP library.  There is no fallback: a missing library is an error."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(_SO):
        raise McrtError(-4, f"{_SO} is missing: run __graft_entry__.build() (make -C mcray-tracing_amd); there is no CPU fallback")
    L = C.CDLL(_SO)
    L.mcrt_last_error.restype = C.c_char_p
    vp, u32, i32 = C.c_void_p, C.c_uint32, C.c_int
    sig = {
        "mcrt_create": [i32, C.POINTER(vp)], "mcrt_destroy": [vp], "mcrt_set_stream": [vp, vp], "mcrt_synchronize": [vp],
        "mcrt_default_params": [C.POINTER(Params)], "mcrt_set_params": [vp, C.POINTER(Params)], "mcrt_get_params": [vp, C.POINTER(Params)], "mcrt_import_rf": [vp, vp, u32, u32, vp],
        "mcrt_upload_scene": [vp, vp, vp, u32, vp, u32, vp, u32, u32, vp],
        "mcrt_upload_texture": [vp, vp, u32], "mcrt_set_transducer": [vp, vp, vp, u32],
        "mcrt_trace_frame": [vp, u32, u32, u32, vp], "mcrt_trace_frames": [vp, u32, u32, u32, u32, vp], "mcrt_trace_frame_debug": [vp, u32, u32, u32, vp, vp, vp, vp],
        "mcrt_cast_rays": [vp, u32, u32, u32, vp, vp, vp],
        "mcrt_convolve": [vp, vp, u32, u32, vp, u32, vp, u32], "mcrt_envelope": [vp, vp, u32, u32],
        "mcrt_scan_convert": [vp, vp, u32, u32, C.c_double, C.c_double, vp, u32, u32],
        "mcrt_convolve_frames": [vp, vp, u32, u32, u32, vp, u32, vp, u32],
        "mcrt_trace_frames_poses": [vp, u32, u32, u32, u32, vp, vp, vp], "mcrt_envelope_frames": [vp, vp, u32, u32, u32],
        "mcrt_scan_convert_frames": [vp, vp, u32, u32, u32, C.c_double, C.c_double, vp, u32, u32],
        "mcrt_set_bvh_builder": [vp, i32], "mcrt_update_triangles": [vp, vp, u32], "mcrt_refit_triangles": [vp, vp, u32],
        "mcrt_export_rf": [vp, vp, u32, u32, vp], "mcrt_alloc": [vp, C.c_size_t, C.POINTER(vp)], "mcrt_free": [vp, vp],
        "mcrt_memcpy_d2h": [vp, vp, vp, C.c_size_t], "mcrt_memcpy_h2d": [vp, vp, vp, C.c_size_t],
        "mcrt_enable_stats": [vp, i32], "mcrt_get_stats": [vp, C.POINTER(Stats), i32],
        "mcrt_enable_timing": [vp, i32], "mcrt_get_kernel_time": [vp, C.POINTER(C.c_double), C.POINTER(u32), i32], "mcrt_get_kernel_times": [vp, vp, vp, i32],
        "mcrt_build_bvh": [vp, vp, u32, C.POINTER(Bvh)], "mcrt_free_bvh": [C.POINTER(Bvh)], "mcrt_get_bvh": [vp, C.POINTER(Bvh)],
        "mcrt_build_bvh4": [C.POINTER(Bvh), C.POINTER(Bvh4)], "mcrt_free_bvh4": [C.POINTER(Bvh4)], "mcrt_get_bvh4": [vp, C.POINTER(Bvh4)],
        "mcrt_row_thresholds": [C.c_double, u32, vp],
        "mcrt_generate_texture": [vp, u32], "mcrt_psf_kernels": [C.c_float, C.c_float, C.c_float, u32, vp, u32, vp, u32],
        "mcrt_transducer_elements": [u32, C.c_double, C.c_double, vp, vp, vp, vp],
        "mcrt_debug_math": [vp, i32, vp, vp, vp, u32], "mcrt_debug_philox": [vp, vp, vp, vp], "mcrt_debug_stamps": [vp, vp, i32], "mcrt_debug_tail_histograms": [vp, vp, i32], "mcrt_debug_set_error": [vp, u32], "mcrt_debug_fast_paths": [vp, vp],
        "mcrt_scan_maps": [u32, u32, C.c_double, C.c_double, u32, u32, u32, u32, vp, vp],
        "mcrt_group_create": [vp, u32, C.POINTER(vp)], "mcrt_group_destroy": [vp], "mcrt_group_size": [vp], "mcrt_group_root": [vp], "mcrt_group_member": [vp, u32],
        "mcrt_group_shard": [u32, u32, u32, C.POINTER(u32), C.POINTER(u32)], "mcrt_group_set_params": [vp, C.POINTER(Params)], "mcrt_group_set_bvh_builder": [vp, i32],
        "mcrt_group_upload_scene": [vp, vp, vp, u32, vp, u32, vp, u32, u32, vp], "mcrt_group_update_triangles": [vp, vp, u32], "mcrt_group_refit_triangles": [vp, vp, u32],
        "mcrt_group_upload_texture": [vp, vp, u32], "mcrt_group_set_transducer": [vp, vp, vp, u32], "mcrt_group_trace_frames": [vp, u32, u32, vp],
        "mcrt_group_trace_frames_poses": [vp, u32, u32, vp, vp, vp], "mcrt_group_synchronize": [vp], "mcrt_group_last_pass_ms": [vp, vp, vp],
        "mcrt_group_last_scene_seconds": [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)],
    }
    for name, args in sig.items():
        f = getattr(L, name)
        f.argtypes = args
        f.restype = None if name in ("mcrt_free_bvh", "mcrt_free_bvh4") else vp if name in ("mcrt_group_root", "mcrt_group_member") else C.c_int
    _LIB = L
    return L


def check(rc):
    if rc != 0:
        raise McrtError(rc, load_library().mcrt_last_error().decode())


def ptr(a):
    """host pointer of a numpy array / raw device pointer of a torch tensor / int passthrough"""
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    if hasattr(a, "data_ptr"):
        return C.c_void_p(a.data_ptr())
    raise TypeError(type(a))
