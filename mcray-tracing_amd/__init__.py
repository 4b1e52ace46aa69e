"""mcray_tracing_amd -- MI355X-native Monte-Carlo ultrasound ray tracer (hot path of
thepochynsons/MCRay-Tracing) behind the C-ABI of include/mcrt.h.

Python here is plumbing for tests and bench.py (ctypes over libmcrt_hip.so); the product is the
HIP library in csrc/ and the C++ host mirror of the reference API in host/.
"""
from ._lib import load_library, build_library, McrtError, Params, MeshRec, BvhNode, Stats, SEGMENT_DTYPE  # noqa: F401
from .api import Context, Group, shard_range, Simulator, Transducer, Psf, host_build_bvh, host_build_bvh4, host_row_thresholds, host_texture, host_psf, host_transducer, host_scan_maps  # noqa: F401
from . import synth, scene_io  # noqa: F401


def __getattr__(name):          # torch is only needed by the multi-GPU helper
    if name == "dist":
        import importlib
        return importlib.import_module("mcray_tracing_amd.dist")
    raise AttributeError(name)

__all__ = ["load_library", "build_library", "McrtError", "Params", "Context", "Group", "shard_range", "Simulator", "Transducer", "Psf",
           "synth", "scene_io", "host_build_bvh", "host_texture", "host_psf", "host_transducer"]
