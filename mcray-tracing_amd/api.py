"""Python mirror of the reference's host interface for the hot path, over the C-ABI.

  Transducer  <-> transducer<N>            (transducer.h:24-67)
  Psf         <-> psf<ax,lat,elev,res>     (psf.h:34-77)
  Simulator   <-> scene + rf_image + the frame loop body of main.cpp:102-148
  Context     <-> thin 1:1 wrapper of include/mcrt.h
"""
import ctypes as C
import math
import numpy as np

from . import _lib
from ._lib import Params, MeshRec, Stats, Bvh, Bvh4, NODE_DTYPE, SEGMENT_DTYPE, check, ptr, load_library


# ------------------------------------------------------------------ host-side pieces (no GPU)
def host_build_bvh(tri, tri_mesh):
    """-> (nodes structured array [n_nodes] of 64-B nodes, bvh_tri float32 [T,12], max_depth)"""
    L = load_library()
    tri = np.ascontiguousarray(tri, np.float32).reshape(-1, 9)
    tm = np.ascontiguousarray(tri_mesh, np.uint32)
    b = Bvh()
    check(L.mcrt_build_bvh(ptr(tri), ptr(tm), tri.shape[0], C.byref(b)))
    try:
        nodes = np.frombuffer(C.string_at(b.nodes, 64 * b.n_nodes), dtype=NODE_DTYPE).copy()
        btri = np.frombuffer(C.string_at(b.tri, 48 * b.n_tri), dtype=np.float32).reshape(-1, 12).copy()
        depth = int(b.max_depth)
    finally:
        L.mcrt_free_bvh(C.byref(b))
    return nodes, btri, depth


def host_build_bvh4(tri, tri_mesh):
    """-> (BVH2 nodes, leaf-order triangles [T,12], BVH4 nodes uint8 [n4,128], max_stack)"""
    L = load_library()
    tri = np.ascontiguousarray(tri, np.float32).reshape(-1, 9)
    tm = np.ascontiguousarray(tri_mesh, np.uint32)
    b = Bvh(); b4 = Bvh4()
    check(L.mcrt_build_bvh(ptr(tri), ptr(tm), tri.shape[0], C.byref(b)))
    try:
        check(L.mcrt_build_bvh4(C.byref(b), C.byref(b4)))
        nodes = np.frombuffer(C.string_at(b.nodes, 64 * b.n_nodes), dtype=NODE_DTYPE).copy()
        btri = np.frombuffer(C.string_at(b.tri, 48 * b.n_tri), dtype=np.float32).reshape(-1, 12).copy()
        n4 = np.frombuffer(C.string_at(b4.nodes, 128 * b4.n_nodes), dtype=np.uint8).reshape(-1, 128).copy()
        ms = int(b4.max_stack)
    finally:
        L.mcrt_free_bvh(C.byref(b)); L.mcrt_free_bvh4(C.byref(b4))
    return nodes, btri, n4, ms


def host_row_thresholds(row_dt_us, n_rows):
    thr = np.zeros(n_rows + 1, np.float64)
    check(load_library().mcrt_row_thresholds(row_dt_us, n_rows, ptr(thr)))
    return thr


def host_texture(n=256):
    out = np.empty((n, n, n, 2), np.float32)
    check(load_library().mcrt_generate_texture(ptr(out), n))
    return out


def host_psf(freq=4.5, var_x=0.05, var_y=0.2, res_um=145, n_ax=7, n_lat=13):
    ax = np.zeros(n_ax, np.float32); lat = np.zeros(n_lat, np.float32)
    check(load_library().mcrt_psf_kernels(freq, var_x, var_y, res_um, ptr(ax), n_ax, ptr(lat), n_lat))
    return ax, lat


def host_transducer(n_elements, radius_cm, sep_mm, position, angles_deg):
    pos = np.zeros((n_elements, 3), np.float32); d = np.zeros((n_elements, 3), np.float32)
    p = np.asarray(position, np.float32); a = np.asarray(angles_deg, np.float32)
    check(load_library().mcrt_transducer_elements(n_elements, radius_cm, sep_mm, ptr(p), ptr(a), ptr(pos), ptr(d)))
    return pos, d


def host_scan_maps(n_elements, n_rows, radius_mm=30.0, total_angle=1.0471975511965976, max_travel_us=100, speed_of_sound=1500, out_rows=400, out_cols=500):
    """rf_image::create_mapping (rfimage.h:183-215) as the library evaluates it: (map_row, map_col), each [out_rows][out_cols]"""
    mr = np.zeros((out_rows, out_cols), np.float32); mc = np.zeros((out_rows, out_cols), np.float32)
    check(load_library().mcrt_scan_maps(n_elements, n_rows, radius_mm, total_angle, max_travel_us, speed_of_sound, out_rows, out_cols, ptr(mr), ptr(mc)))
    return mr, mc


class Transducer:
    """transducer<N>(frequency, radius, element_separation, position, angles) -- transducer.h:24-62.
    main.cpp:28-29,66: total aperture 60 deg on a 3 cm radius; separation = amplitude * radius / N."""

    def __init__(self, n_elements=512, frequency=4.5, radius_cm=3.0, amplitude_deg=60.0, position=(0, 0, 0), angles_deg=(0, 0, 0), separation_mm=None):
        self.n_elements = n_elements
        self.frequency = frequency
        self.radius_cm = radius_cm
        self.amplitude_rad = (amplitude_deg * math.pi * 1.0) / 180.0
        if separation_mm is None:
            # millimeter_t sep = amplitude.to<float>() * radius / N  (float * centimeter_t -> cm, then -> mm: *10)
            separation_mm = ((float(np.float32(self.amplitude_rad)) * radius_cm) / n_elements) * 10.0
        self.separation_mm = separation_mm
        self.position = tuple(float(x) for x in position)
        self.angles = tuple(float(x) for x in angles_deg)
        self.update()

    def update(self):
        self.pos, self.dir = host_transducer(self.n_elements, self.radius_cm, self.separation_mm, self.position, self.angles)

    def element(self, i):
        return self.pos[i], self.dir[i]


class Psf:
    """psf<axial,lateral,elevation,resolution_um>{freq, var_x, var_y, var_z} -- psf.h:34-58"""

    def __init__(self, freq=4.5, var_x=0.05, var_y=0.2, var_z=0.1, axial_size=7, lateral_size=13, resolution_um=145):
        self.axial_kernel, self.lateral_kernel = host_psf(freq, var_x, var_y, resolution_um, axial_size, lateral_size)


# ------------------------------------------------------------------ C-ABI context
class Context:
    def __init__(self, device=0, _borrowed=None):
        self.L = load_library()
        self.owned = _borrowed is None
        if self.owned:
            h = C.c_void_p()
            check(self.L.mcrt_create(device, C.byref(h)))
        else:                                   # a context that belongs to a Group (root / member): not destroyed here
            h = C.c_void_p(_borrowed)
        self.h = h
        self.params = Params()
        check(self.L.mcrt_default_params(C.byref(self.params)))

    def close(self):
        if getattr(self, "h", None):
            if self.owned:
                self.L.mcrt_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_params(self, **kw):
        for k, v in kw.items():
            if not hasattr(self.params, k):
                raise AttributeError(k)
            setattr(self.params, k, v)
        check(self.L.mcrt_set_params(self.h, C.byref(self.params)))

    def set_stream(self, stream_ptr):
        check(self.L.mcrt_set_stream(self.h, C.c_void_p(stream_ptr) if stream_ptr else None))

    def synchronize(self):
        check(self.L.mcrt_synchronize(self.h))

    def debug_fast_paths(self):
        """(fast voxel quotient, branch-free voxel cell, entries of k_march's padded LDS image or 0) of the last traced frame"""
        out = (C.c_uint32 * 4)()
        check(self.L.mcrt_debug_fast_paths(self.h, out))
        return bool(out[0]), bool(out[1]), int(out[2])

    def debug_set_error(self, bits):
        """test hook: mark the context as an abandoned launch would (mcrt_debug_set_error)"""
        check(self.L.mcrt_debug_set_error(self.h, int(bits)))

    def upload_scene(self, sd):
        meshes = (MeshRec * len(sd.meshes))(*[MeshRec(a, b, c, 0) for a, b, c in sd.meshes])
        sp = np.asarray(sd.spacing, np.float32)
        check(self.L.mcrt_upload_scene(self.h, ptr(sd.tri), ptr(sd.tri_mesh), sd.n_tri, C.cast(meshes, C.c_void_p), len(sd.meshes),
                                       ptr(sd.materials), sd.materials.shape[0], sd.start_mat, ptr(sp)))

    def set_bvh_builder(self, builder):
        """'sah' (host, default) or 'lbvh' (built on the GPU); applies to the next upload_scene / update_triangles"""
        kind = {"sah": 0, "lbvh": 1}[builder] if isinstance(builder, str) else int(builder)
        check(self.L.mcrt_set_bvh_builder(self.h, kind))

    def update_triangles(self, tri):
        """new vertex positions [T,9] (numpy array, or a CUDA torch tensor) for the uploaded scene's triangles"""
        if isinstance(tri, np.ndarray):
            tri = np.ascontiguousarray(tri, np.float32).reshape(-1, 9)
            n = tri.shape[0]
        else:
            n = tri.numel() // 9
        check(self.L.mcrt_update_triangles(self.h, ptr(tri), n))

    def refit_triangles(self, tri):
        """new vertex positions [T,9] for the uploaded triangles, keeping the tree: boxes are refitted on the GPU"""
        if isinstance(tri, np.ndarray):
            tri = np.ascontiguousarray(tri, np.float32).reshape(-1, 9)
            n = tri.shape[0]
        else:
            n = tri.numel() // 9
        check(self.L.mcrt_refit_triangles(self.h, ptr(tri), n))

    def upload_texture(self, vox=None, n=256):
        if vox is not None:
            vox = np.ascontiguousarray(vox, np.float32)
        check(self.L.mcrt_upload_texture(self.h, ptr(vox), n))

    def set_transducer(self, pos, d):
        pos = np.ascontiguousarray(pos, np.float32); d = np.ascontiguousarray(d, np.float32)
        check(self.L.mcrt_set_transducer(self.h, ptr(pos), ptr(d), pos.shape[0]))

    def get_bvh(self):
        b = Bvh()
        check(self.L.mcrt_get_bvh(self.h, C.byref(b)))
        # (a tree built on the device has no BVH2: n_nodes == 0)
        nodes = np.frombuffer(C.string_at(b.nodes, 64 * b.n_nodes), dtype=NODE_DTYPE).copy() if b.n_nodes else np.zeros(0, NODE_DTYPE)
        btri = np.frombuffer(C.string_at(b.tri, 48 * b.n_tri), dtype=np.float32).reshape(-1, 12).copy()
        return nodes, btri, int(b.max_depth)

    def get_bvh4(self):
        b = Bvh4()
        check(self.L.mcrt_get_bvh4(self.h, C.byref(b)))
        return np.frombuffer(C.string_at(b.nodes, 128 * b.n_nodes), dtype=np.uint8).reshape(-1, 128).copy(), int(b.max_stack)

    # device memory
    def alloc(self, nbytes):
        p = C.c_void_p()
        check(self.L.mcrt_alloc(self.h, nbytes, C.byref(p)))
        return p.value

    def free(self, dev):
        check(self.L.mcrt_free(self.h, C.c_void_p(dev)))

    def d2h(self, dev, shape, dtype=np.float32):
        out = np.empty(shape, dtype)
        check(self.L.mcrt_memcpy_d2h(self.h, ptr(out), ptr(dev), out.nbytes))
        return out

    def h2d(self, dev, arr):
        arr = np.ascontiguousarray(arr)
        check(self.L.mcrt_memcpy_h2d(self.h, ptr(dev), ptr(arr), arr.nbytes))

    # frame
    def trace_frame(self, frame_id, rf_dev, e_begin=0, e_end=None):
        e_end = self.params.n_elements if e_end is None else e_end
        check(self.L.mcrt_trace_frame(self.h, frame_id, e_begin, e_end, ptr(rf_dev)))

    def trace_frames(self, frame_id, n_frames, rf_dev, e_begin=0, e_end=None):
        """n_frames consecutive frames in one pass; rf_dev holds [n_frames][e_end-e_begin][R] floats"""
        e_end = self.params.n_elements if e_end is None else e_end
        check(self.L.mcrt_trace_frames(self.h, frame_id, n_frames, e_begin, e_end, ptr(rf_dev)))

    def trace_frames_poses(self, frame_id, pos, dirs, rf_dev, e_begin=0, e_end=None, n_frames=None):
        """a pass with a probe pose per frame: pos / dirs [F][E][3] (numpy arrays, CUDA torch tensors, or raw device pointers with
        n_frames given); rf_dev [F][e_end-e_begin][R]"""
        e_end = self.params.n_elements if e_end is None else e_end
        if isinstance(pos, np.ndarray):
            pos = np.ascontiguousarray(pos, np.float32); dirs = np.ascontiguousarray(dirs, np.float32)
        if n_frames is None:
            n_frames = pos.shape[0]
            assert tuple(pos.shape) == (n_frames, self.params.n_elements, 3) and tuple(dirs.shape) == tuple(pos.shape)
        check(self.L.mcrt_trace_frames_poses(self.h, frame_id, n_frames, e_begin, e_end, ptr(pos), ptr(dirs), ptr(rf_dev)))

    def trace_frame_debug(self, frame_id, rf_dev, e_begin=0, e_end=None, want_hits=True, want_segs=False):
        e_end = self.params.n_elements if e_end is None else e_end
        ne, S, B = e_end - e_begin, self.params.n_samples, self.params.max_depth
        hits = np.full((ne, S, B), -3, np.int32) if want_hits else None
        segs = np.zeros((ne, S, B), SEGMENT_DTYPE) if want_segs else None
        cnt = np.zeros((ne, S), np.uint32) if want_segs else None
        check(self.L.mcrt_trace_frame_debug(self.h, frame_id, e_begin, e_end, ptr(rf_dev), ptr(hits), ptr(segs), ptr(cnt)))
        return hits, segs, cnt

    def cast_rays(self, frame_id, e_begin=0, e_end=None, want_hits=True):
        e_end = self.params.n_elements if e_end is None else e_end
        ne, S, B = e_end - e_begin, self.params.n_samples, self.params.max_depth
        segs = np.zeros((ne, S, B), SEGMENT_DTYPE); cnt = np.zeros((ne, S), np.uint32)
        hits = np.full((ne, S, B), -3, np.int32) if want_hits else None
        check(self.L.mcrt_cast_rays(self.h, frame_id, e_begin, e_end, ptr(segs), ptr(cnt), ptr(hits)))
        return segs, cnt, hits

    def convolve_frames(self, rf_dev, n_frames, n_elements, n_rows, axial, lateral):
        """rf_image::convolve on the [n_frames][E][R] images of a trace_frames pass, one launch per convolution pass"""
        ax = np.ascontiguousarray(axial, np.float32); lat = np.ascontiguousarray(lateral, np.float32)
        check(self.L.mcrt_convolve_frames(self.h, ptr(rf_dev), n_frames, n_elements, n_rows, ptr(ax), ax.size, ptr(lat), lat.size))

    def convolve(self, rf_dev, n_elements, n_rows, axial, lateral):
        ax = np.ascontiguousarray(axial, np.float32); lat = np.ascontiguousarray(lateral, np.float32)
        check(self.L.mcrt_convolve(self.h, ptr(rf_dev), n_elements, n_rows, ptr(ax), ax.size, ptr(lat), lat.size))

    def envelope(self, rf_dev, n_elements, n_rows):
        check(self.L.mcrt_envelope(self.h, ptr(rf_dev), n_elements, n_rows))

    def envelope_frames(self, rf_dev, n_frames, n_elements, n_rows):
        check(self.L.mcrt_envelope_frames(self.h, ptr(rf_dev), n_frames, n_elements, n_rows))

    def scan_convert_frames(self, rf_dev, n_frames, n_elements, n_rows, out_dev, radius_mm=30.0, total_angle=1.0471975511965976, out_rows=400, out_cols=500):
        check(self.L.mcrt_scan_convert_frames(self.h, ptr(rf_dev), n_frames, n_elements, n_rows, radius_mm, total_angle, ptr(out_dev), out_rows, out_cols))

    def scan_convert(self, rf_dev, n_elements, n_rows, out_dev, radius_mm=30.0, total_angle=1.0471975511965976, out_rows=400, out_cols=500):
        check(self.L.mcrt_scan_convert(self.h, ptr(rf_dev), n_elements, n_rows, radius_mm, total_angle, ptr(out_dev), out_rows, out_cols))

    def export_rf(self, rf_dev, n_elements, n_rows):
        out = np.empty((n_rows, n_elements), np.float32)
        check(self.L.mcrt_export_rf(self.h, ptr(rf_dev), n_elements, n_rows, ptr(out)))
        return out

    # instrumentation
    def enable_stats(self, on=True):
        check(self.L.mcrt_enable_stats(self.h, int(on)))

    def get_stats(self, reset=True):
        s = Stats()
        check(self.L.mcrt_get_stats(self.h, C.byref(s), int(reset)))
        return s.as_dict()

    def enable_timing(self, on=True):
        check(self.L.mcrt_enable_timing(self.h, int(on)))

    def kernel_times(self, reset=True):
        """{"walk" | "shade" | "march": (average ms per launch, launches)} since the last reset (shade / march only under enable_timing(2))"""
        ms = (C.c_double * 3)(); n = (C.c_uint32 * 3)()
        check(self.L.mcrt_get_kernel_times(self.h, ms, n, int(reset)))
        return {k: (ms[i], n[i]) for i, k in enumerate(("walk", "shade", "march"))}

    def kernel_time(self, reset=True):
        ms = C.c_double(); n = C.c_uint32()
        check(self.L.mcrt_get_kernel_time(self.h, C.byref(ms), C.byref(n), int(reset)))
        return ms.value, n.value

    def debug_math(self, op, x, y=None):
        x = np.ascontiguousarray(x, np.float64); out = np.empty_like(x)
        if y is not None:
            y = np.ascontiguousarray(y, np.float64)
        check(self.L.mcrt_debug_math(self.h, op, ptr(x), ptr(y), ptr(out), x.size))
        return out

    def debug_philox(self, ctr, key):
        c = np.asarray(ctr, np.uint32); k = np.asarray(key, np.uint32); o = np.zeros(4, np.uint32)
        check(self.L.mcrt_debug_philox(self.h, ptr(c), ptr(k), ptr(o)))
        return o


# ------------------------------------------------------------------ several GPUs behind one call (mcrt_group_*)
def shard_range(rank, n_ranks, n_elements):
    """the contiguous scan-line block of `rank` (mcrt_group_shard; dist.shard_range is the same rule in Python)"""
    b = C.c_uint32(); e = C.c_uint32()
    check(load_library().mcrt_group_shard(rank, n_ranks, n_elements, C.byref(b), C.byref(e)))
    return int(b.value), int(e.value)


class Group:
    """mcrt_group: one tracing context per listed device (a device may repeat), scan-lines cut into contiguous shards, the blocks
    gathered on devices[0].  `root` is the context that owns the gathered frames (post-processing, alloc, exports)."""

    def __init__(self, devices):
        self.L = load_library()
        devs = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        check(self.L.mcrt_group_create(C.cast(devs, C.c_void_p), len(devices), C.byref(h)))
        self.h = h
        self.size = int(self.L.mcrt_group_size(self.h))
        self.root = Context(_borrowed=self.L.mcrt_group_root(self.h))
        self.members = [Context(_borrowed=self.L.mcrt_group_member(self.h, r)) for r in range(self.size)]
        self.params = Params()
        check(self.L.mcrt_default_params(C.byref(self.params)))

    def close(self):
        if getattr(self, "h", None):
            self.L.mcrt_group_destroy(self.h)
            self.h = None
            self.root.h = None
            for m in self.members:
                m.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_params(self, **kw):
        for k, v in kw.items():
            if not hasattr(self.params, k):
                raise AttributeError(k)
            setattr(self.params, k, v)
        try:
            check(self.L.mcrt_group_set_params(self.h, C.byref(self.params)))
        finally:       # (refused parameters leave every context on the old ones: read back what the group really holds)
            for c in [self.root] + self.members:
                check(self.L.mcrt_get_params(c.h, C.byref(c.params)))
            check(self.L.mcrt_get_params(self.root.h, C.byref(self.params)))

    def set_bvh_builder(self, builder):
        kind = {"sah": 0, "lbvh": 1}[builder] if isinstance(builder, str) else int(builder)
        check(self.L.mcrt_group_set_bvh_builder(self.h, kind))

    def upload_scene(self, sd):
        meshes = (MeshRec * len(sd.meshes))(*[MeshRec(a, b, c, 0) for a, b, c in sd.meshes])
        sp = np.asarray(sd.spacing, np.float32)
        check(self.L.mcrt_group_upload_scene(self.h, ptr(sd.tri), ptr(sd.tri_mesh), sd.n_tri, C.cast(meshes, C.c_void_p), len(sd.meshes),
                                             ptr(sd.materials), sd.materials.shape[0], sd.start_mat, ptr(sp)))

    def update_triangles(self, tri):
        tri = np.ascontiguousarray(tri, np.float32).reshape(-1, 9)
        check(self.L.mcrt_group_update_triangles(self.h, ptr(tri), tri.shape[0]))

    def refit_triangles(self, tri):
        tri = np.ascontiguousarray(tri, np.float32).reshape(-1, 9)
        check(self.L.mcrt_group_refit_triangles(self.h, ptr(tri), tri.shape[0]))

    def upload_texture(self, vox=None, n=256):
        if vox is not None:
            vox = np.ascontiguousarray(vox, np.float32)
        check(self.L.mcrt_group_upload_texture(self.h, ptr(vox), n))

    def set_transducer(self, pos, d):
        pos = np.ascontiguousarray(pos, np.float32); d = np.ascontiguousarray(d, np.float32)
        check(self.L.mcrt_group_set_transducer(self.h, ptr(pos), ptr(d), pos.shape[0]))

    def trace_frames(self, frame_id, n_frames, rf_dev):
        """rf_dev: [n_frames][E][R] on devices[0]; complete on the root context's stream"""
        check(self.L.mcrt_group_trace_frames(self.h, frame_id, n_frames, ptr(rf_dev)))

    def trace_frames_poses(self, frame_id, pos, dirs, rf_dev):
        pos = np.ascontiguousarray(pos, np.float32); dirs = np.ascontiguousarray(dirs, np.float32)
        assert tuple(pos.shape) == (pos.shape[0], self.params.n_elements, 3) and tuple(dirs.shape) == tuple(pos.shape)
        check(self.L.mcrt_group_trace_frames_poses(self.h, frame_id, pos.shape[0], ptr(pos), ptr(dirs), ptr(rf_dev)))

    def synchronize(self):
        check(self.L.mcrt_group_synchronize(self.h))

    def last_scene_seconds(self):
        """(host SAH build -- once, on the calling thread --, the ranks' concurrent uploads) of the last upload_scene / update_triangles"""
        b, u = C.c_double(), C.c_double()
        check(self.L.mcrt_group_last_scene_seconds(self.h, C.byref(b), C.byref(u)))
        return b.value, u.value

    def last_pass_ms(self):
        t = np.zeros(self.size, np.float32); c = np.zeros(self.size, np.float32)
        check(self.L.mcrt_group_last_pass_ms(self.h, ptr(t), ptr(c)))
        return t, c


# ------------------------------------------------------------------ frame-level mirror of main.cpp:92-152
class Simulator:
    """scene + transducer + rf_image of the reference, driven frame by frame.

        sim = Simulator(scene_data, transducer, n_samples=5)
        rf = sim.frame(0)                 # clear -> cast_rays -> accumulate -> convolve, returns [R][E] host image
    """

    def __init__(self, scene_data, transducer, n_samples=5, n_rows=None, device=0, seed=0x5EED, psf=None, texture=None,
                 max_depth=10, sanitize_tir=0, tex_n=256, bvh_builder="sah"):
        self.ctx = Context(device)
        self.ctx.set_bvh_builder(bvh_builder)
        self.tr = transducer
        E = transducer.n_elements
        self.ctx.set_params(n_elements=E, n_samples=n_samples, frequency=transducer.frequency, seed=seed, max_depth=max_depth,
                            sanitize_tir=sanitize_tir, tex_n=tex_n, **({"n_rows": n_rows} if n_rows else {}))
        self.E, self.R, self.S = E, self.ctx.params.n_rows, n_samples
        self.ctx.upload_scene(scene_data)
        self.ctx.upload_texture(texture, tex_n)
        self.ctx.set_transducer(transducer.pos, transducer.dir)
        self.psf = psf or Psf(freq=transducer.frequency)
        self.rf_dev = self.ctx.alloc(E * self.R * 4)

    def close(self):
        if self.ctx.h:
            self.ctx.free(self.rf_dev)
            self.ctx.close()

    def trace(self, frame_id=0):
        self.ctx.trace_frame(frame_id, self.rf_dev)

    def convolve(self):
        self.ctx.convolve(self.rf_dev, self.E, self.R, self.psf.axial_kernel, self.psf.lateral_kernel)

    def frame(self, frame_id=0, convolve=True):
        self.trace(frame_id)
        if convolve:
            self.convolve()
        return self.ctx.export_rf(self.rf_dev, self.E, self.R)
