"""ctypes binding of the CPU ORACLE (oracle/libmcrt_oracle.so).

TEST INFRASTRUCTURE ONLY: import from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Consts(C.Structure):
    _fields_ = [("axial_res_mm", C.c_double), ("axial_res_f", C.c_float), ("time_step_us", C.c_double),
                ("row_dt_us", C.c_double), ("max_travel_us", C.c_double), ("axial_res_um", C.c_uint32),
                ("max_rows", C.c_uint32)]


class Mesh(C.Structure):
    _fields_ = [("mat_inside", C.c_uint32), ("mat_outside", C.c_uint32), ("vascular", C.c_uint32), ("_pad", C.c_uint32)]


class Scene(C.Structure):
    _fields_ = [("n_tri", C.c_uint32), ("tri", C.c_void_p), ("tri_mesh", C.c_void_p),
                ("n_mesh", C.c_uint32), ("mesh", C.c_void_p),
                ("n_mat", C.c_uint32), ("mat", C.c_void_p),
                ("start_mat", C.c_uint32), ("spacing", C.c_float * 3),
                ("n_nodes", C.c_uint32), ("nodes", C.c_void_p), ("bvh_tri", C.c_void_p), ("pad_abs", C.c_float), ("n_nodes4", C.c_uint32), ("nodes4", C.c_void_p)]


class Params(C.Structure):
    _fields_ = [("n_elements", C.c_uint32), ("n_samples", C.c_uint32), ("max_depth", C.c_uint32), ("n_rows", C.c_uint32),
                ("frequency", C.c_float), ("intensity_epsilon", C.c_float), ("initial_intensity", C.c_float),
                ("ray_start_offset", C.c_float), ("sos", C.c_uint32), ("depth_cm", C.c_double),
                ("seed", C.c_uint32), ("sanitize_tir", C.c_uint32), ("tex_n", C.c_uint32), ("tex_res", C.c_float)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("queries", "nodes_visited", "tris_tested", "segments", "rf_steps", "hits")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class RayState(C.Structure):
    """orc_ray_state: outside = -1 (nullptr), -2 (aliases the ray's own media, quirk 2) or a material index"""
    _fields_ = [("origin", C.c_float * 3), ("dir", C.c_float * 3), ("media", C.c_int32), ("outside", C.c_int32),
                ("intensity", C.c_float), ("frequency", C.c_float), ("dist_mm", C.c_double)]


class HitDebug(C.Structure):
    _fields_ = [("reflected_intensity", C.c_float), ("_pad0", C.c_uint32), ("returned", RayState),
                ("random_angle", C.c_float), ("random_normal", C.c_float * 3), ("incidence", C.c_float), ("refr_ratio", C.c_float),
                ("refraction_angle", C.c_float), ("refr_dir", C.c_float * 3), ("refl_dir", C.c_float * 3),
                ("intensity_refl", C.c_float), ("intensity_refr", C.c_float), ("refraction_factor", C.c_float), ("reflection_factor", C.c_float),
                ("u_pc", C.c_double), ("u_x", C.c_double),
                ("tir", C.c_int32), ("chose_reflection", C.c_int32), ("mat_after", C.c_int32), ("after_vasc", C.c_int32),
                ("ruv_attempts", C.c_uint32), ("_pad1", C.c_uint32)]


OUT_NONE, OUT_SELF = -1, -2
COUNTER_NAMES = ("pad_rule_rejects_in_bounds", "pad_rule_rejects_anywhere", "echo_guard_trips", "ruv_retries", "ruv_giveups", "tir_hits", "nan_echoes")

SEGMENT_DTYPE = np.dtype([("from", "<f4", 3), ("to", "<f4", 3), ("dir", "<f4", 3),
                          ("reflected_intensity", "<f4"), ("initial_intensity", "<f4"), ("attenuation", "<f4"),
                          ("distance_traveled", "<f8"), ("media", "<i4"), ("tri", "<i4")])
assert SEGMENT_DTYPE.itemsize == 64


def build(force=False):
    so = os.path.join(_HERE, "libmcrt_oracle.so")
    src = os.path.join(_HERE, "mcrt_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libmcrt_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_log_d.restype = C.c_double; L.orc_log_d.argtypes = [C.c_double]
        L.orc_exp_d.restype = C.c_double; L.orc_exp_d.argtypes = [C.c_double]
        L.orc_pow_d.restype = C.c_double; L.orc_pow_d.argtypes = [C.c_double, C.c_double]
        L.orc_logf.restype = C.c_float; L.orc_logf.argtypes = [C.c_float]
        L.orc_expf.restype = C.c_float; L.orc_expf.argtypes = [C.c_float]
        L.orc_powf.restype = C.c_float; L.orc_powf.argtypes = [C.c_float, C.c_float]
        L.orc_sincos_d.restype = None; L.orc_sincos_d.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_u53.restype = C.c_double; L.orc_u53.argtypes = [C.c_uint32, C.c_uint32]
        L.orc_closest_hit.restype = C.c_int32
        L.orc_closest_hit.argtypes = [C.POINTER(Scene), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_trace_frame.restype = None
        L.orc_trace_frame.argtypes = [C.POINTER(Scene), C.POINTER(Params), C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_debug_power_cosine.restype = C.c_float; L.orc_debug_power_cosine.argtypes = [C.c_int, C.c_double]
        L.orc_debug_random_unit_vector.restype = C.c_uint32
        L.orc_debug_random_unit_vector.argtypes = [C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]
        L.orc_debug_hit_boundary.restype = None
        L.orc_debug_hit_boundary.argtypes = [C.POINTER(Scene), C.POINTER(Params), C.POINTER(RayState), C.c_void_p, C.c_void_p, C.c_uint32,
                                             C.c_void_p, C.POINTER(HitDebug)]
        L.orc_debug_ray_segment.restype = C.c_float
        L.orc_debug_ray_segment.argtypes = [C.POINTER(Scene), C.POINTER(Params), C.POINTER(RayState), C.c_void_p, C.c_void_p]
        L.orc_debug_travel.restype = C.c_double; L.orc_debug_travel.argtypes = [C.POINTER(Scene), C.POINTER(RayState), C.c_void_p]
        L.orc_debug_thickness.restype = C.c_float; L.orc_debug_thickness.argtypes = [C.c_float, C.c_void_p]
        L.orc_debug_accumulate_segment.restype = C.c_uint64
        L.orc_debug_accumulate_segment.argtypes = [C.POINTER(Scene), C.POINTER(Params), C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_debug_counting.restype = None; L.orc_debug_counting.argtypes = [C.c_void_p, C.c_int]
        L.orc_debug_counters.restype = None; L.orc_debug_counters.argtypes = [C.c_void_p]
        _LIB = L
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def set_math_mode(mode):
    lib().orc_set_math_mode(C.c_int(mode))


def philox(ctr, key):
    c = np.asarray(ctr, dtype=np.uint32); k = np.asarray(key, dtype=np.uint32); o = np.zeros(4, np.uint32)
    lib().orc_philox4x32_10(_p(c), _p(k), _p(o))
    return o


def u53(hi, lo):
    return lib().orc_u53(C.c_uint32(int(hi)), C.c_uint32(int(lo)))


def rng_block(rng, block):
    """the contract's two uniforms of `block` for rng = (seed, frame, element, sample, bounce)  """
    o = philox([rng[2], rng[3], rng[4], block], [rng[0], rng[1]])
    return u53(o[0], o[1]), u53(o[2], o[3])


def ray_state(origin, direction, media, outside=OUT_NONE, intensity=1.0, frequency=4.5, dist_mm=0.0):
    r = RayState()
    r.origin = (C.c_float * 3)(*[float(x) for x in origin]); r.dir = (C.c_float * 3)(*[float(x) for x in direction])
    r.media = int(media); r.outside = int(outside); r.intensity = float(intensity); r.frequency = float(frequency); r.dist_mm = float(dist_mm)
    return r


def power_cosine(v, number):
    return lib().orc_debug_power_cosine(int(v), float(number))


def random_unit_vector(v, cos_theta, rng):
    vv = np.asarray(v, np.float32); g = np.asarray(rng, np.uint32); w = np.zeros(3, np.float32)
    n = lib().orc_debug_random_unit_vector(_p(vv), C.c_float(cos_theta), _p(g), _p(w))
    return w, int(n)


def thickness(sigma, rng):
    g = np.asarray(rng, np.uint32)
    return lib().orc_debug_thickness(C.c_float(sigma), _p(g))


def counters():
    o = np.zeros(8, np.uint64)
    lib().orc_debug_counters(_p(o))
    return {n: int(o[i]) for i, n in enumerate(COUNTER_NAMES)}


def math_vec(name, x, y=None):
    """Evaluate a scalar contract-math function over an array (slow path, test only)."""
    L = lib(); f = getattr(L, name)
    if y is None:
        return np.array([f(v) for v in x.tolist()], dtype=x.dtype)
    return np.array([f(a, b) for a, b in zip(x.tolist(), y.tolist())], dtype=x.dtype)


def sincos(a):
    s = C.c_double(); c = C.c_double()
    lib().orc_sincos_d(float(a), C.byref(s), C.byref(c))
    return s.value, c.value


def texture(n=256):
    out = np.empty((n, n, n, 2), dtype=np.float32)
    lib().orc_texture_generate(_p(out), C.c_uint32(n))
    return out


def psf(freq=4.5, var_x=0.05, var_y=0.2, res_um=145, n_ax=7, n_lat=13):
    ax = np.zeros(n_ax, np.float32); lat = np.zeros(n_lat, np.float32)
    lib().orc_psf(C.c_float(freq), C.c_float(var_x), C.c_float(var_y), C.c_uint32(res_um), _p(ax), C.c_uint32(n_ax), _p(lat), C.c_uint32(n_lat))
    return ax, lat


def transducer(n_elem, radius_cm, sep_mm, position, angles_deg):
    pos = np.zeros((n_elem, 3), np.float32); d = np.zeros((n_elem, 3), np.float32)
    p = np.asarray(position, np.float32); a = np.asarray(angles_deg, np.float32)
    lib().orc_transducer(C.c_uint32(n_elem), C.c_double(radius_cm), C.c_double(sep_mm), _p(p), _p(a), _p(pos), _p(d))
    return pos, d


def place_vertices(v, scaling, deltas, origin):
    v = np.ascontiguousarray(v, np.float32).copy()
    dl = np.asarray(deltas, np.float32); og = np.asarray(origin, np.float32)
    lib().orc_place_vertices(_p(v), C.c_uint32(v.size // 3), C.c_float(scaling), _p(dl), _p(og))
    return v


def constants(freq=4.5, sos=1500, depth_cm=15.0):
    c = Consts()
    lib().orc_constants(C.c_float(freq), C.c_uint32(sos), C.c_double(depth_cm), C.byref(c))
    return c


def default_params(**kw):
    p = Params()
    lib().orc_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


class OracleScene:
    """Owns the numpy arrays behind an orc_scene."""

    def __init__(self, tri, tri_mesh, meshes, materials, start_mat, spacing=(1.0, 1.0, 1.0), bvh=None):
        self.tri = np.ascontiguousarray(tri, np.float32).reshape(-1, 9)
        self.tri_mesh = np.ascontiguousarray(tri_mesh, np.uint32)
        self.mesh = (Mesh * len(meshes))(*[Mesh(int(a), int(b), int(c), 0) for a, b, c in meshes])
        self.mat = np.ascontiguousarray(materials, np.float32).reshape(-1, 8)
        self.bvh_nodes = self.bvh_tri = self.bvh4_nodes = None
        s = Scene()
        s.n_tri = self.tri.shape[0]; s.tri = self.tri.ctypes.data; s.tri_mesh = self.tri_mesh.ctypes.data
        s.n_mesh = len(meshes); s.mesh = C.cast(self.mesh, C.c_void_p)
        s.n_mat = self.mat.shape[0]; s.mat = self.mat.ctypes.data
        s.start_mat = int(start_mat)
        s.spacing = (C.c_float * 3)(*[float(x) for x in spacing])
        s.n_nodes = 0; s.nodes = None; s.bvh_tri = None; s.n_nodes4 = 0; s.nodes4 = None
        lib().orc_pad_abs.restype = C.c_float
        s.pad_abs = lib().orc_pad_abs(_p(self.tri), C.c_uint32(self.tri.shape[0]))
        self.c = s
        if bvh is not None:
            self.set_bvh(*bvh)

    def set_bvh(self, nodes, bvh_tri):
        """nodes: uint8/structured array of 64-byte nodes built by the PRODUCT; bvh_tri: [T][12] float32."""
        self.bvh_nodes = np.ascontiguousarray(nodes)
        self.bvh_tri = np.ascontiguousarray(bvh_tri, np.float32)
        assert self.bvh_nodes.nbytes % 64 == 0
        self.c.n_nodes = self.bvh_nodes.nbytes // 64
        self.c.nodes = self.bvh_nodes.ctypes.data
        self.c.bvh_tri = self.bvh_tri.ctypes.data

    def set_bvh4(self, nodes4, bvh_tri=None):
        """nodes4: array of 128-byte BVH4 nodes built by the PRODUCT (use_bvh=2); triangles are shared with the BVH2"""
        self.bvh4_nodes = np.ascontiguousarray(nodes4)
        assert self.bvh4_nodes.nbytes % 128 == 0
        if bvh_tri is not None:
            self.bvh_tri = np.ascontiguousarray(bvh_tri, np.float32)
            self.c.bvh_tri = self.bvh_tri.ctypes.data
        self.c.n_nodes4 = self.bvh4_nodes.nbytes // 128
        self.c.nodes4 = self.bvh4_nodes.ctypes.data

    # ---- test entry points into the physics (ray.cpp) ----
    def hit_boundary(self, params, ray, hit_point, normal, mesh, rng):
        hp = np.asarray(hit_point, np.float32); n = np.asarray(normal, np.float32); g = np.asarray(rng, np.uint32)
        out = HitDebug()
        lib().orc_debug_hit_boundary(C.byref(self.c), C.byref(params), C.byref(ray), _p(hp), _p(n), C.c_uint32(mesh), _p(g), C.byref(out))
        return out

    def ray_segment(self, params, ray):
        f = np.zeros(3, np.float32); t = np.zeros(3, np.float32)
        L = lib().orc_debug_ray_segment(C.byref(self.c), C.byref(params), C.byref(ray), _p(f), _p(t))
        return L, f, t

    def travel(self, ray, to_point):
        t = np.asarray(to_point, np.float32)
        return lib().orc_debug_travel(C.byref(self.c), C.byref(ray), _p(t))

    def accumulate_segment(self, params, tex, seg):
        """seg: one SEGMENT_DTYPE record -> (rf[R] float32 in reference order, RF steps)"""
        sg = np.ascontiguousarray(np.asarray(seg, SEGMENT_DTYPE).reshape(1))
        rf = np.zeros(params.n_rows, np.float32)
        tex = np.ascontiguousarray(tex, np.float32)
        n = lib().orc_debug_accumulate_segment(C.byref(self.c), C.byref(params), _p(tex), _p(sg), _p(rf))
        return rf, int(n)

    def counting(self, on=True):
        lib().orc_debug_counting(C.cast(C.byref(self.c), C.c_void_p) if on else None, 1 if on else 0)

    def closest_hit(self, frm, to, use_bvh=False):
        f = np.asarray(frm, np.float32); t = np.asarray(to, np.float32)
        frac = np.zeros(1, np.float32); n = np.zeros(3, np.float32); p = np.zeros(3, np.float32)
        st = Stats()
        tri = lib().orc_closest_hit(C.byref(self.c), _p(f), _p(t), int(use_bvh), _p(frac), _p(n), _p(p), C.cast(C.byref(st), C.c_void_p))
        return tri, float(frac[0]), n, p, st.as_dict()

    def trace_frame(self, params, el_pos, el_dir, tex, frame_id=0, e_begin=0, e_end=None, use_bvh=False, n_threads=1,
                    want_hits=True, want_segs=False, want_ref=True, want_fix=True, want_ref64=False):
        E = params.n_elements if e_end is None else e_end
        e_end = E
        ne = e_end - e_begin; S = params.n_samples; B = params.max_depth; R = params.n_rows
        el_pos = np.ascontiguousarray(el_pos, np.float32); el_dir = np.ascontiguousarray(el_dir, np.float32)
        tex = np.ascontiguousarray(tex, np.float32)
        out = {}
        hits = np.full((ne, S, B), -2, np.int32) if want_hits else None
        segs = np.zeros((ne, S, B), SEGMENT_DTYPE) if want_segs else None
        segc = np.zeros((ne, S), np.uint32) if want_segs else None
        rf_ref = np.zeros((R, ne), np.float32) if want_ref else None
        rf_fix = np.zeros((ne, R), np.int64) if want_fix else None
        rf_flg = np.zeros((ne, R), np.uint8) if want_fix else None
        st = Stats()
        # the reference-order sum carried in double beside the float one (what the float running sum loses: tests at large S)
        rf_ref64 = np.zeros((R, ne), np.float64) if (want_ref and want_ref64) else None
        lib().orc_set_ref64.argtypes = [C.c_void_p]; lib().orc_set_ref64.restype = None
        lib().orc_set_ref64(_p(rf_ref64))
        try:
            self._trace(params, el_pos, el_dir, tex, frame_id, e_begin, e_end, use_bvh, n_threads, hits, segs, segc, rf_ref, rf_fix, rf_flg, st)
        finally:
            lib().orc_set_ref64(None)
        out.update(hits=hits, segs=segs, seg_count=segc, rf_ref=rf_ref, rf_ref64=rf_ref64, rf_fix=rf_fix, rf_flags=rf_flg, stats=st.as_dict())
        if want_fix:
            out["rf"] = finalize_rf(rf_fix, rf_flg)
        return out

    def _trace(self, params, el_pos, el_dir, tex, frame_id, e_begin, e_end, use_bvh, n_threads, hits, segs, segc, rf_ref, rf_fix, rf_flg, st):
        lib().orc_trace_frame(C.byref(self.c), C.byref(params), _p(el_pos), _p(el_dir), _p(tex),
                              C.c_uint32(frame_id), C.c_uint32(e_begin), C.c_uint32(e_end), int(use_bvh), int(n_threads),
                              _p(hits), _p(segs), _p(segc), _p(rf_ref), _p(rf_fix), _p(rf_flg), C.cast(C.byref(st), C.c_void_p))


def finalize_rf(rf_fix, rf_flags):
    ne, R = rf_fix.shape
    out = np.zeros((R, ne), np.float32)
    lib().orc_finalize_rf(_p(np.ascontiguousarray(rf_fix)), _p(np.ascontiguousarray(rf_flags)), C.c_uint32(ne), C.c_uint32(R), _p(out))
    return out


def convolve(img, axial, lateral):
    img = np.ascontiguousarray(img, np.float32).copy()
    tmp = np.zeros_like(img)
    ax = np.ascontiguousarray(axial, np.float32); lat = np.ascontiguousarray(lateral, np.float32)
    lib().orc_convolve(_p(img), _p(tmp), C.c_uint32(img.shape[0]), C.c_uint32(img.shape[1]), _p(ax), C.c_uint32(ax.size), _p(lat), C.c_uint32(lat.size))
    return img


def envelope(img):
    img = np.ascontiguousarray(img, np.float32).copy()
    lib().orc_envelope(_p(img), C.c_uint32(img.shape[0]), C.c_uint32(img.shape[1]))
    return img


def scan_convert(img, radius_mm=30.0, total_angle=1.0471975511965976, max_travel_us=100.0, sos=1500.0, out_rows=400, out_cols=500):
    img = np.ascontiguousarray(img, np.float32)
    out = np.zeros((out_rows, out_cols), np.float32)
    lib().orc_scan_convert(_p(img), C.c_uint32(img.shape[0]), C.c_uint32(img.shape[1]), C.c_double(radius_mm), C.c_double(total_angle),
                           C.c_double(max_travel_us), C.c_double(sos), _p(out), C.c_uint32(out_rows), C.c_uint32(out_cols))
    return out


def scan_maps(rows, cols, radius_mm=30.0, total_angle=1.0471975511965976, max_travel_us=100, sos=1500, out_rows=400, out_cols=500):
    """rfimage.h:183-215 create_mapping -> (map_row = the reference's map_x, map_col = its map_y), each [out_rows][out_cols]"""
    mr = np.zeros((out_rows, out_cols), np.float32); mc = np.zeros((out_rows, out_cols), np.float32)
    lib().orc_scan_maps(C.c_uint32(rows), C.c_uint32(cols), C.c_double(radius_mm), C.c_double(total_angle), C.c_uint32(max_travel_us), C.c_uint32(sos),
                        C.c_uint32(out_rows), C.c_uint32(out_cols), _p(mr), _p(mc))
    return mr, mc
