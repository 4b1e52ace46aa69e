"""GPU parity tests (run on the MI355X box): the HIP path, called through the C-ABI, against the CPU
oracle on the same seeded inputs.  Bars: hit indices bit-exact; fixed-point RF image bit-exact against the
oracle's contract accumulation; within 1e-4 (relative to the image peak) of the oracle's float summation in
the reference's order (main.cpp:106-144); PSF convolution bit-exact."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL_REF = 1e-4   # north_star: RF image within 1e-4 relative of the CPU reference path
ATOL_FLOOR = 1e-5  # element-wise criterion: |d| <= RTOL_REF * |ref| + ATOL_FLOOR * peak (the floor covers the float reference sum's own rounding at <= 1024 samples; tests/test_gpu_baseline_configs.py separates it out exactly)


def _sim(mcrt, cfg, sd, E, S, **kw):
    tr = mcrt.Transducer(E, position=cfg["transducerPosition"], angles_deg=cfg["transducerAngles"])
    return tr, mcrt.Simulator(sd, tr, n_samples=S, **kw)


def _oracle(orc, sd, tr, tex, E, S, bvh=None, threads=8, **pk):
    osc = orc.OracleScene(sd.tri, sd.tri_mesh, sd.meshes, sd.materials, sd.start_mat, sd.spacing, bvh=bvh)
    p = orc.default_params(n_elements=E, n_samples=S, **pk)
    return osc, p, osc.trace_frame(p, tr.pos, tr.dir, tex, use_bvh=bvh is not None, n_threads=threads, want_segs=True)


def _assert_rf(rf_gpu, o):
    assert np.array_equal(rf_gpu.view(np.uint32), o["rf"].view(np.uint32)), "fixed-point RF not bit-exact"
    ref = o["rf_ref"]
    assert np.array_equal(np.isnan(rf_gpu), np.isnan(ref))
    m = ~np.isnan(ref)
    peak = np.abs(ref[m]).max()
    d = np.abs(rf_gpu[m] - ref[m])
    assert d.max() <= RTOL_REF * peak
    assert np.all(d <= RTOL_REF * np.abs(ref[m]) + ATOL_FLOOR * peak)


def test_contract_math_on_gpu(mcrt, orc):
    """every transcendental / IEEE op of the contract, GPU vs oracle, bit for bit"""
    ctx = mcrt.Context(0)
    rng = np.random.default_rng(7)
    n = 200000
    pos = np.exp(rng.uniform(-700, 700, n)); sub = rng.uniform(0, 1, n) * 2.0 ** -1040
    with np.errstate(over="ignore"):            # (doubles beyond the float range become inf: wanted, inf is one of the inputs)
        pos_f = pos[:n // 4].astype(np.float32).astype(np.float64)
    cases = {
        0: (np.concatenate([pos, sub, rng.uniform(0, 2, n), [0.0, 1.0, np.inf, -1.0, np.nan]]), None, "orc_log_d"),
        1: (np.concatenate([rng.uniform(-750, 715, n), rng.uniform(-1, 1, n), [0.0, 800.0, -800.0, np.nan]]), None, "orc_exp_d"),
        6: (np.concatenate([rng.uniform(0, 1, n) ** 8, pos_f]), None, "orc_logf"),
        7: (rng.uniform(-110, 90, n), None, "orc_expf"),
    }
    for op, (x, y, name) in cases.items():
        x = x.astype(np.float32).astype(np.float64) if op >= 6 else x
        g = ctx.debug_math(op, x, y)
        ref = np.array([getattr(orc.lib(), name)(v) for v in x.tolist()], np.float64)
        assert np.array_equal(g.view(np.uint64), ref.view(np.uint64)), name
    # sin / cos
    a = np.concatenate([rng.uniform(0, 2 * np.pi, n), rng.uniform(-100, 100, 1000)])
    gs, gc = ctx.debug_math(2, a), ctx.debug_math(3, a)
    sc = np.array([orc.sincos(v) for v in a.tolist()])
    assert np.array_equal(gs.view(np.uint64), sc[:, 0].copy().view(np.uint64))
    assert np.array_equal(gc.view(np.uint64), sc[:, 1].copy().view(np.uint64))
    # IEEE sqrt / division must be correctly rounded on the GPU (host is IEEE)
    x = np.concatenate([pos, sub, rng.uniform(0, 4, n)]); y = np.concatenate([rng.uniform(-3, 3, n), pos, rng.uniform(1e-300, 1e300, n)])
    assert np.array_equal(ctx.debug_math(4, x).view(np.uint64), np.sqrt(x).view(np.uint64))
    assert np.array_equal(ctx.debug_math(5, x, y).view(np.uint64), (x / y).view(np.uint64))
    xf = rng.uniform(0, 1e6, n).astype(np.float32); yf = (rng.uniform(-1e3, 1e3, n)).astype(np.float32)
    yf[yf == 0] = 1
    den = (rng.uniform(0, 1, n) * 1e-38).astype(np.float32)    # float denormals must survive
    assert np.array_equal(ctx.debug_math(9, xf.astype(np.float64)).astype(np.float32).view(np.uint32), np.sqrt(xf).view(np.uint32))
    assert np.array_equal(ctx.debug_math(10, xf.astype(np.float64), yf.astype(np.float64)).astype(np.float32).view(np.uint32), (xf / yf).view(np.uint32))
    assert np.array_equal(ctx.debug_math(10, den.astype(np.float64), np.full(n, 3.0)).astype(np.float32).view(np.uint32), (den / np.float32(3)).view(np.uint32))
    # pow forms
    u = rng.uniform(0, 1, 20000); e = np.full(20000, float(np.float32(1.0 / 1000001)))
    g = ctx.debug_math(11, u, e)
    ref = np.array([orc.lib().orc_pow_d(a_, b_) for a_, b_ in zip(u.tolist(), e.tolist())])
    assert np.array_equal(g.view(np.uint64), ref.view(np.uint64))
    xb = rng.uniform(-2, 2, 20000).astype(np.float32).astype(np.float64); yb = rng.choice([1.0, 2.0, 0.5, 3.0, 0.2, 0.001], 20000)
    g = ctx.debug_math(8, xb, yb)
    ref = np.array([orc.lib().orc_powf(a_, b_) for a_, b_ in zip(xb.tolist(), yb.tolist())], np.float64)
    assert np.array_equal(np.isnan(g), np.isnan(ref)) and np.array_equal(g[~np.isnan(g)], ref[~np.isnan(ref)])
    # fixed-point echo conversion (one fma + an integer subtract in the kernel) == rint(echo * 2^40), incl. ties and sub-unit echoes
    ech = np.concatenate([rng.normal(size=n) * 10.0 ** rng.uniform(-14, 2, n), (rng.uniform(-1, 1, n) * 2.0 ** -40 * rng.integers(1, 1 << 24, n)),
                          [0.0, -0.0, 2.0 ** -41, 3 * 2.0 ** -42, 5 * 2.0 ** -42, -2.0 ** -41, 2.0 ** -17, 1023.99994, -1023.99994, 1e-45, -1e-45]]).astype(np.float32)
    ech = ech[np.abs(ech) < 1024]
    lo = ctx.debug_math(12, ech.astype(np.float64)).astype(np.int64); hi = ctx.debug_math(13, ech.astype(np.float64)).astype(np.int64)
    ref = np.rint(ech.astype(np.float64) * 2.0 ** 40).astype(np.int64)
    assert np.array_equal((hi << 31) | lo, ref)
    # philox
    for ctr, key in [([0, 0, 0, 0], [0, 0]), ([0xffffffff] * 4, [0xffffffff] * 2), ([1, 2, 3, 4], [5, 6])]:
        assert np.array_equal(ctx.debug_philox(ctr, key), orc.philox(ctr, key))
    ctx.close()


def test_c1_sphere_32x64_bruteforce(mcrt, orc, sphere, tex256):
    """BASELINE config 1 (examples/sphere, 32 scan-lines x 64 rays): semantic ground truth = brute force"""
    cfg, sd = sphere
    E, S = 32, 64
    tr, sim = _sim(mcrt, cfg, sd, E, S, texture=tex256)
    hits, segs, cnt = sim.ctx.trace_frame_debug(0, sim.rf_dev, want_segs=True)
    rf = sim.ctx.export_rf(sim.rf_dev, E, sim.R)
    osc, p, o = _oracle(orc, sd, tr, tex256, E, S)
    assert np.array_equal(hits, o["hits"])
    assert np.array_equal(cnt, o["seg_count"])
    for f in o["segs"].dtype.names:
        a, b = np.ascontiguousarray(segs[f]), np.ascontiguousarray(o["segs"][f])
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), "segment field %s" % f
    _assert_rf(rf, o)
    # convolution
    sim.convolve()
    rfc = sim.ctx.export_rf(sim.rf_dev, E, sim.R)
    oc = orc.convolve(o["rf"], sim.psf.axial_kernel, sim.psf.lateral_kernel)
    assert np.array_equal(rfc.view(np.uint32), oc.view(np.uint32))
    # cast_rays alone returns the same segments
    segs2, cnt2, hits2 = sim.ctx.cast_rays(0)
    assert np.array_equal(hits2, hits) and np.array_equal(cnt2, cnt) and segs2.tobytes() == segs.tobytes()
    sim.close()


@pytest.mark.parametrize("builder", ["sah", "lbvh"])
def test_axis_parallel_rays_and_rays_in_box_planes(mcrt, orc, sphere, tex256, builder):
    """the slab rule's corner cases (one fma per plane with a finite reciprocal): rays along the axes (zero and negative-zero
    direction components), rays lying IN a face plane / along an edge of the axis-aligned box, through the face's diagonal (a tie
    between its two triangles), from inside the sphere, pointing away, and with components so small that the reciprocal is clamped
    -- hits, segments and RF against brute force"""
    cfg, sd = sphere
    V = sd.tri.reshape(-1, 3)
    lo, hi = V.min(0), V.max(0)                          # the box's faces are the scene bounds
    tiny, den = np.float32(1e-30), np.float32(1e-45)
    el = [((-13.5, 0, 0), (1, 0, 0)), ((-13.5, 0, 0), (1, -0.0, 0.0)), ((-13.5, 0, 0), (1, 0.0, -0.0)),
          ((-13.5, hi[1], 0), (1, 0, 0)), ((-13.5, hi[1], hi[2]), (1, 0, 0)), ((-13.5, lo[1], 0.5), (1, 0, 0)),
          ((0, 0, 0), (0, 1, 0)), ((0, 0, 0), (0, 0, 1)), ((0, 0, 0), (0, -1, 0)), ((0, 0, 0), (-1, 0, 0)), ((0, 0, 0), (0, 0, -1)),
          ((-13.5, 0, 0), (1, tiny, 0)), ((-13.5, 0, 0), (1, -tiny, den)), ((-13.5, 0, 0), (1, 1e-38, -1e-38)),
          ((-13.5, 2, 0), (1, 0, 0)), ((-13.5, 0, 2), (1, 0, 0)), ((-13.5, 0, 0), (-1, 0, 0)), ((0, hi[1], 0), (0, -1, 0)),
          ((lo[0], lo[1], lo[2]), (1, 1, 1)), ((hi[0], 0, 0), (-1, 0, 0)), ((0, 0, -13.5), (0, 0, 1)), ((0.25, -13.5, 0.25), (0, 1, 0))]
    pos = np.array([e[0] for e in el], np.float32)
    d = np.array([e[1] for e in el], np.float32)
    d[18] /= np.float32(np.sqrt(3.0))
    E, S = len(el), 48
    tr, sim = _sim(mcrt, cfg, sd, E, S, texture=tex256, bvh_builder=builder)
    sim.ctx.set_transducer(pos, d)
    hits, segs, cnt = sim.ctx.trace_frame_debug(2, sim.rf_dev, want_segs=True)
    rf = sim.ctx.export_rf(sim.rf_dev, E, sim.R)
    osc = orc.OracleScene(sd.tri, sd.tri_mesh, sd.meshes, sd.materials, sd.start_mat, sd.spacing)
    p = orc.default_params(n_elements=E, n_samples=S)
    o = osc.trace_frame(p, pos, d, tex256, frame_id=2, use_bvh=False, n_threads=16, want_segs=True)
    assert (o["hits"][:, :, 0] >= 0).sum() >= 12 * S, "most of these rays must hit something"
    assert np.array_equal(hits, o["hits"])
    assert np.array_equal(cnt, o["seg_count"])
    assert segs.tobytes() == o["segs"].tobytes()
    _assert_rf(rf, o)
    # ... and the oracle's walk of the product's tree agrees with its own brute force on them
    nodes4, _ = sim.ctx.get_bvh4(); _, btri, _ = sim.ctx.get_bvh()
    osc.set_bvh4(nodes4, btri)
    o2 = osc.trace_frame(p, pos, d, tex256, frame_id=2, use_bvh=2, n_threads=16, want_ref=False)
    assert np.array_equal(o2["hits"], o["hits"])
    sim.close()


def test_c2_sphere_128x1024_depth512(mcrt, orc, sphere, tex256):
    """BASELINE config 2: 128 scan-lines x 1024 rays, 512 RF rows, GPU BVH vs CPU parity (oracle walks the product's BVH,
    itself validated against brute force in tests/test_host_pieces.py)"""
    cfg, sd = sphere
    E, S = 128, 1024
    tr, sim = _sim(mcrt, cfg, sd, E, S, n_rows=512, texture=tex256)
    hits, _, _ = sim.ctx.trace_frame_debug(3, sim.rf_dev)
    rf = sim.ctx.export_rf(sim.rf_dev, E, sim.R)
    nodes, btri, _ = sim.ctx.get_bvh()
    nodes4, max_stack = sim.ctx.get_bvh4()
    assert max_stack <= 64
    osc = orc.OracleScene(sd.tri, sd.tri_mesh, sd.meshes, sd.materials, sd.start_mat, sd.spacing, bvh=(nodes, btri))
    osc.set_bvh4(nodes4)
    p = orc.default_params(n_elements=E, n_samples=S, n_rows=512)
    o = osc.trace_frame(p, tr.pos, tr.dir, tex256, frame_id=3, use_bvh=2, n_threads=16)
    assert np.array_equal(hits, o["hits"])
    _assert_rf(rf, o)
    # node / triangle visit counts are part of the contract (same walk on both sides)
    # (with statistics enabled the library walks every ray once -- no sub-range pieces -- so the counts are comparable)
    sim.ctx.enable_stats(True); sim.ctx.get_stats(reset=True)
    sim.trace(3); st = sim.ctx.get_stats()
    sim.ctx.enable_stats(False)
    # bounce 0 is walked once per scan-line on the GPU (all S samples start as copies of first_ray, scene.cpp:83-101)
    p0 = orc.default_params(n_elements=E, n_samples=S, n_rows=512, max_depth=1)
    o0 = osc.trace_frame(p0, tr.pos, tr.dir, tex256, frame_id=3, use_bvh=2, n_threads=16, want_ref=False, want_fix=False)["stats"]
    for k in ("queries", "nodes_visited", "tris_tested"):
        assert o0[k] % S == 0
        assert st[k] == o["stats"][k] - o0[k] + o0[k] // S, k
    for k in ("segments", "hits"):
        assert st[k] == o["stats"][k], k
    assert 0 < st["rf_steps"] <= o["stats"]["rf_steps"]      # the GPU skips the steps of silent media (mu0 = sigma = 0): exact no-ops
    sim.close()


def test_element_sharding_is_exact(mcrt, orc, sphere, tex256):
    """scan-line shards (what each GPU of a node traces) concatenate to the full frame bit for bit"""
    cfg, sd = sphere
    E, S = 16, 128
    tr, sim = _sim(mcrt, cfg, sd, E, S, texture=tex256)
    sim.trace(1)
    full = sim.ctx.d2h(sim.rf_dev, (E, sim.R))
    parts = []
    for g in range(4):
        sim.ctx.trace_frame(1, sim.rf_dev, g * 4, g * 4 + 4)
        parts.append(sim.ctx.d2h(sim.rf_dev, (4, sim.R)))
    assert np.array_equal(np.concatenate(parts).view(np.uint32), full.view(np.uint32))
    # and the frame is reproducible run to run
    sim.trace(1)
    assert np.array_equal(sim.ctx.d2h(sim.rf_dev, (E, sim.R)).view(np.uint32), full.view(np.uint32))
    sim.close()


def test_reference_shape_512x5_and_tir_flag(mcrt, orc, tex256):
    """the reference's own launch shape (512 elements x 5 samples, main.cpp:26-27) on the single-mesh 'simple' scene
    (SPHERE = BONE in FAT, simple.scene) where total internal reflection produces NaN echoes (quirk 5)"""
    cfg, meshes = mcrt.synth.sphere_scene(4)
    cfg["meshes"] = [m for m in cfg["meshes"] if m["file"] == "SPHERE.obj"]
    cfg["meshes"][0]["outsideMaterial"] = "FAT"; cfg["startingMaterial"] = "FAT"
    sd = mcrt.scene_io.build_scene(cfg, meshes)
    E, S = 512, 5
    for sanitize in (0, 1):
        tr, sim = _sim(mcrt, cfg, sd, E, S, texture=tex256, sanitize_tir=sanitize)
        hits, _, _ = sim.ctx.trace_frame_debug(0, sim.rf_dev)
        rf = sim.ctx.export_rf(sim.rf_dev, E, sim.R)
        osc, p, o = _oracle(orc, sd, tr, tex256, E, S, sanitize_tir=sanitize)
        assert np.array_equal(hits, o["hits"])
        _assert_rf(rf, o)
        sim.close()


def test_liver_like_scene_vessels_and_thickness(mcrt, orc, tex256):
    """the ircad11-like scene: vascular meshes (BLOOD), BONE thickness 0.3 (Box-Muller draw, scene.cpp:132-139), scaling 0.1,
    rotated probe -- exercises every branch of the material-transition logic (ray.cpp:14-47)"""
    cfg, meshes = mcrt.synth.liver_scene(3)
    sd = mcrt.scene_io.build_scene(cfg, meshes)
    E, S = 48, 256
    tr, sim = _sim(mcrt, cfg, sd, E, S, texture=tex256)
    hits, segs, cnt = sim.ctx.trace_frame_debug(5, sim.rf_dev, want_segs=True)
    rf = sim.ctx.export_rf(sim.rf_dev, E, sim.R)
    nodes, btri, _ = sim.ctx.get_bvh()
    osc = orc.OracleScene(sd.tri, sd.tri_mesh, sd.meshes, sd.materials, sd.start_mat, sd.spacing, bvh=(nodes, btri))
    osc.set_bvh4(sim.ctx.get_bvh4()[0])
    p = orc.default_params(n_elements=E, n_samples=S)
    o = osc.trace_frame(p, tr.pos, tr.dir, tex256, frame_id=5, use_bvh=2, n_threads=16, want_segs=True)
    assert np.array_equal(hits, o["hits"]) and np.array_equal(cnt, o["seg_count"])
    assert segs.tobytes() == o["segs"].tobytes()
    assert (hits >= 0).sum() > E * S            # the scene is actually hit, several bounces deep
    assert len(np.unique(o["segs"]["media"][o["segs"]["initial_intensity"] > 0])) >= 4
    _assert_rf(rf, o)
    sim.close()


def test_random_triangle_soup_100k(mcrt, orc, tex256):
    """BASELINE config 4's geometry at 100k triangles: incoherent secondary rays, deep BVH"""
    cfg, meshes = mcrt.synth.random_scene(100000, 8, seed=99)
    sd = mcrt.scene_io.build_scene(cfg, meshes)
    E, S = 32, 512
    tr, sim = _sim(mcrt, cfg, sd, E, S, texture=tex256)
    hits, _, cnt = sim.ctx.trace_frame_debug(1, sim.rf_dev, want_segs=True)
    rf = sim.ctx.export_rf(sim.rf_dev, E, sim.R)
    nodes, btri, _ = sim.ctx.get_bvh()
    osc = orc.OracleScene(sd.tri, sd.tri_mesh, sd.meshes, sd.materials, sd.start_mat, sd.spacing, bvh=(nodes, btri))
    osc.set_bvh4(sim.ctx.get_bvh4()[0])
    p = orc.default_params(n_elements=E, n_samples=S)
    o = osc.trace_frame(p, tr.pos, tr.dir, tex256, frame_id=1, use_bvh=2, n_threads=16)
    assert np.array_equal(hits, o["hits"])
    _assert_rf(rf, o)
    sim.close()


def test_envelope_and_scan_conversion(mcrt, orc, sphere, tex256):
    """the rest of a B-mode frame: rf_image::envelope (rfimage.h:54-91) and the polar->Cartesian remap
    (rfimage.h:125-140,183-215; exact bilinear, maps computed on the host as the reference's constructor does)"""
    cfg, sd = sphere
    E, S = 128, 64
    tr, sim = _sim(mcrt, cfg, sd, E, S, texture=tex256, sanitize_tir=1)
    sim.trace(0); sim.convolve()
    img = sim.ctx.export_rf(sim.rf_dev, E, sim.R)
    sim.ctx.envelope(sim.rf_dev, E, sim.R)
    env = sim.ctx.export_rf(sim.rf_dev, E, sim.R)
    ref_env = orc.envelope(img)
    assert np.array_equal(env.view(np.uint32), ref_env.view(np.uint32))
    out_dev = sim.ctx.alloc(400 * 500 * 4)
    sim.ctx.scan_convert(sim.rf_dev, E, sim.R, out_dev)
    sc = sim.ctx.d2h(out_dev, (400, 500))
    ref_sc = orc.scan_convert(ref_env)
    assert np.array_equal(sc.view(np.uint32), ref_sc.view(np.uint32))
    assert np.count_nonzero(sc) > 10000
    sim.ctx.free(out_dev)
    sim.close()


def test_user_texture_and_small_shapes(mcrt, orc):
    """caller-supplied texture (non-finite voxels disable the silent-medium shortcut), odd sizes: E=3, S=7, R=100, B=3"""
    cfg, meshes = mcrt.synth.sphere_scene(2)
    sd = mcrt.scene_io.build_scene(cfg, meshes)
    rng = np.random.default_rng(3)
    tex = rng.normal(size=(8, 8, 8, 2)).astype(np.float32)
    tex[1, 2, 3, 0] = np.inf
    E, S = 3, 7
    tr = mcrt.Transducer(E, position=cfg["transducerPosition"], angles_deg=cfg["transducerAngles"])
    sim = mcrt.Simulator(sd, tr, n_samples=S, n_rows=100, texture=tex, tex_n=8, max_depth=3)
    hits, _, _ = sim.ctx.trace_frame_debug(9, sim.rf_dev)
    rf = sim.ctx.export_rf(sim.rf_dev, E, sim.R)
    osc = orc.OracleScene(sd.tri, sd.tri_mesh, sd.meshes, sd.materials, sd.start_mat, sd.spacing)
    p = orc.default_params(n_elements=E, n_samples=S, n_rows=100, max_depth=3, tex_n=8)
    o = osc.trace_frame(p, tr.pos, tr.dir, tex, frame_id=9)
    assert hits.shape == (3, 7, 3) and np.array_equal(hits, o["hits"])
    assert np.array_equal(rf.view(np.uint32), o["rf"].view(np.uint32))
    sim.close()


def test_error_paths(mcrt):
    ctx = mcrt.Context(0)
    with pytest.raises(mcrt.McrtError):
        ctx.set_params(max_depth=99)
    with pytest.raises(mcrt.McrtError):
        ctx.set_params(n_rows=0)
    dev = ctx.alloc(1024)
    with pytest.raises(mcrt.McrtError) as e:
        ctx.trace_frame(0, dev)
    assert "no scene uploaded" in str(e.value)
    ctx.free(dev)
    ctx.close()


def test_pass_size_limits(mcrt, sphere, tex256):
    """mcrt_trace_frames: up to 1024 frames and 2^27 paths per pass, refused beyond with MCRT_ERR_LIMIT (not an allocation failure)"""
    cfg, sd = sphere
    tr, sim = _sim(mcrt, cfg, sd, 16, 64, texture=tex256)
    dev = sim.ctx.alloc(4096)
    with pytest.raises(mcrt.McrtError, match="n_frames must be 1..1024"):
        sim.ctx.trace_frames(0, 1025, dev)
    with pytest.raises(mcrt.McrtError, match="n_frames must be 1..1024"):
        sim.ctx.trace_frames(0, 0, dev)
    sim.ctx.free(dev); sim.close()
    tr, sim = _sim(mcrt, cfg, sd, 512, 16384, texture=tex256)
    dev = sim.ctx.alloc(4096)
    with pytest.raises(mcrt.McrtError, match="more than 2\\^27 paths"):
        sim.ctx.trace_frames(0, 17, dev)            # 17 x 512 x 16384 > 2^27 (checked before anything is allocated or written)
    sim.ctx.free(dev); sim.close()


def test_fast_paths_are_on_for_the_reference_constants(mcrt, orc, sphere, tex256):
    """the RF accumulation's fast paths are switched on by checks made on the device (the reciprocal-multiply voxel quotient must
    equal IEEE division for every float in its gate); for the reference's constants (0.145 mm texels, 256^3, 100 us) they must be ON
    -- from round 1 to round 3 a bitwise comparison of -0 with +0 had kept them off, unnoticed -- and switching them off by the
    knobs changes no bit of the image"""
    cfg, sd = sphere
    E, S = 16, 96
    tr, sim = _sim(mcrt, cfg, sd, E, S, texture=tex256)
    rf = sim.frame(1, convolve=False).copy()
    fast_div, lean, rows = sim.ctx.debug_fast_paths()
    assert fast_div and lean and rows == max(sim.R + 1, 467)      # (100 us x 4.658 rows per us: the largest row guess is 465)
    sim.close()
    for knob in ("MCRT_NO_LEAN", "MCRT_NO_FAST_DIV"):
        os.environ[knob] = "1"
        try:
            tr2, sim2 = _sim(mcrt, cfg, sd, E, S, texture=tex256)
            rf2 = sim2.frame(1, convolve=False).copy()
            f2 = sim2.ctx.debug_fast_paths()
            sim2.close()
        finally:
            del os.environ[knob]
        assert f2[2] == 0 and not f2[1]
        assert np.array_equal(rf2.view(np.uint32), rf.view(np.uint32)), knob


def test_knobs_need_mcrt_tuning(mcrt, sphere, tex256, monkeypatch):
    """the library reads its tuning knobs only in a process started with MCRT_TUNING=1 (tests/conftest.py sets it for this suite): linked
    into someone else's program it looks at that ONE variable and has its defaults"""
    cfg, sd = sphere
    monkeypatch.setenv("MCRT_NO_LEAN", "1")
    monkeypatch.delenv("MCRT_TUNING", raising=False)
    tr, sim = _sim(mcrt, cfg, sd, 16, 64, texture=tex256)
    rf = sim.frame(1, convolve=False).copy()
    assert sim.ctx.debug_fast_paths()[1] and sim.ctx.debug_fast_paths()[2] > 0           # the knob was NOT read
    sim.close()
    monkeypatch.setenv("MCRT_TUNING", "1")
    tr, sim = _sim(mcrt, cfg, sd, 16, 64, texture=tex256)
    rf2 = sim.frame(1, convolve=False).copy()
    assert not sim.ctx.debug_fast_paths()[1] and sim.ctx.debug_fast_paths()[2] == 0       # now it was
    sim.close()
    assert np.array_equal(rf.view(np.uint32), rf2.view(np.uint32))


def test_abandoned_launch_poisons_frames_until_asked(mcrt, sphere, tex256, monkeypatch):
    """a launch abandoned by a kernel watchdog sets the context's device error word: every image finalised from then on is NaN
    throughout, mcrt_synchronize reports MCRT_ERR_LIMIT once and clears it, the next frame is the frame again.  (The hook that sets
    the word by hand is refused unless the context was created with MCRT_TEST_HOOKS in the environment.)"""
    cfg, sd = sphere
    monkeypatch.delenv("MCRT_TEST_HOOKS", raising=False)
    tr, sim = _sim(mcrt, cfg, sd, 16, 64, texture=tex256)
    with pytest.raises(mcrt.McrtError, match="test hook"):
        sim.ctx.debug_set_error(2)
    sim.close()
    monkeypatch.setenv("MCRT_TEST_HOOKS", "1")
    tr, sim = _sim(mcrt, cfg, sd, 16, 64, texture=tex256)
    good = sim.frame(3, convolve=False).copy()
    assert np.isfinite(good).all() and np.abs(good).sum() > 0
    sim.ctx.debug_set_error(2)
    assert np.isnan(sim.frame(3, convolve=False)).all()
    assert np.isnan(sim.frame(4, convolve=False)).all()   # sticky: nobody has asked yet
    with pytest.raises(mcrt.McrtError, match="watchdog"):
        sim.ctx.synchronize()
    sim.ctx.synchronize()                              # reported once
    assert np.array_equal(sim.frame(3, convolve=False).view(np.uint32), good.view(np.uint32))
    sim.close()


def test_cpp_host_cli_matches_oracle(mcrt, orc, tex256, tmp_path):
    """the C++ host mirror (host/mcrt_host.hpp: JSON scene file, OBJ meshes, transducer<512>, psf, rf_image) driven by the
    mattausch_hip CLI with the reference's launch shape (512 x 5) against the oracle: trace + convolve + envelope"""
    import json, os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "mcray-tracing_amd", "mattausch_hip")
    subprocess.check_call(["make", "-C", os.path.join(root, "mcray-tracing_amd"), "mattausch_hip"])      # make's dependency check decides (host/*.cpp, the .so)
    cfg, meshes = mcrt.synth.sphere_scene(3)
    cfg["workingDirectory"] = str(tmp_path) + "/"
    for f, (V, F) in meshes.items():
        mcrt.scene_io.save_obj(str(tmp_path / f), V, F)
    (tmp_path / "sphere.scene").write_text(json.dumps(cfg))
    r = subprocess.run([exe, str(tmp_path / "sphere.scene"), "3", "5", str(tmp_path / "bmode.pgm"), str(tmp_path / "rf.bin")],
                       capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "rays/s" in r.stdout and (tmp_path / "bmode.pgm").stat().st_size == 15 + 400 * 500
    got = np.fromfile(str(tmp_path / "rf.bin"), np.float32).reshape(465, 512)
    sd = mcrt.scene_io.load_scene_file(str(tmp_path / "sphere.scene"))         # OBJ round trip, like the CLI
    tr = mcrt.Transducer(512, position=cfg["transducerPosition"], angles_deg=cfg["transducerAngles"])
    osc = orc.OracleScene(sd.tri, sd.tri_mesh, sd.meshes, sd.materials, sd.start_mat, sd.spacing)
    p = orc.default_params()
    o = osc.trace_frame(p, tr.pos, tr.dir, tex256, frame_id=2, n_threads=8)       # the CLI's last frame id
    ax, lat = orc.psf()
    ref = orc.envelope(orc.convolve(o["rf"], ax, lat))
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    # the same program over a GROUP (mcrt_group_*: here three ranks sharing GPU 0, 171 + 171 + 170 scan-lines): the same image and B-mode file
    r3 = subprocess.run([exe, str(tmp_path / "sphere.scene"), "3", "5", str(tmp_path / "bmode3.pgm"), str(tmp_path / "rf3.bin"), "--devices", "0,0,0"],
                        capture_output=True, text=True, timeout=240)
    assert r3.returncode == 0 and "3 GPU context(s)" in r3.stdout, r3.stdout + r3.stderr
    assert np.array_equal(np.fromfile(str(tmp_path / "rf3.bin"), np.uint32), got.view(np.uint32).ravel())
    assert (tmp_path / "bmode3.pgm").read_bytes() == (tmp_path / "bmode.pgm").read_bytes()
    bad = subprocess.run([exe, str(tmp_path / "missing.scene")], capture_output=True, text=True)
    assert bad.returncode == 1 and "The program found an error" in bad.stdout


def test_frames_in_flight_are_bit_identical(mcrt, orc, sphere, tex256):
    """mcrt_trace_frames: several frame ids traced as one pass give exactly the images of one-at-a-time tracing"""
    cfg, sd = sphere
    E, S, F = 16, 128, 3
    tr, sim = _sim(mcrt, cfg, sd, E, S, texture=tex256)
    dev = sim.ctx.alloc(F * E * sim.R * 4)
    sim.ctx.trace_frames(7, F, dev)
    batch = sim.ctx.d2h(dev, (F, E, sim.R))
    for f in range(F):
        sim.trace(7 + f)
        one = sim.ctx.d2h(sim.rf_dev, (E, sim.R))
        assert np.array_equal(batch[f].view(np.uint32), one.view(np.uint32))
    osc, p, o = _oracle(orc, sd, tr, tex256, E, S)
    o8 = osc.trace_frame(p, tr.pos, tr.dir, tex256, frame_id=8, want_hits=False, want_ref=False)
    assert np.array_equal(batch[1].T.view(np.uint32), o8["rf"].view(np.uint32))
    # mcrt_convolve_frames on the whole pass == mcrt_convolve frame by frame
    sim.ctx.convolve_frames(dev, F, E, sim.R, sim.psf.axial_kernel, sim.psf.lateral_kernel)
    conv = sim.ctx.d2h(dev, (F, E, sim.R))
    for f in range(F):
        sim.trace(7 + f); sim.convolve()
        assert np.array_equal(conv[f].view(np.uint32), sim.ctx.d2h(sim.rf_dev, (E, sim.R)).view(np.uint32))
    sim.ctx.free(dev)
    sim.close()


def test_a_large_pass_is_bit_identical_too(mcrt, sphere, tex256):
    """300 frames in one pass (the pass size bench.py's default sits between: 128) against one-at-a-time tracing of the first, a
    middle and the last frame: the images must not depend on how many frames share the launches"""
    cfg, sd = sphere
    E, S, F = 24, 96, 300
    tr, sim = _sim(mcrt, cfg, sd, E, S, texture=tex256)
    dev = sim.ctx.alloc(F * E * sim.R * 4)
    sim.ctx.trace_frames(1000, F, dev)
    batch = sim.ctx.d2h(dev, (F, E, sim.R))
    for f in (0, 131, F - 1):
        sim.trace(1000 + f)
        assert np.array_equal(batch[f].view(np.uint32), sim.ctx.d2h(sim.rf_dev, (E, sim.R)).view(np.uint32)), f
    assert not np.array_equal(batch[0].view(np.uint32), batch[1].view(np.uint32))       # (frames do differ: the random draws are keyed by frame)
    sim.ctx.free(dev)
    sim.close()


def test_device_lbvh_gives_the_same_frames(mcrt, orc, tex256):
    """the BVH built on the GPU (Morton LBVH -> BVH4) is a different tree, yet hits, segments and the RF image
    are bit-identical to the host SAH tree's and to the oracle's (the closest-hit contract does not depend on the hierarchy);
    the oracle also walks the downloaded device-built tree and must count exactly the GPU's node visits"""
    cfg, meshes = mcrt.synth.random_scene(100000, 8, seed=99)
    sd = mcrt.scene_io.build_scene(cfg, meshes)
    E, S = 32, 256
    tr, sah = _sim(mcrt, cfg, sd, E, S, texture=tex256)
    h0, s0, c0 = sah.ctx.trace_frame_debug(3, sah.rf_dev, want_segs=True)
    rf0 = sah.ctx.export_rf(sah.rf_dev, E, sah.R)
    sah.close()
    tr, sim = _sim(mcrt, cfg, sd, E, S, texture=tex256, bvh_builder="lbvh")
    h1, s1, c1 = sim.ctx.trace_frame_debug(3, sim.rf_dev, want_segs=True)
    rf1 = sim.ctx.export_rf(sim.rf_dev, E, sim.R)
    assert np.array_equal(h0, h1) and np.array_equal(c0, c1) and s0.tobytes() == s1.tobytes()
    assert np.array_equal(rf0.view(np.uint32), rf1.view(np.uint32))
    # the device-built tree, walked by the oracle
    nodes4, max_stack = sim.ctx.get_bvh4()
    _, btri, depth = sim.ctx.get_bvh()
    assert sorted(btri[:, 3].view(np.uint32).tolist()) == list(range(sd.n_tri))          # every triangle exactly once
    assert 1 <= max_stack <= 64 and depth >= 10
    osc = orc.OracleScene(sd.tri, sd.tri_mesh, sd.meshes, sd.materials, sd.start_mat, sd.spacing)
    osc.set_bvh4(nodes4, btri)
    p = orc.default_params(n_elements=E, n_samples=S)
    o = osc.trace_frame(p, tr.pos, tr.dir, tex256, frame_id=3, use_bvh=2, n_threads=16)
    assert np.array_equal(h1, o["hits"])
    _assert_rf(rf1, o)
    sim.ctx.enable_stats(True); sim.ctx.get_stats(reset=True)
    sim.trace(3); st = sim.ctx.get_stats()
    sim.ctx.enable_stats(False)
    p0 = orc.default_params(n_elements=E, n_samples=S, max_depth=1)
    o0 = osc.trace_frame(p0, tr.pos, tr.dir, tex256, frame_id=3, use_bvh=2, n_threads=16, want_ref=False, want_fix=False)["stats"]
    for k in ("queries", "nodes_visited", "tris_tested"):       # bounce 0 is walked once per scan-line on the GPU
        assert st[k] == o["stats"][k] - o0[k] + o0[k] // S, k
    # moved geometry: re-index on the device, compare with a fresh host-SAH context
    moved = sd.tri.reshape(-1, 3, 3).copy()
    moved[:, :, 1] += 0.37 * np.sin(moved[:, :, 0])                # a smooth deformation
    moved = moved.reshape(-1, 9).astype(np.float32)
    sim.ctx.update_triangles(moved)
    h2, _, c2 = sim.ctx.trace_frame_debug(4, sim.rf_dev, want_segs=True)
    rf2 = sim.ctx.export_rf(sim.rf_dev, E, sim.R)
    sim.close()
    import copy
    sd2 = copy.copy(sd); sd2.tri = moved
    tr, ref = _sim(mcrt, cfg, sd2, E, S, texture=tex256)
    h3, _, c3 = ref.ctx.trace_frame_debug(4, ref.rf_dev, want_segs=True)
    rf3 = ref.ctx.export_rf(ref.rf_dev, E, ref.R)
    ref.close()
    assert np.array_equal(h2, h3) and np.array_equal(c2, c3)
    assert np.array_equal(rf2.view(np.uint32), rf3.view(np.uint32))
    assert not np.array_equal(h2, h1)


def test_update_and_builder_error_paths(mcrt, sphere):
    cfg, sd = sphere
    ctx = mcrt.Context(0)
    with pytest.raises(mcrt.McrtError, match="no scene uploaded"):
        ctx.update_triangles(sd.tri)
    with pytest.raises(mcrt.McrtError, match="unknown BVH builder"):
        ctx.set_bvh_builder(7)
    ctx.upload_scene(sd)
    with pytest.raises(mcrt.McrtError, match="triangles, the update has 100"):     # an update keeps the triangle count
        ctx.update_triangles(sd.tri[:100])
    ctx.update_triangles(sd.tri)                                                  # and succeeds with the right one
    ctx.close()


def test_device_lbvh_on_the_liver_scene(mcrt, orc, tex256):
    """the device builder on closed, well-shaped meshes (vessels, thickness draw, scaling 0.1): same frame as the SAH tree"""
    cfg, meshes = mcrt.synth.liver_scene(3)
    sd = mcrt.scene_io.build_scene(cfg, meshes)
    E, S = 32, 128
    out = []
    for builder in ("sah", "lbvh"):
        tr, sim = _sim(mcrt, cfg, sd, E, S, texture=tex256, bvh_builder=builder)
        hits, segs, cnt = sim.ctx.trace_frame_debug(9, sim.rf_dev, want_segs=True)
        out.append((hits, segs.tobytes(), cnt, sim.ctx.export_rf(sim.rf_dev, E, sim.R)))
        sim.close()
    assert np.array_equal(out[0][0], out[1][0]) and out[0][1] == out[1][1] and np.array_equal(out[0][2], out[1][2])
    assert np.array_equal(out[0][3].view(np.uint32), out[1][3].view(np.uint32))
    assert (out[0][0] >= 0).sum() > E * S


def test_large_passes_use_the_per_xcd_queues(mcrt, orc, sphere, tex256):
    """bounces with >= 262144 work items are walked through eight per-XCD sub-queues (k_trace) and queued scan-line-major:
    a 6-frame pass of 64 x 1024 paths (393216 in the first bounces) must equal the same frames traced one at a time (single queue),
    and one of them is checked against the oracle"""
    cfg, sd = sphere
    E, S, F = 64, 1024, 6
    tr, sim = _sim(mcrt, cfg, sd, E, S, texture=tex256)
    dev = sim.ctx.alloc(F * E * sim.R * 4)
    sim.ctx.trace_frames(20, F, dev)
    batch = sim.ctx.d2h(dev, (F, E, sim.R))
    for f in range(F):
        sim.trace(20 + f)
        one = sim.ctx.d2h(sim.rf_dev, (E, sim.R))
        assert np.array_equal(batch[f].view(np.uint32), one.view(np.uint32)), f
    osc, p, o = _oracle(orc, sd, tr, tex256, E, S, threads=32)
    o22 = osc.trace_frame(p, tr.pos, tr.dir, tex256, frame_id=22, n_threads=32, want_hits=False, want_ref=False)
    assert np.array_equal(batch[2].T.view(np.uint32), o22["rf"].view(np.uint32))
    sim.ctx.free(dev)
    sim.close()


@pytest.mark.parametrize("shape", [(48, 40), (128, 1024)])
def test_duplicated_triangles_every_hit_is_a_tie(mcrt, orc, tex256, shape):
    """the closest-hit rule's tie-break (same fraction -> smaller triangle id) when EVERY hit is a tie: the scene's triangles are
    uploaded twice, so each hit has a twin with id + T at exactly the same fraction.  One frame at a time the walk runs in its
    end-of-launch mode from the start (rays cut into pieces, idle lanes taking over subtrees of walking lanes, and of lanes that
    themselves walk a taken-over subtree): whichever lane finds whichever twin first, the answer must be the smaller id -- hit
    indices bit for bit against the oracle (brute force on the small shape, BVH on the large one).  (The twins mostly share a
    leaf; the tie that crosses two hand-overs -- a lane that took over a subtree handing part of it on before its own first find
    -- is the vertex hit of scan-line 327 in test_reference_shape_512x5_and_tir_flag, which caught exactly that in round 2.)"""
    E, S = shape
    cfg, meshes = mcrt.synth.sphere_scene(3)
    sd0 = mcrt.scene_io.build_scene(cfg, meshes)
    T = sd0.n_tri
    sd = mcrt.scene_io.SceneData(np.concatenate([sd0.tri, sd0.tri]), np.concatenate([sd0.tri_mesh, sd0.tri_mesh]), sd0.meshes, sd0.materials,
                                 sd0.material_names, sd0.start_mat, sd0.spacing, sd0.config)
    for builder in ("sah", "lbvh"):
        tr = mcrt.Transducer(E, position=cfg["transducerPosition"], angles_deg=cfg["transducerAngles"])
        sim = mcrt.Simulator(sd, tr, n_samples=S, texture=tex256, bvh_builder=builder)
        hits, _, _ = sim.ctx.trace_frame_debug(5, sim.rf_dev)
        rf = sim.ctx.export_rf(sim.rf_dev, E, sim.R)
        osc = orc.OracleScene(sd.tri, sd.tri_mesh, sd.meshes, sd.materials, sd.start_mat, sd.spacing)
        use_bvh = 0
        if E * S > 4096:
            nodes4, _ = sim.ctx.get_bvh4(); _, btri, _ = sim.ctx.get_bvh()
            osc.set_bvh4(nodes4, btri); use_bvh = 2
        p = orc.default_params(n_elements=E, n_samples=S)
        o = osc.trace_frame(p, tr.pos, tr.dir, tex256, frame_id=5, use_bvh=use_bvh, n_threads=16, want_ref=False)
        assert (o["hits"] >= 0).sum() > E * S // 2
        assert o["hits"].max() < T, "the oracle itself must pick the smaller twin"
        assert np.array_equal(hits, o["hits"]), builder
        assert np.array_equal(rf.view(np.uint32), o["rf"].view(np.uint32)), builder
        sim.close()


@pytest.mark.parametrize("case", range(int(os.environ.get("MCRT_FUZZ_CASES", "16"))))      # (a longer one-off sweep: MCRT_FUZZ_CASES=300)
def test_randomised_configurations(mcrt, orc, case, monkeypatch, tex256):
    """a sweep over shapes and parameters nobody picked by hand: odd element / sample / row counts, depths 1..16, both builders,
    textures of several sizes, TIR sanitising on and off, frequency, seed, frames in flight -- hits and RF bit for bit"""
    rng = np.random.default_rng(1000 + case)
    # every fourth case without cutting the rays of small bounces into pieces, one with two scan-line groups on two streams (the library reads its knobs at mcrt_create)
    for k in ("MCRT_KSPLIT_LIMIT", "MCRT_GROUPS", "MCRT_PACKET_BOUNCES", "MCRT_PACKET_FROM", "MCRT_PATH_MAX"):
        monkeypatch.delenv(k, raising=False)
    if case % 2 == 1 or case == 6:
        # these small passes would all take the LATENCY form (k_path: one launch for every bounce); every other case keeps the staged pipeline
        # (walk / shade / accumulate per bounce, queues, compaction), which is what the knobs below act on
        monkeypatch.setenv("MCRT_PATH_MAX", "0")
    if case % 4 == 3:
        monkeypatch.setenv("MCRT_KSPLIT_LIMIT", "0")
    if case == 6:
        monkeypatch.setenv("MCRT_GROUPS", "2")
    if case % 2 == 1:
        # every other case walks some of its bounces a WAVEFRONT PER RAY PACKET (k_trace_packet: by default only bounce 1 of passes of >= 262144 paths):
        # a drawn set of bounces at ANY pass size -- partial packets (rays not a multiple of 64), packets of mixed directions, depth-16 paths
        monkeypatch.setenv("MCRT_PACKET_BOUNCES", hex(int(rng.integers(1, 1 << 16)) | 2)); monkeypatch.setenv("MCRT_PACKET_FROM", "0")
    if case % 3 == 0:
        cfg, meshes = mcrt.synth.random_scene(int(rng.integers(2000, 30000)), 8, seed=int(rng.integers(1, 1000)))
    elif case % 3 == 1:
        cfg, meshes = mcrt.synth.liver_scene(int(rng.integers(1, 3)))
    else:
        cfg, meshes = mcrt.synth.sphere_scene(int(rng.integers(1, 4)))
    sd = mcrt.scene_io.build_scene(cfg, meshes)
    E, S = int(rng.integers(1, 40)), int(rng.integers(1, 200))
    R, B = int(rng.integers(20, 700)), int(rng.integers(1, 17))
    tex_n = int(rng.choice([4, 16, 37, 64, 256]))           # (256: the reference's own texture -- k_march's fast variant, with odd row counts)
    freq = float(rng.choice([2.5, 4.5, 7.0]))
    sanitize = int(rng.integers(0, 2))
    seed = int(rng.integers(0, 2 ** 31))
    F = int(rng.integers(1, 5))
    builder = "lbvh" if (case % 2 and sd.n_tri >= 8) else "sah"
    tex = tex256 if tex_n == 256 else rng.normal(size=(tex_n, tex_n, tex_n, 2)).astype(np.float32)
    tr = mcrt.Transducer(E, frequency=freq, position=cfg["transducerPosition"], angles_deg=cfg["transducerAngles"])
    sim = mcrt.Simulator(sd, tr, n_samples=S, n_rows=R, texture=tex, tex_n=tex_n, max_depth=B, sanitize_tir=sanitize, seed=seed, bvh_builder=builder)
    frame0 = int(rng.integers(0, 1000))
    dev = sim.ctx.alloc(F * E * R * 4)
    sim.ctx.trace_frames(frame0, F, dev)
    batch = sim.ctx.d2h(dev, (F, E, R))
    hits, _, _ = sim.ctx.trace_frame_debug(frame0 + F - 1, sim.rf_dev)
    osc = orc.OracleScene(sd.tri, sd.tri_mesh, sd.meshes, sd.materials, sd.start_mat, sd.spacing)
    nodes4, _ = sim.ctx.get_bvh4()
    _, btri, _ = sim.ctx.get_bvh()
    osc.set_bvh4(nodes4, btri)
    p = orc.default_params(n_elements=E, n_samples=S, n_rows=R, max_depth=B, tex_n=tex_n, frequency=freq, sanitize_tir=sanitize, seed=seed)
    o = osc.trace_frame(p, tr.pos, tr.dir, tex, frame_id=frame0 + F - 1, use_bvh=2, n_threads=16, want_ref=False)
    what = dict(case=case, E=E, S=S, R=R, B=B, tex_n=tex_n, freq=freq, sanitize=sanitize, F=F, builder=builder, tris=sd.n_tri)
    assert np.array_equal(hits, o["hits"]), what
    got, want = batch[F - 1].T.view(np.uint32), o["rf"].view(np.uint32)
    assert np.array_equal(got, want), what
    sim.ctx.free(dev)
    sim.close()


@pytest.mark.parametrize("builder", ["sah", "lbvh"])
def test_refit_keeps_frames_exact(mcrt, orc, tex256, builder):
    """mcrt_refit_triangles: moved vertices, same tree, boxes refitted on the GPU -- the frame equals that of a context built on
    the moved geometry, and the oracle walking the downloaded refitted tree counts the GPU's node visits"""
    cfg, meshes = mcrt.synth.random_scene(50000, 8, seed=7)
    sd = mcrt.scene_io.build_scene(cfg, meshes)
    E, S = 24, 192
    tr, sim = _sim(mcrt, cfg, sd, E, S, texture=tex256, bvh_builder=builder)
    sim.trace(0)
    moved = sd.tri.reshape(-1, 3, 3).copy()
    moved[:, :, 2] += 0.25 * np.cos(0.7 * moved[:, :, 0]) + 0.1 * moved[:, :, 1]        # smooth deformation + shear
    moved = moved.reshape(-1, 9).astype(np.float32)
    sim.ctx.refit_triangles(moved)
    h1, _, c1 = sim.ctx.trace_frame_debug(6, sim.rf_dev, want_segs=True)
    rf1 = sim.ctx.export_rf(sim.rf_dev, E, sim.R)
    nodes4, max_stack = sim.ctx.get_bvh4()
    _, btri, _ = sim.ctx.get_bvh()
    assert np.array_equal(btri[np.argsort(btri[:, 3].view(np.uint32))][:, [0, 1, 2, 4, 5, 6, 8, 9, 10]], moved)   # the records carry the new vertices
    sim.ctx.enable_stats(True); sim.ctx.get_stats(reset=True)
    sim.trace(6); st = sim.ctx.get_stats()
    sim.ctx.enable_stats(False)
    sim.close()
    import copy
    sd2 = copy.copy(sd); sd2.tri = moved
    tr, ref = _sim(mcrt, cfg, sd2, E, S, texture=tex256)
    h2, _, c2 = ref.ctx.trace_frame_debug(6, ref.rf_dev, want_segs=True)
    rf2 = ref.ctx.export_rf(ref.rf_dev, E, ref.R)
    ref.close()
    assert np.array_equal(h1, h2) and np.array_equal(c1, c2)
    assert np.array_equal(rf1.view(np.uint32), rf2.view(np.uint32))
    osc = orc.OracleScene(moved, sd.tri_mesh, sd.meshes, sd.materials, sd.start_mat, sd.spacing)
    osc.set_bvh4(nodes4, btri)
    p = orc.default_params(n_elements=E, n_samples=S)
    o = osc.trace_frame(p, tr.pos, tr.dir, tex256, frame_id=6, use_bvh=2, n_threads=16, want_ref=False)
    p0 = orc.default_params(n_elements=E, n_samples=S, max_depth=1)
    o0 = osc.trace_frame(p0, tr.pos, tr.dir, tex256, frame_id=6, use_bvh=2, n_threads=16, want_ref=False, want_fix=False)["stats"]
    assert np.array_equal(h1, o["hits"])
    for k in ("queries", "nodes_visited", "tris_tested"):
        assert st[k] == o["stats"][k] - o0[k] + o0[k] // S, k


def _run_bench(extra, nproc=1, timeout=600):
    """bench.py in fresh child processes (torchrun for nproc > 1); a run that outlives `timeout` is killed with its whole
    process group and reported as (None, None)"""
    import json, os, signal, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable]
    if nproc > 1:
        port = 29600 + os.getpid() % 300
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1", "--master-port", str(port)]
    cmd += [os.path.join(root, "bench.py"), "--gpus", str(nproc)] + extra
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=root, start_new_session=True)
    try:
        out, err = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)
        p.communicate()
        return None, None
    r = subprocess.CompletedProcess(cmd, p.returncode, out, err)
    lines = [l for l in out.splitlines() if l.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def _why(stderr_tail):
    """the most telling line of a failed child's stderr"""
    lines = [l.strip() for l in (stderr_tail or "").splitlines() if l.strip() and set(l.strip()) - set("=-")]
    hits = [l for l in lines if ("NCCL" in l or "nccl" in l or "Duplicate GPU" in l or "invalid usage" in l) and "traceback" not in l.lower()] or [l for l in lines if "rror" in l and "traceback" not in l.lower()]
    return (hits[-1] if hits else (lines[-1] if lines else "timed out"))[:200]


def test_two_ranks_gather_equals_single_process():
    """bench.py's N > 1 path (scan-line shards, ONE all-gather per pass on the post stream, PSF on the gathered frames,
    double-buffered against the next pass's trace) with two ranks in fresh child processes on this box's one GPU: the gathered
    frames equal the frames one process traces alone, bit for bit.  RCCL is tried first; where it refuses two ranks on one
    device the same code path runs over gloo (host-staged collective)."""
    small = ["--workload", "sphere", "--scanlines", "16", "--rays", "128", "--steps", "6", "--warmup", "2", "--frames-in-flight", "3",
             "--no-cpu-baseline", "--no-latency-leg", "--no-pmc", "--same-gpu", "--check-gather", "--min-time", "0.05"]
    tried = []
    for backend, limit in (("nccl", 90), ("gloo", 300)):          # (two RCCL ranks on ONE device may be refused, or never finish their rendezvous)
        r, out = _run_bench(small + ["--backend", backend], nproc=2, timeout=limit)
        tried.append((backend, None if r is None else r.returncode, "" if r is None else (r.stderr or "")[-6000:]))
        if r is not None and r.returncode == 0 and out is not None:
            break
    assert r is not None and out is not None and r.returncode == 0, [(t[0], t[1], t[2][-400:]) for t in tried]
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["scan_lines_total"] == 32
    g = out["gather_check"]
    assert g["equal"] and g["ranks"] == 2 and g["nonzero"] > 1000, g
    # WHICH backend carried the green run is part of the record (pytest -q prints the warnings summary): "nccl" = RCCL saw two ranks
    import warnings
    assert g["backend"] in ("nccl", "gloo") and g["backend"] == tried[-1][0]
    warnings.warn("two-rank gather on this box ran over backend=%s%s" % (g["backend"], " (RCCL)" if g["backend"] == "nccl" else
                  " (host-staged; RCCL did not carry two ranks on one device: %s)" % _why(tried[0][2])))
    # strong scaling: a fixed 24-scan-line frame over the two ranks
    r, out = _run_bench([a for a in small if a not in ("--scanlines", "16")] + ["--scanlines-total", "24", "--backend", g["backend"]], nproc=2, timeout=300)
    assert r is not None and r.returncode == 0 and out is not None, "" if r is None else (r.stderr or "")[-800:]
    assert out["scaling"] == "strong" and out["config"]["scan_lines_total"] == 24 and out["gather_check"]["equal"]
    assert out["ranks_seen"] == 2 and [p["rank"] for p in out["per_rank"]] == [0, 1] and all(p["trace_ms"] > 0 for p in out["per_rank"])
    assert out["per_rank"][0]["scan_lines"] == [0, 12] and out["per_rank"][1]["scan_lines"] == [12, 24]
    # BASELINE C4's shape as it is sharded over GPUs: the 1 M-triangle scene, 256 scan-lines x 8192 rays over the two ranks; the
    # gathered whole B-mode frames (PSF, envelope, scan conversion after the gather) equal one process's, and the line explains itself
    r, out = _run_bench(["--workload", "random1m", "--scanlines-total", "256", "--rays", "8192", "--steps", "2", "--warmup", "2", "--frames-in-flight", "2",
                         "--no-cpu-baseline", "--no-latency-leg", "--no-pmc", "--same-gpu", "--check-gather", "--min-time", "0.05", "--backend", g["backend"]], nproc=2, timeout=500)
    assert r is not None and r.returncode == 0 and out is not None, "" if r is None else (r.stderr or "")[-800:]
    assert out["scaling"] == "strong" and out["config"]["scan_lines_total"] == 256 and out["config"]["rays_per_scan_line"] == 8192
    assert out["gather_check"]["equal"] and out["gather_check"]["nonzero_bmode"] > 10000 and out["ranks_seen"] == 2
    assert out["roofline"]["frac"] is not None and out["roofline"]["derived"] is True and 0.0 < out["roofline"]["frac"] < 1.0


def test_eight_ranks_weak_and_ragged_strong():
    """What the driver's SCALE step runs at 2, 4 and 8 GPUs, at its widest, on this box's one GPU: bench.py with EIGHT ranks in fresh child
    processes (gloo carries the collectives: RCCL refuses eight ranks on one device), weak scaling (8 x 16 scan-lines) and strong scaling over a
    ragged split (100 scan-lines = four ranks of 13 and four of 12).  ranks_seen == 8, the gathered whole B-mode frames equal one process's bit
    for bit, and the N > 1 line carries the oracle's word on rank 0's own shard (`parity_check`), not only `gather_check`."""
    small = ["--workload", "sphere", "--rays", "64", "--steps", "4", "--warmup", "2", "--frames-in-flight", "2",
             "--no-latency-leg", "--no-pmc", "--same-gpu", "--check-gather", "--min-time", "0.05", "--backend", "gloo"]
    r, out = _run_bench(small + ["--scanlines", "16"], nproc=8, timeout=900)
    assert r is not None and r.returncode == 0 and out is not None, "" if r is None else _why((r.stderr or "")[-6000:])
    assert out["n_gpus"] == 8 and out["ranks_seen"] == 8 and out["scaling"] == "weak" and out["config"]["scan_lines_total"] == 128
    assert out["gather_check"]["equal"] and out["gather_check"]["ranks"] == 8 and out["gather_check"]["nonzero_bmode"] > 1000
    assert [p["rank"] for p in out["per_rank"]] == list(range(8)) and [p["scan_lines"] for p in out["per_rank"]] == [[16 * k, 16 * k + 16] for k in range(8)]
    assert out["parity_check"]["rf_bit_exact"] is True and out["parity_check"]["scan_lines"] == [0, 16] and out["parity_check"]["rank"] == 0
    assert "cpu_baseline" not in out                                  # (the timed CPU leg stays an N = 1 matter)
    r, out = _run_bench(small + ["--scanlines-total", "100"], nproc=8, timeout=900)
    assert r is not None and r.returncode == 0 and out is not None, "" if r is None else _why((r.stderr or "")[-6000:])
    assert out["ranks_seen"] == 8 and out["scaling"] == "strong" and out["config"]["scan_lines_total"] == 100 and out["gather_check"]["equal"]
    assert [p["scan_lines"][1] - p["scan_lines"][0] for p in out["per_rank"]] == [13, 13, 13, 13, 12, 12, 12, 12]
    assert out["parity_check"]["rf_bit_exact"] is True and out["parity_check"]["scan_lines"] == [0, 13]


def test_rccl_carries_the_collectives_of_the_pass():
    """What a one-GPU box can show of the RCCL leg: a process group over the `nccl` backend (= RCCL on ROCm) with its one rank on this
    GPU runs every collective bench.py's N > 1 path issues -- the gather-to-root probe, the gather of a pass's [F][E/N][R] block on a side
    stream, the all-gather form, the MAX all-reduce of the timed region, the object all-gather of the per-rank records, the barrier --
    and hands the block back unchanged.  (Two ranks on one device are RCCL's to refuse: test_two_ranks_gather_equals_single_process.)"""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    child = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from mcray_tracing_amd import dist as md
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
g = torch.Generator(device="cuda").manual_seed(5)
blk = torch.rand((3, 16, 465), device="cuda", generator=g)
assert md._gather_to_root_ok(dist, None, blk) is True                       # RCCL gathers to a root: no all-gather stand-in
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    out = [torch.empty_like(blk)]
    dist.gather(blk, out, dst=0)
    flat = torch.empty_like(blk)
    dist.all_gather_into_tensor(flat, blk)
s.synchronize()
assert torch.equal(out[0], blk) and torch.equal(flat, blk)
t = torch.tensor([1.25], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t.item()) == 1.25
rec = [None]
dist.all_gather_object(rec, {"rank": 0, "trace_ms": 0.5})
assert rec[0]["trace_ms"] == 0.5
assert md.gather_rf(blk, 16, 465, dist, root=0) is blk                      # one rank: the block IS the frame
dist.barrier()
dist.destroy_process_group()
print("rccl ok")
""" % root
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29950 + os.getpid() % 40),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", child], env=env, cwd=root, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0 and "rccl ok" in r.stdout, (r.stdout[-400:], _why(r.stderr[-6000:]))


def test_bench_line_contract_and_inline_parity():
    """a small single-GPU bench run: the JSON contract keys, the inline parity check against the oracle, a VALU roofline with
    frac <= 1"""
    r, out = _run_bench(["--workload", "sphere", "--scanlines", "16", "--rays", "256", "--steps", "8", "--warmup", "4", "--frames-in-flight", "4",
                         "--no-pmc", "--min-time", "0.05", "--check-gather"], timeout=300)
    assert r is not None and r.returncode == 0 and out is not None, "" if r is None else (r.stderr or "")[-800:]
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity_check"):
        assert k in out, k
    assert out["steps"] == 8 and out["warmup"] == 4 and out["config"]["passes_per_timed_region"] == [4, 4]
    assert out["parity_check"]["rf_bit_exact"] is True and out["parity_check"]["scan_lines"] == 16
    assert out["gather_check"]["equal"]
    assert out["roofline"]["bound"] == "valu" and out["cpu_baseline"]["kind"] == "port" and out["cpu_baseline"]["seconds"] >= 5.0


def test_host_shim_surface(mcrt, orc, tex256, tmp_path):
    """tests/host/host_surface_test.cpp calls every public method of the C++ host shim once and checks the result (exit code =
    failed checks); the RF image it deposits echo by echo on the host (rf_image::add_echo over the segments of scene::cast_rays<5,512>)
    must agree with the frame the GPU accumulates and with the oracle's reference-order image"""
    import json, os, subprocess
    from test_abi import _build_host_surface_test
    exe = _build_host_surface_test()
    cfg, meshes = mcrt.synth.sphere_scene(3)
    cfg["workingDirectory"] = str(tmp_path) + "/"
    for f, (V, F) in meshes.items():
        mcrt.scene_io.save_obj(str(tmp_path / f), V, F)
    (tmp_path / "sphere.scene").write_text(json.dumps(cfg))
    r = subprocess.run([exe, str(tmp_path / "sphere.scene"), str(tmp_path / "host.bin"), str(tmp_path / "fused.bin")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "host surface: 0 check(s) failed" in r.stdout, r.stdout + r.stderr
    assert "rf_image: 465, 512" in r.stdout
    host = np.fromfile(str(tmp_path / "host.bin"), np.float32).reshape(465, 512)
    fused = np.fromfile(str(tmp_path / "fused.bin"), np.float32).reshape(465, 512)
    sd = mcrt.scene_io.load_scene_file(str(tmp_path / "sphere.scene"))
    tr = mcrt.Transducer(512, position=cfg["transducerPosition"], angles_deg=cfg["transducerAngles"])
    osc = orc.OracleScene(sd.tri, sd.tri_mesh, sd.meshes, sd.materials, sd.start_mat, sd.spacing)
    o = osc.trace_frame(orc.default_params(), tr.pos, tr.dir, tex256, frame_id=0, n_threads=8)
    ax, lat = orc.psf()
    assert np.array_equal(fused.view(np.uint32), orc.convolve(o["rf"], ax, lat).view(np.uint32))       # the fused path: bit-exact as ever
    ref = orc.convolve(o["rf_ref"], ax, lat)                                                          # the reference's summation order
    m = ~np.isnan(ref)
    assert np.array_equal(np.isnan(host), np.isnan(ref)) and m.sum() > 0.9 * m.size
    peak = np.abs(ref[m]).max()
    assert peak > 0 and np.abs(host[m] - ref[m]).max() <= 1e-5 * peak          # (std::exp in the host loop vs the contract's expf: <= 1 ulp per step)
    assert np.abs(host[m] - fused[m]).max() <= RTOL_REF * peak


def test_schedules_do_not_change_anything(mcrt, orc, tex256, monkeypatch):
    """how a pass is scheduled is free (counter-keyed RNG, integer RF bins): the default, scan-line groups on their own streams, rays
    of small bounces cut into pieces or not, every bounce walked a wavefront per ray packet (k_trace_packet) or none, the accumulation (and the walk)
    confined to their own CUs, everything on one stream, the walk in its five-wavefronts-per-SIMD form (k_trace_lane_wide, which large launches take by themselves) -- all give bit-identical hits,
    segments, RF images and visit counts, equal to the oracle's."""
    cfg, meshes = mcrt.synth.random_scene(60000, 8, seed=5)
    sd = mcrt.scene_io.build_scene(cfg, meshes)
    E, S, frame = 24, 160, 11
    got = {}
    # ("default" at this size is the LATENCY form -- k_path, every bounce in one launch; the other variants switch it off: they are schedules of the staged pipeline)
    staged = {"MCRT_PATH_MAX": "0"}
    variants = (("default", {}), ("staged", {}), ("packets_every_bounce", {"MCRT_PACKET_BOUNCES": "0xfffe", "MCRT_PACKET_FROM": "0"}), ("no_packets", {"MCRT_PACKET_BOUNCES": "0"}),
                ("two_groups", {"MCRT_GROUPS": "2"}), ("three_groups_no_split", {"MCRT_GROUPS": "3", "MCRT_KSPLIT_LIMIT": "0"}),
                ("masked", {"MCRT_MARCH_CUS": "64", "MCRT_MAIN_MASK": "1"}), ("march_masked", {"MCRT_MARCH_CUS": "96"}), ("no_overlap", {"MCRT_NO_OVERLAP": "1"}),
                ("wide_walk", {"MCRT_WIDE_FROM": "1"}), ("wide_walk_two_groups_no_split", {"MCRT_WIDE_FROM": "1", "MCRT_GROUPS": "2", "MCRT_KSPLIT_LIMIT": "0"}),
                ("narrow_walk", {"MCRT_WIDE_FROM": "4294967295"}), ("narrow_walk_two_groups", {"MCRT_WIDE_FROM": "4294967295", "MCRT_GROUPS": "2"}),      # (round 6: the five-wavefront form is the default from the first ray)
                ("latency_form_masked", {"MCRT_MARCH_CUS": "64", "MCRT_MAIN_MASK": "1", "MCRT_PATH_MAX": "1000000"}))
    for name, env in variants:
        for k in ("MCRT_KSPLIT_LIMIT", "MCRT_GROUPS", "MCRT_MARCH_CUS", "MCRT_MAIN_MASK", "MCRT_NO_OVERLAP", "MCRT_WIDE_FROM", "MCRT_PACKET_BOUNCES", "MCRT_PACKET_FROM", "MCRT_PATH_MAX"):
            monkeypatch.delenv(k, raising=False)
        for k, v in (env if name == "default" else dict(staged, **env)).items():
            monkeypatch.setenv(k, v)                               # (the library reads its knobs once, at mcrt_create)
        tr, sim = _sim(mcrt, cfg, sd, E, S, texture=tex256)
        hits, segs, cnt = sim.ctx.trace_frame_debug(frame, sim.rf_dev, want_segs=True)
        rf = sim.ctx.export_rf(sim.rf_dev, E, sim.R)
        sim.ctx.enable_stats(True); sim.ctx.get_stats(reset=True)
        sim.trace(frame); st = sim.ctx.get_stats()
        sim.ctx.enable_stats(False)
        dev = sim.ctx.alloc(3 * E * sim.R * 4)
        sim.ctx.trace_frames(frame - 1, 3, dev)                   # a 3-frame pass through the same pipeline: its middle frame is `frame`
        batch = sim.ctx.d2h(dev, (3, E, sim.R))
        sim.ctx.free(dev)
        nodes, btri, _ = sim.ctx.get_bvh()
        nodes4 = sim.ctx.get_bvh4()[0]
        sim.close()
        got[name] = (hits, segs.tobytes(), cnt, rf, st, batch[1].T.copy(), nodes4)
    for k in ("MCRT_KSPLIT_LIMIT", "MCRT_GROUPS", "MCRT_MARCH_CUS", "MCRT_MAIN_MASK", "MCRT_NO_OVERLAP", "MCRT_WIDE_FROM", "MCRT_PACKET_BOUNCES", "MCRT_PACKET_FROM", "MCRT_PATH_MAX"):
        monkeypatch.delenv(k, raising=False)
    a = got["default"]
    for name in [v[0] for v in variants[1:]]:
        b = got[name]
        assert np.array_equal(a[0], b[0]) and a[1] == b[1] and np.array_equal(a[2], b[2]), name
        assert np.array_equal(a[3].view(np.uint32), b[3].view(np.uint32)), name
        assert {k: v for k, v in a[4].items() if k != "rf_steps"} == {k: v for k, v in b[4].items() if k != "rf_steps"}, name
    for name, g in got.items():
        assert np.array_equal(g[5].view(np.uint32), g[3].view(np.uint32)), name
    p = orc.default_params(n_elements=E, n_samples=S)
    p0 = orc.default_params(n_elements=E, n_samples=S, max_depth=1)
    osc = orc.OracleScene(sd.tri, sd.tri_mesh, sd.meshes, sd.materials, sd.start_mat, sd.spacing, bvh=(nodes, btri))
    osc.set_bvh4(a[6])
    o = osc.trace_frame(p, tr.pos, tr.dir, tex256, frame_id=frame, use_bvh=2, n_threads=16)
    assert np.array_equal(a[0], o["hits"])
    _assert_rf(a[3], o)
    o0 = osc.trace_frame(p0, tr.pos, tr.dir, tex256, frame_id=frame, use_bvh=2, n_threads=16, want_ref=False, want_fix=False)["stats"]
    for k in ("queries", "nodes_visited", "tris_tested"):
        assert a[4][k] == o["stats"][k] - o0[k] + o0[k] // S, k


def test_deep_tree_overflow_stacks_of_two_groups(mcrt, orc, tex256, monkeypatch):
    """a tree whose worst-case traversal stack exceeds the walk's 32 LDS entries, traced as TWO scan-line groups whose walks
    run side by side: each group owns its overflow array, so hits stay the oracle's; and once more with the walk in its five-wavefront
    form (28 LDS entries, 1280 workgroups: another split of the stack, another stride of the overflow array)"""
    rng = np.random.default_rng(77)
    # triangles at ten nested scales around a point on the probe's line of sight: the SAH tree splits the scales off one by one and
    # gets deep (worst-case stack 51 entries), and the rays through the centre cross every scale
    n, levels = 20000, 10
    k = rng.integers(0, levels, size=n)
    scale = (0.5 ** k).astype(np.float32)
    c = (rng.uniform(-1, 1, size=(n, 3)).astype(np.float32) * scale[:, None] * np.array([8, 6, 6], np.float32)) + np.array([-2, 0, 0], np.float32)
    e1 = rng.uniform(-0.3, 0.3, size=(n, 3)).astype(np.float32) * scale[:, None]
    e2 = rng.uniform(-0.3, 0.3, size=(n, 3)).astype(np.float32) * scale[:, None]
    tri = np.concatenate([c, c + e1, c + e2], axis=1).astype(np.float32)
    cfg, meshes = mcrt.synth.random_scene(64, 8, seed=3)
    sd = mcrt.scene_io.build_scene(cfg, meshes)
    sd.tri = np.ascontiguousarray(tri); sd.tri_mesh = (np.arange(n) % len(sd.meshes)).astype(np.uint32)
    E, S = 16, 96
    out = {}
    for groups in ("1", "2", "2w", "1p", "1f"):
        monkeypatch.setenv("MCRT_GROUPS", groups[0])
        monkeypatch.setenv("MCRT_PATH_MAX", "1000000" if groups == "1f" else "0")      # "1f": the latency form (k_path), whose lanes stack into the same overflow array
        if groups == "1f":
            monkeypatch.delenv("MCRT_PACKET_BOUNCES", raising=False); monkeypatch.delenv("MCRT_PACKET_FROM", raising=False)
        if groups == "2w":
            monkeypatch.setenv("MCRT_WIDE_FROM", "1")
        if groups == "1p":          # every bounce a wavefront per ray packet: the packet's ONE 64-entry stack register against a 51-entry worst case
            monkeypatch.delenv("MCRT_WIDE_FROM", raising=False)
            monkeypatch.setenv("MCRT_PACKET_BOUNCES", "0xfffe"); monkeypatch.setenv("MCRT_PACKET_FROM", "0")
        tr, sim = _sim(mcrt, cfg, sd, E, S, texture=tex256)
        _, max_stack = sim.ctx.get_bvh4()
        dev = sim.ctx.alloc(2 * E * sim.R * 4)
        sim.ctx.trace_frames(4, 2, dev)
        out[groups] = sim.ctx.d2h(dev, (2, E, sim.R))
        sim.ctx.free(dev)
        if groups == "1":
            hits, _, _ = sim.ctx.trace_frame_debug(5, sim.rf_dev)
            nodes4 = sim.ctx.get_bvh4()[0]; _, btri, _ = sim.ctx.get_bvh()
        sim.close()
    for k in ("MCRT_GROUPS", "MCRT_WIDE_FROM", "MCRT_PACKET_BOUNCES", "MCRT_PACKET_FROM", "MCRT_PATH_MAX"):
        monkeypatch.delenv(k, raising=False)
    assert np.array_equal(out["1"].view(np.uint32), out["1f"].view(np.uint32))
    assert np.array_equal(out["1"].view(np.uint32), out["2"].view(np.uint32))
    assert np.array_equal(out["1"].view(np.uint32), out["2w"].view(np.uint32))
    assert np.array_equal(out["1"].view(np.uint32), out["1p"].view(np.uint32))
    osc = orc.OracleScene(sd.tri, sd.tri_mesh, sd.meshes, sd.materials, sd.start_mat, sd.spacing)
    osc.set_bvh4(nodes4, btri)
    o = osc.trace_frame(orc.default_params(n_elements=E, n_samples=S), tr.pos, tr.dir, tex256, frame_id=5, use_bvh=2, n_threads=16, want_ref=False)
    assert np.array_equal(hits, o["hits"])
    assert np.array_equal(out["2"][1].T.view(np.uint32), o["rf"].view(np.uint32))
    assert max_stack > 32, "the scene was meant to need the overflow stack (max_stack %d)" % max_stack


def test_per_frame_poses_in_one_pass(mcrt, orc, sphere, tex256):
    """mcrt_trace_frames_poses: the frames of a pass each with their own probe pose (the moving probe of transducer.h:82-118 /
    inputmanager.cpp:117-121) == the same frames traced one at a time with mcrt_set_transducer between them, bit for bit; one of
    them against the oracle; host and device pose tables alike; scan-line shards too"""
    cfg, sd = sphere
    E, S, F, f0 = 24, 96, 6, 40
    tr, sim = _sim(mcrt, cfg, sd, E, S, texture=tex256)
    base = np.asarray(cfg["transducerAngles"], np.float64)
    poses = []
    for f in range(F):                                           # the probe rotates and slides a little between frames
        t = mcrt.Transducer(E, position=np.asarray(cfg["transducerPosition"], np.float64) + np.array([0.0, 0.15 * f, -0.1 * f]),
                            angles_deg=base + np.array([1.5 * f, -0.7 * f, 2.0 * f]))
        poses.append(t)
    pos = np.stack([t.pos for t in poses]); dirs = np.stack([t.dir for t in poses])
    dev = sim.ctx.alloc(F * E * sim.R * 4)
    sim.ctx.trace_frames_poses(f0, pos, dirs, dev)
    batch = sim.ctx.d2h(dev, (F, E, sim.R))
    for f in range(F):
        sim.ctx.set_transducer(poses[f].pos, poses[f].dir)
        sim.trace(f0 + f)
        one = sim.ctx.d2h(sim.rf_dev, (E, sim.R))
        assert np.array_equal(batch[f].view(np.uint32), one.view(np.uint32)), f
    assert not np.array_equal(batch[0].view(np.uint32), batch[F - 1].view(np.uint32))
    osc = orc.OracleScene(sd.tri, sd.tri_mesh, sd.meshes, sd.materials, sd.start_mat, sd.spacing)
    o = osc.trace_frame(orc.default_params(n_elements=E, n_samples=S), poses[3].pos, poses[3].dir, tex256, frame_id=f0 + 3, use_bvh=False, n_threads=8, want_ref=False)
    assert np.array_equal(batch[3].T.view(np.uint32), o["rf"].view(np.uint32))
    # device-resident pose tables, and a shard of the scan-lines
    dpos, ddir = sim.ctx.alloc(pos.nbytes), sim.ctx.alloc(dirs.nbytes)
    sim.ctx.h2d(dpos, pos); sim.ctx.h2d(ddir, dirs)
    sim.ctx.trace_frames_poses(f0, dpos, ddir, dev, n_frames=F)
    assert np.array_equal(sim.ctx.d2h(dev, (F, E, sim.R)).view(np.uint32), batch.view(np.uint32))
    sim.ctx.trace_frames_poses(f0, dpos, ddir, dev, 5, 17, n_frames=F)
    assert np.array_equal(sim.ctx.d2h(dev, (F, 12, sim.R)).view(np.uint32), batch[:, 5:17].view(np.uint32))
    sim.ctx.free(dpos); sim.ctx.free(ddir)
    # the context's own transducer is untouched: a plain pass afterwards uses the pose set last with mcrt_set_transducer
    sim.ctx.trace_frames(f0 + F - 1, 1, dev)
    assert np.array_equal(sim.ctx.d2h(dev, (1, E, sim.R))[0].view(np.uint32), batch[F - 1].view(np.uint32))
    import ctypes
    assert sim.ctx.L.mcrt_trace_frames_poses(sim.ctx.h, 0, 2, 0, E, None, None, ctypes.c_void_p(dev)) == -1      # MCRT_ERR_INVALID: no pose tables
    sim.ctx.free(dev)
    sim.close()


def test_whole_bmode_frames_of_a_pass(mcrt, orc, sphere, tex256):
    """main.cpp:146-148 for every frame of a pass in three launches: mcrt_convolve_frames, mcrt_envelope_frames,
    mcrt_scan_convert_frames == the per-image calls == the oracle, bit for bit"""
    cfg, sd = sphere
    E, S, F = 64, 48, 4
    tr, sim = _sim(mcrt, cfg, sd, E, S, texture=tex256, sanitize_tir=1)
    R = sim.R
    dev = sim.ctx.alloc(F * E * R * 4); out = sim.ctx.alloc(F * 400 * 500 * 4); one = sim.ctx.alloc(400 * 500 * 4)
    sim.ctx.trace_frames(3, F, dev)
    raw = sim.ctx.d2h(dev, (F, E, R))
    sim.ctx.convolve_frames(dev, F, E, R, sim.psf.axial_kernel, sim.psf.lateral_kernel)
    sim.ctx.envelope_frames(dev, F, E, R)
    env = sim.ctx.d2h(dev, (F, E, R))
    sim.ctx.scan_convert_frames(dev, F, E, R, out)
    sc = sim.ctx.d2h(out, (F, 400, 500))
    ax, lat = orc.psf()
    for f in range(F):
        sim.trace(3 + f); sim.convolve()
        sim.ctx.envelope(sim.rf_dev, E, R)
        assert np.array_equal(env[f].view(np.uint32), sim.ctx.d2h(sim.rf_dev, (E, R)).view(np.uint32)), f
        sim.ctx.scan_convert(sim.rf_dev, E, R, one)
        assert np.array_equal(sc[f].view(np.uint32), sim.ctx.d2h(one, (400, 500)).view(np.uint32)), f
        ref_env = orc.envelope(orc.convolve(np.ascontiguousarray(raw[f].T), ax, lat))
        assert np.array_equal(env[f].T.view(np.uint32), ref_env.view(np.uint32)), f
        assert np.array_equal(sc[f].view(np.uint32), orc.scan_convert(ref_env).view(np.uint32)), f
    assert np.count_nonzero(sc[F - 1]) > 10000
    for d in (dev, out, one):
        sim.ctx.free(d)
    sim.close()


def test_scene_with_more_tables_than_fit_in_lds(mcrt, orc, tex256):
    """k_shade and k_march keep the material / mesh tables in LDS when they have at most 32 rows (round 4); a scene with 40 materials and
    36 meshes takes the other path -- the tables read from memory -- and must give the oracle's frame bit for bit as well; the same
    scene cut down to tables that fit gives its own oracle frame (both sides of the switch in one test)"""
    rng = np.random.default_rng(77)
    base = mcrt.synth.materials()
    for n_extra, n_mesh in ((31, 36), (0, 6)):
        mats = [dict(m) for m in base]
        for k in range(n_extra):            # copies of the tissue rows under new names, slightly different impedance / attenuation: every row is used
            m = dict(base[1 + k % (len(base) - 1)]); m["name"] = "X%d" % k
            m["impedance"] = float(m["impedance"]) * (1.0 + 0.01 * (k + 1)); m["attenuation"] = float(m["attenuation"]) * (1.0 + 0.02 * k)
            mats.append(m)
        names = [m["name"] for m in mats if m["name"] != "GEL"]
        cfg = {"transducerPosition": [-13.5, 0.0, 0.0], "transducerAngles": [0.0, 0.0, -90.0], "materials": mats, "meshes": [],
               "origin": [0.0, 0.0, 0.0], "spacing": [1.0, 1.0, 1.0], "scaling": 1.0, "startingMaterial": "GEL"}
        meshes = {}
        for i in range(n_mesh):
            f = "soup_%d.obj" % i
            meshes[f] = mcrt.synth.random_triangles(400, 900 + i, lo=(-10.0, -4.0, -4.0), hi=(2.0, 4.0, 4.0), edge=0.6)
            cfg["meshes"].append({"file": f, "rigid": True, "vascular": bool(i % 5 == 4), "deltas": [0.0, 0.0, 0.0],
                                  "material": names[(7 * i + 3) % len(names)], "outsideMaterial": names[(5 * i) % len(names)], "outsideNormals": True})
        sd = mcrt.scene_io.build_scene(cfg, meshes)
        assert sd.materials.shape[0] == len(base) + n_extra and len(sd.meshes) == n_mesh
        E, S = 12, 96
        tr, sim = _sim(mcrt, cfg, sd, E, S, texture=tex256)
        hits, _, _ = sim.ctx.trace_frame_debug(4, sim.rf_dev)
        rf = sim.ctx.export_rf(sim.rf_dev, E, sim.R)
        osc = orc.OracleScene(sd.tri, sd.tri_mesh, sd.meshes, sd.materials, sd.start_mat, sd.spacing)
        o = osc.trace_frame(orc.default_params(n_elements=E, n_samples=S), tr.pos, tr.dir, tex256, frame_id=4, use_bvh=False, n_threads=8)
        assert np.array_equal(hits, o["hits"]) and (hits >= 0).sum() > E * S
        assert np.array_equal(rf.view(np.uint32), o["rf"].view(np.uint32)) and np.isnan(rf).any()       # (AIR interfaces: total internal reflection, NaN bins included)
        # the frames of a pass too (k_march's pair variant and its tile sort)
        dev = sim.ctx.alloc(2 * E * sim.R * 4)
        sim.ctx.trace_frames(4, 2, dev)
        assert np.array_equal(sim.ctx.d2h(dev, (2, E, sim.R))[0].T.view(np.uint32), o["rf"].view(np.uint32))
        sim.ctx.free(dev)
        sim.close()
