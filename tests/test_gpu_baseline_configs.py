"""GPU parity at the sizes BASELINE.json quotes (run on the MI355X box).

Every configuration is traced on the GPU at FULL size and compared with the CPU oracle on the same seeded inputs:
  headline  1 M random triangles, 128 scan-lines x 1024 rays        (the bench workload: `metric`)
  C3        liver-like scene (225 280 triangles), 128 x 4096
  C4        1 M random triangles, 256 x 8192
  C5        liver-like scene, 512 x 16384, + PSF convolution
(C1 and C2, the sphere configurations, are in test_gpu_parity.py.)  The oracle checks EVERY scan-line of every configuration (since
round 5 also C5's: ~40 s of the box's 16 cores), in blocks of scan-lines (its e_begin/e_end range); the GPU traces the whole frame at once.  Bars: hit indices bit-exact; fixed-point RF image bit-exact; reference-order float
image within 1e-4 both relative to the peak and element-wise (|d| <= 1e-4 |ref| + 1e-6 peak; see assert_rf for how the
reference's own float accumulation noise is kept out of the comparison at thousands of samples per scan-line); BVH node / triangle
visit counts equal to the oracle's walk of the same tree (they are the roofline's algorithmic bytes)."""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL_REF = 1e-4          # north_star: RF image within 1e-4 relative of the CPU reference path
ATOL_FLOOR = 1e-6        # element-wise absolute floor, in units of the image peak


def assert_rf(rf_gpu, o, cols=None):
    """rf_gpu [R][E] against the oracle: (1) the contract image, bit for bit; (2) the reference's own summation order
    (main.cpp:106-144) within 1e-4, peak-relative AND element-wise.  The reference adds echoes into a float image
    (`cv::Mat += echo`, rfimage.h:38); that running float sum carries its own rounding error, which grows with the samples per
    scan-line (measured: 3e-7 of the peak at 1024, 1.5e-4 at 16384).  Where the oracle also carried the same order in double
    (`rf_ref64`), the 1e-4 bars are applied to THAT image -- the reference's arithmetic without its accumulation noise -- and the
    float image must differ from the GPU's by no more than it differs from its own double-precision twin (+10 %)."""
    want, ref = o["rf"], o["rf_ref"]
    if cols is not None:
        rf_gpu = rf_gpu[:, cols[0]:cols[1]]
    assert np.array_equal(rf_gpu.view(np.uint32), want.view(np.uint32)), "fixed-point RF not bit-exact"
    assert np.array_equal(np.isnan(rf_gpu), np.isnan(ref))
    m = ~np.isnan(ref)
    peak = np.abs(ref[m]).max()
    d32 = np.abs(rf_gpu[m] - ref[m]).max() / peak
    if o.get("rf_ref64") is not None:
        ref64 = o["rf_ref64"]
        d = np.abs(rf_gpu[m].astype(np.float64) - ref64[m])
        own = np.abs(ref[m].astype(np.float64) - ref64[m]).max() / peak          # the float running sum against itself in double
        assert d32 <= 1.1 * own + 1e-6, "float reference image: off by %g of the peak, its own rounding is %g" % (d32, own)
        ref_abs = np.abs(ref64[m])
    else:
        d = np.abs(rf_gpu[m] - ref[m])
        ref_abs = np.abs(ref[m])
    assert d.max() <= RTOL_REF * peak, "relative to the peak: %g" % (d.max() / peak)
    assert np.all(d <= RTOL_REF * ref_abs + ATOL_FLOOR * peak), "element-wise: worst excess %g of the peak" % ((d - RTOL_REF * ref_abs).max() / peak)
    return float(d.max() / peak), float(d32)


def _setup(mcrt, orc, cfg, sd, E, S, tex, **kw):
    tr = mcrt.Transducer(E, position=cfg["transducerPosition"], angles_deg=cfg["transducerAngles"])
    sim = mcrt.Simulator(sd, tr, n_samples=S, texture=tex, **kw)
    nodes, btri, _ = sim.ctx.get_bvh()
    osc = orc.OracleScene(sd.tri, sd.tri_mesh, sd.meshes, sd.materials, sd.start_mat, sd.spacing, bvh=(nodes, btri))
    osc.set_bvh4(sim.ctx.get_bvh4()[0])
    return tr, sim, osc


def _blocks(E, width, n, seed):
    """n disjoint scan-line blocks of `width`, seeded; always includes the first and the last block of the frame"""
    starts = list(range(0, E - width + 1, width))
    if n is None or os.environ.get("MCRT_FULL_ORACLE"):          # the oracle on EVERY scan-line (C4: ~16 s of 16 cores; C5: ~40 s)
        return [(s0, s0 + width) for s0 in starts]
    rng = np.random.default_rng(seed)
    pick = {0, len(starts) - 1}
    while len(pick) < min(n, len(starts)):
        pick.add(int(rng.integers(0, len(starts))))
    return [(starts[i], starts[i] + width) for i in sorted(pick)]


def _check_blocks(orc, osc, tr, tex, hits, rf, p, frame, blocks, threads):
    worst = 0.0
    for b0, b1 in blocks:
        o = osc.trace_frame(p, tr.pos, tr.dir, tex, frame_id=frame, e_begin=b0, e_end=b1, use_bvh=2, n_threads=threads, want_ref64=True)
        assert np.array_equal(hits[b0:b1], o["hits"]), "hit indices differ in scan-lines [%d,%d)" % (b0, b1)
        worst = max(worst, assert_rf(rf, o, (b0, b1))[1])
    return worst


def _visit_counts(mcrt, orc, sim, osc, tr, tex, p_kw, frame, S, o_stats, threads):
    """the counting build of the kernels against the oracle's walk (bounce 0 is walked once per scan-line on the GPU)"""
    sim.ctx.enable_stats(True); sim.ctx.get_stats(reset=True)
    sim.trace(frame); st = sim.ctx.get_stats()
    sim.ctx.enable_stats(False)
    p0 = orc.default_params(max_depth=1, **p_kw)
    o0 = osc.trace_frame(p0, tr.pos, tr.dir, tex, frame_id=frame, use_bvh=2, n_threads=threads, want_hits=False, want_ref=False, want_fix=False)["stats"]
    for k in ("queries", "nodes_visited", "tris_tested"):
        assert o0[k] % S == 0
        assert st[k] == o_stats[k] - o0[k] + o0[k] // S, k
    for k in ("segments", "hits"):
        assert st[k] == o_stats[k], k


def _tree_against_brute_force(mcrt, orc, cfg, sd, sim, E, tex, threads, frame=5, S2=64, n_lines=2, seed=2026):
    """The tree the PRODUCT built, at this size (the counted walk of the big tests runs over the product's own tree, so a
    builder that lost triangles at scale would pass): every triangle id sits in the leaf-order array exactly once, and on the same
    tree a block of rays of every bounce -- seeded scan-lines x 64 samples, ~500 closest-hit queries per line, each over ALL
    triangles -- gives the BRUTE-FORCE answer, on the GPU and in the oracle's walk alike.  Closes `sim`."""
    _, btri, _ = sim.ctx.get_bvh()
    assert np.array_equal(np.sort(btri.view(np.uint32)[:, 3]), np.arange(sd.n_tri, dtype=np.uint32))
    sim.close()
    tr2, sim2, osc2 = _setup(mcrt, orc, cfg, sd, E, S2, tex)
    hits2, _, _ = sim2.ctx.trace_frame_debug(frame, sim2.rf_dev)
    p2 = orc.default_params(n_elements=E, n_samples=S2)
    n_q = 0
    for e in sorted(int(x) for x in np.random.default_rng(seed).choice(E, size=n_lines, replace=False)):
        brute = osc2.trace_frame(p2, tr2.pos, tr2.dir, tex, frame_id=frame, e_begin=e, e_end=e + 1, use_bvh=0, n_threads=threads, want_ref=False, want_fix=False)
        walked = osc2.trace_frame(p2, tr2.pos, tr2.dir, tex, frame_id=frame, e_begin=e, e_end=e + 1, use_bvh=2, n_threads=threads, want_ref=False, want_fix=False)
        assert np.array_equal(brute["hits"], walked["hits"]) and np.array_equal(brute["hits"][0], hits2[e]), e
        n_q += brute["stats"]["queries"] - S2                        # (bounces >= 1)
    sim2.close()
    return n_q


def test_headline_1m_triangles_128x1024(mcrt, orc, tex256):
    """the workload `metric` is quoted on and bench.py times: every scan-line against the oracle"""
    cfg, meshes = mcrt.synth.random_scene(1_000_000, 8, 12345)
    sd = mcrt.scene_io.build_scene(cfg, meshes)
    E, S, frame = 128, 1024, 0
    threads = os.cpu_count() or 8
    tr, sim, osc = _setup(mcrt, orc, cfg, sd, E, S, tex256)
    hits, _, _ = sim.ctx.trace_frame_debug(frame, sim.rf_dev)
    rf = sim.ctx.export_rf(sim.rf_dev, E, sim.R)
    p_kw = dict(n_elements=E, n_samples=S)
    p = orc.default_params(**p_kw)
    o = osc.trace_frame(p, tr.pos, tr.dir, tex256, frame_id=frame, use_bvh=2, n_threads=threads, want_ref64=True)
    assert np.array_equal(hits, o["hits"])
    assert (hits >= 0).sum() > 2 * E * S                      # the soup is hit, several bounces deep
    d64, d32 = assert_rf(rf, o)
    assert d32 <= RTOL_REF                                      # at 1024 samples per scan-line even the float running sum is within the bar
    _visit_counts(mcrt, orc, sim, osc, tr, tex256, p_kw, frame, S, o["stats"], threads)
    # the batched pass the bench times (frames in flight) reproduces the frame bit for bit
    F = 4
    dev = sim.ctx.alloc(F * E * sim.R * 4)
    sim.ctx.trace_frames(0, F, dev)
    batch = sim.ctx.d2h(dev, (F, E, sim.R))
    assert np.array_equal(batch[0].T.view(np.uint32), o["rf"].view(np.uint32))
    o3 = osc.trace_frame(p, tr.pos, tr.dir, tex256, frame_id=3, use_bvh=2, n_threads=threads, want_hits=False, want_ref=False)
    assert np.array_equal(batch[3].T.view(np.uint32), o3["rf"].view(np.uint32))
    sim.ctx.free(dev)
    assert _tree_against_brute_force(mcrt, orc, cfg, sd, sim, E, tex256, threads) >= 512


def test_c3_liver_128x4096(mcrt, orc, tex256):
    """BASELINE config 3 at the benchmarked size: liver_scene(5) = 225 280 triangles, 128 scan-lines x 4096 rays"""
    cfg, meshes = mcrt.synth.liver_scene(5)
    sd = mcrt.scene_io.build_scene(cfg, meshes)
    assert sd.n_tri == 225280
    E, S, frame = 128, 4096, 2
    threads = os.cpu_count() or 8
    tr, sim, osc = _setup(mcrt, orc, cfg, sd, E, S, tex256)
    hits, _, _ = sim.ctx.trace_frame_debug(frame, sim.rf_dev)
    rf = sim.ctx.export_rf(sim.rf_dev, E, sim.R)
    p_kw = dict(n_elements=E, n_samples=S)
    p = orc.default_params(**p_kw)
    o = osc.trace_frame(p, tr.pos, tr.dir, tex256, frame_id=frame, use_bvh=2, n_threads=threads, want_ref64=True)
    assert np.array_equal(hits, o["hits"])
    assert (hits >= 0).sum() > E * S
    assert_rf(rf, o)
    _visit_counts(mcrt, orc, sim, osc, tr, tex256, p_kw, frame, S, o["stats"], threads)
    assert _tree_against_brute_force(mcrt, orc, cfg, sd, sim, E, tex256, threads, n_lines=4) >= 512      # (the liver scene's tree: C3 and C5)


def test_c4_1m_triangles_256x8192(mcrt, orc, tex256):
    """BASELINE config 4: 1 M random triangles, 256 scan-lines x 8192 rays (2.1 M paths) -- GPU traces the full frame, the
    oracle checks EVERY scan-line (round 4: the whole frame by default, in blocks of 16 scan-lines)"""
    cfg, meshes = mcrt.synth.random_scene(1_000_000, 8, 12345)
    sd = mcrt.scene_io.build_scene(cfg, meshes)
    E, S, frame = 256, 8192, 1
    threads = os.cpu_count() or 8
    tr, sim, osc = _setup(mcrt, orc, cfg, sd, E, S, tex256)
    hits, _, _ = sim.ctx.trace_frame_debug(frame, sim.rf_dev)
    rf = sim.ctx.export_rf(sim.rf_dev, E, sim.R)
    p = orc.default_params(n_elements=E, n_samples=S)
    _check_blocks(orc, osc, tr, tex256, hits, rf, p, frame, _blocks(E, 16, None, seed=4), threads)
    # scan-line shards (what 2/4/8 GPUs would each trace) reproduce the full frame bit for bit
    full = sim.ctx.d2h(sim.rf_dev, (E, sim.R))
    for g in (1, 6):                                              # two of the eight shards
        sim.ctx.trace_frame(frame, sim.rf_dev, g * 32, g * 32 + 32)
        part = sim.ctx.d2h(sim.rf_dev, (32, sim.R))
        assert np.array_equal(part.view(np.uint32), full[g * 32:g * 32 + 32].view(np.uint32))
    sim.close()


def test_c5_liver_512x16384_psf(mcrt, orc, tex256):
    """BASELINE config 5: liver-like scene, 512 scan-lines x 16384 rays (8.4 M paths) + PSF convolution -- GPU traces and
    convolves the full frame; the oracle checks EVERY scan-line (round 5: the whole frame by default, in sixteen blocks of 32 scan-lines,
    each traced once by the oracle: hits, fixed-point and reference-order RF, and the convolution of the block's interior -- the lateral pass
    reads 12 columns to the right, rfimage.h:113-118)"""
    cfg, meshes = mcrt.synth.liver_scene(5)
    sd = mcrt.scene_io.build_scene(cfg, meshes)
    E, S, frame = 512, 16384, 7
    threads = os.cpu_count() or 8
    tr, sim, osc = _setup(mcrt, orc, cfg, sd, E, S, tex256)
    hits, _, _ = sim.ctx.trace_frame_debug(frame, sim.rf_dev)
    rf = sim.ctx.export_rf(sim.rf_dev, E, sim.R)
    p = orc.default_params(n_elements=E, n_samples=S)
    # PSF convolution of the whole frame on the GPU == the oracle's convolution of the GPU's raw image (verified block by block below)
    sim.convolve()
    rfc = sim.ctx.export_rf(sim.rf_dev, E, sim.R)
    oc = orc.convolve(rf, sim.psf.axial_kernel, sim.psf.lateral_kernel)
    assert np.array_equal(rfc.view(np.uint32), oc.view(np.uint32))
    blocks = _blocks(E, 32, None, seed=5)
    assert blocks[0][0] == 0 and blocks[-1][1] == E and len(blocks) == 16
    for b0, b1 in blocks:
        o = osc.trace_frame(p, tr.pos, tr.dir, tex256, frame_id=frame, e_begin=b0, e_end=b1, use_bvh=2, n_threads=threads, want_ref64=True)
        assert np.array_equal(hits[b0:b1], o["hits"]), "hit indices differ in scan-lines [%d,%d)" % (b0, b1)
        assert_rf(rf, o, (b0, b1))
        # ... and, independently of the GPU's raw image, the convolved frame inside the block
        ob = orc.convolve(o["rf"], sim.psf.axial_kernel, sim.psf.lateral_kernel)          # [R][32]: columns 6..18 of the block are complete
        lo, hi = 6, (b1 - b0) - 13
        assert np.array_equal(rfc[:, b0 + lo:b0 + hi].view(np.uint32), ob[:, lo:hi].view(np.uint32))
    sim.close()


def test_random16m_streaming_scene(mcrt, orc, tex256):
    """The streaming regime bench.py --workload random16m measures: 16 M random triangles (1.5 GB of triangle records,
    ~0.5 GB of walked nodes: past the 256 MiB Infinity Cache; inside the walk's 2^25-triangle / 32-bit-offset limits), indexed by the
    DEVICE builder.  A seeded block of scan-lines against the oracle walking the product's own tree (hits, fixed-point RF, visit counts),
    every triangle id once in the leaf array, and ~500 closest-hit queries of every bounce against BRUTE FORCE over all 16 M triangles."""
    cfg, meshes = mcrt.synth.random_scene(16_000_000, 8, 12345, edge=0.025)
    sd = mcrt.scene_io.build_scene(cfg, meshes)
    del meshes
    assert sd.n_tri == 16_000_000
    E, S, frame = 16, 256, 3
    threads = os.cpu_count() or 8
    tr, sim, osc = _setup(mcrt, orc, cfg, sd, E, S, tex256, bvh_builder="lbvh")
    hits, _, _ = sim.ctx.trace_frame_debug(frame, sim.rf_dev)
    rf = sim.ctx.export_rf(sim.rf_dev, E, sim.R)
    p_kw = dict(n_elements=E, n_samples=S)
    p = orc.default_params(**p_kw)
    o = osc.trace_frame(p, tr.pos, tr.dir, tex256, frame_id=frame, use_bvh=2, n_threads=threads, want_ref64=True)
    assert np.array_equal(hits, o["hits"])
    assert (hits >= 0).sum() > 2 * E * S
    assert_rf(rf, o)
    _visit_counts(mcrt, orc, sim, osc, tr, tex256, p_kw, frame, S, o["stats"], threads)
    _, btri, _ = sim.ctx.get_bvh()
    assert np.array_equal(np.sort(btri.view(np.uint32)[:, 3]), np.arange(sd.n_tri, dtype=np.uint32))
    del btri
    # brute force over all 16 M triangles: one scan-line x 64 samples of a fresh small frame on the same tree
    sim.close()
    E2, S2 = 4, 64
    tr2, sim2, osc2 = _setup(mcrt, orc, cfg, sd, E2, S2, tex256, bvh_builder="lbvh")
    hits2, _, _ = sim2.ctx.trace_frame_debug(5, sim2.rf_dev)
    p2 = orc.default_params(n_elements=E2, n_samples=S2)
    brute = osc2.trace_frame(p2, tr2.pos, tr2.dir, tex256, frame_id=5, e_begin=1, e_end=2, use_bvh=0, n_threads=threads, want_ref=False, want_fix=False)
    assert np.array_equal(brute["hits"][0], hits2[1]) and brute["stats"]["queries"] >= 300
    sim2.close()
