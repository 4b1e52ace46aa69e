"""Host-side pieces of the product (no GPU): the reference's static tables computed by libmcrt_hip's host code must equal
the oracle's and the reference probe's; the BVH must be structurally sound and its walk must equal brute force."""
import hashlib
import os
import numpy as np


def test_texture_psf_transducer_match_oracle_and_reference(mcrt, orc, golden):
    t = mcrt.host_texture(256)
    digest = open(os.path.join(os.path.dirname(__file__), "golden", "texture_sha256.txt")).read().strip()
    assert hashlib.sha256(t.tobytes()).hexdigest() == digest
    assert np.array_equal(mcrt.host_texture(16), orc.texture(16))
    ax, lat = mcrt.host_psf()
    assert np.array_equal(ax.view(np.uint32), np.asarray(golden["psf_axial_bits"], np.uint32))
    assert np.array_equal(lat.view(np.uint32), np.asarray(golden["psf_lateral_bits"], np.uint32))
    for E, pos, ang in [(512, (-13.5, 0, 0), (0, 0, -90)), (128, (-17.5, 1.0, 5.0), (120.0, 0.0, -90.0)), (32, (-16, 3, 14), (45, 45, -90))]:
        tr = mcrt.Transducer(E, position=pos, angles_deg=ang)
        if E == 512:
            assert tr.separation_mm == golden["element_separation_mm"]
        po, do = orc.transducer(E, 3.0, tr.separation_mm, pos, ang)
        assert np.array_equal(po.view(np.uint32), tr.pos.view(np.uint32)) and np.array_equal(do.view(np.uint32), tr.dir.view(np.uint32))
        assert np.allclose(np.linalg.norm(tr.dir, axis=1), 1.0, atol=1e-6)


def _check_bvh(nodes, btri, n_tri, depth, pad_abs_fn):
    assert depth <= 32
    ids = btri[:, 3].copy().view(np.uint32)
    assert sorted(ids.tolist()) == list(range(n_tri)), "every triangle appears in exactly one leaf slot"
    seen = np.zeros(n_tri, bool)
    stack = [(0, 0)]
    maxd = 0
    while stack:
        n, d = stack.pop()
        nd = nodes[n]
        for c, lo, hi in ((nd["c0"], nd["lo0"], nd["hi0"]), (nd["c1"], nd["lo1"], nd["hi1"])):
            if c >= 0:
                ch = nodes[c]
                assert np.all(np.minimum(ch["lo0"], ch["lo1"]) >= lo) and np.all(np.maximum(ch["hi0"], ch["hi1"]) <= hi)
                stack.append((int(c), d + 1))
            else:
                v = (~int(c)) & 0xFFFFFFFF
                first, cnt = v >> 3, (v & 7) + 1
                maxd = max(maxd, d + 1)
                for k in range(first, first + cnt):
                    t = btri[k]
                    V = np.stack([t[0:3], t[4:7], t[8:11]])
                    assert np.all(V.min(0) >= lo) and np.all(V.max(0) <= hi)
                    seen[k] = True
    assert seen.all() and maxd <= 32


def test_bvh_structure_and_walk_equals_bruteforce(mcrt, orc):
    rng = np.random.default_rng(11)
    cases = [mcrt.synth.sphere_scene(3), mcrt.synth.random_scene(20000, 4, seed=7), mcrt.synth.liver_scene(2)]
    for cfg, meshes in cases:
        sd = mcrt.scene_io.build_scene(cfg, meshes)
        nodes, btri, depth = mcrt.host_build_bvh(sd.tri, sd.tri_mesh)
        _check_bvh(nodes, btri, sd.n_tri, depth, None)
        osc = orc.OracleScene(sd.tri, sd.tri_mesh, sd.meshes, sd.materials, sd.start_mat, sd.spacing, bvh=(nodes, btri))
        lo, hi = sd.tri.reshape(-1, 3).min(0) - 2, sd.tri.reshape(-1, 3).max(0) + 2
        nhit = 0
        for i in range(400):
            o = rng.uniform(lo, hi).astype(np.float32)
            d = sd.tri[rng.integers(sd.n_tri)][:3] + rng.normal(size=3) * 0.3 - o     # aim near the geometry
            d = d / np.linalg.norm(d)
            if i % 7 == 0:
                d[rng.integers(3)] = 0.0                      # axis-parallel rays: 1/0 = inf in the slab test
            if i % 11 == 0:
                o = sd.tri[rng.integers(sd.n_tri)][:3].copy()   # start exactly on a vertex
            to = (o + d * rng.choice([3.0, 40.0, 1e9])).astype(np.float32)
            a = osc.closest_hit(o, to, False); b = osc.closest_hit(o, to, True)
            assert a[0] == b[0] and a[1] == b[1] and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
            nhit += a[0] >= 0
        assert nhit > 25


def test_bvh4_collapse_and_row_thresholds(mcrt, orc):
    """the BVH4 the GPU walks: every BVH2 leaf reachable exactly once, child boxes equal the BVH2 boxes, stack bound
    honoured; walk equals brute force.  Row thresholds reproduce (int)(t / dt) for every t."""
    rng = np.random.default_rng(5)
    for cfg, meshes in [mcrt.synth.sphere_scene(3), mcrt.synth.random_scene(30000, 8, seed=3)]:
        sd = mcrt.scene_io.build_scene(cfg, meshes)
        nodes, btri, n4, max_stack = mcrt.host_build_bvh4(sd.tri, sd.tri_mesh)
        assert max_stack <= 64
        rec = n4.view(np.dtype([("lo", "<f4", 3), ("hi", "<f4", 3), ("ref", "<i4"), ("pad", "<u4")])).reshape(-1, 4)
        leaves = []
        stack = [0]
        seen_nodes = 0
        while stack:
            n = stack.pop(); seen_nodes += 1
            for c in rec[n]:
                if c["ref"] == -2 ** 31:
                    continue
                if c["ref"] >= 0:
                    stack.append(int(c["ref"]))
                else:
                    v = (~int(c["ref"])) & 0xFFFFFFFF
                    leaves.append((v >> 3, (v & 7) + 1))
                    for k in range(v >> 3, (v >> 3) + (v & 7) + 1):
                        V = np.stack([btri[k][0:3], btri[k][4:7], btri[k][8:11]])
                        assert np.all(V.min(0) >= c["lo"]) and np.all(V.max(0) <= c["hi"])
        assert seen_nodes == len(rec)
        cover = np.zeros(sd.n_tri, int)
        for f, c in leaves:
            assert c <= 4
            cover[f:f + c] += 1
        assert np.all(cover == 1)
        osc = orc.OracleScene(sd.tri, sd.tri_mesh, sd.meshes, sd.materials, sd.start_mat, sd.spacing, bvh=(nodes, btri))
        osc.set_bvh4(n4)
        for i in range(300):
            o = rng.uniform(-9, 9, 3).astype(np.float32)
            d = sd.tri[rng.integers(sd.n_tri)][:3] + rng.normal(size=3) * 0.3 - o
            d /= np.linalg.norm(d)
            if i % 5 == 0:
                d[rng.integers(3)] = 0.0
            to = (o + d * rng.choice([4.0, 40.0, 1e9])).astype(np.float32)
            a = osc.closest_hit(o, to, 0); c = osc.closest_hit(o, to, 2)
            assert a[0] == c[0] and a[1] == c[1]
    for dt, R in [(322 / 1500.0, 465), (322 / 1500.0, 512), (0.1234567, 100)]:
        thr = mcrt.host_row_thresholds(dt, R)
        t = np.concatenate([rng.uniform(0, (R + 2) * dt, 400000), thr, np.nextafter(thr, -1), np.nextafter(thr, 1e9), np.arange(R + 2) * dt])
        t = t[t >= 0]
        ref = np.where(t / dt < R, np.floor(t / dt), -1).astype(np.int64)
        got = np.searchsorted(thr, t, side="right") - 1
        got = np.where(t < thr[R], got, -1)
        assert np.array_equal(ref, got)


def test_bvh_degenerate_inputs(mcrt, orc):
    # one triangle; identical triangles; a zero-area triangle
    one = np.array([[0, 0, 0, 1, 0, 0, 0, 1, 0]], np.float32)
    for tri in (one, np.repeat(one, 9, 0), np.concatenate([one, np.zeros((1, 9), np.float32)])):
        tm = np.zeros(len(tri), np.uint32)
        nodes, btri, depth = mcrt.host_build_bvh(tri, tm)
        osc = orc.OracleScene(tri, tm, [(0, 0, 0)], np.ones((1, 8), np.float32), 0, bvh=(nodes, btri))
        a = osc.closest_hit((0.2, 0.2, 1), (0.2, 0.2, -1), False); b = osc.closest_hit((0.2, 0.2, 1), (0.2, 0.2, -1), True)
        assert a[0] == b[0] == 0 and a[1] == b[1] == 0.5
        assert osc.closest_hit((2, 2, 1), (2, 2, -1), True)[0] == -1


def test_single_triangle_known_answers(orc):
    """analytic KATs for the triangle arithmetic: fraction, hit point, normal orientation, back-face, edge tolerance"""
    tri = np.array([[0, 0, 0, 1, 0, 0, 0, 1, 0]], np.float32)
    osc = orc.OracleScene(tri, np.zeros(1, np.uint32), [(0, 0, 0)], np.ones((1, 8), np.float32), 0)
    t, f, n, p, _ = osc.closest_hit((0.25, 0.25, 2), (0.25, 0.25, -2))
    assert t == 0 and f == 0.5 and np.array_equal(p, [0.25, 0.25, 0]) and np.array_equal(n, [0, 0, 1])     # faces the origin
    t, f, n, p, _ = osc.closest_hit((0.25, 0.25, -2), (0.25, 0.25, 2))
    assert t == 0 and np.array_equal(n, [0, 0, -1])                                                            # flipped for the back side
    assert osc.closest_hit((0.25, 0.25, 2), (0.25, 0.25, 0.5))[0] == -1                                        # segment ends before the plane
    assert osc.closest_hit((0.9, 0.9, 2), (0.9, 0.9, -2))[0] == -1                                             # outside
    assert osc.closest_hit((-0.00004, 0.5, 1), (-0.00004, 0.5, -1))[0] == 0                                   # inside Bullet's 1e-4 edge tolerance
    assert osc.closest_hit((-0.001, 0.5, 1), (-0.001, 0.5, -1))[0] == -1
    assert osc.closest_hit((0.25, 0.25, 0.0), (0.25, 0.25, -1))[0] == -1                                       # dist_a*dist_b >= 0: starts on the plane


def test_tuning_knobs_are_gated(mcrt, monkeypatch):
    """the library's environment knobs are read only under MCRT_TUNING=1 (csrc/mcrt_host.cpp tuning_env): here the SAH builder's leaf size"""
    rng = np.random.default_rng(4)
    V, F = mcrt.synth.random_triangles(3000, seed=4)
    tri = V[F].reshape(-1, 9); tm = np.zeros(len(tri), np.uint32)
    monkeypatch.delenv("MCRT_TUNING", raising=False)
    monkeypatch.delenv("MCRT_SAH_LEAF_MAX", raising=False); monkeypatch.delenv("MCRT_SAH_COST_TRI", raising=False)
    base = mcrt.host_build_bvh(tri, tm)[0]
    monkeypatch.setenv("MCRT_SAH_LEAF_MAX", "8"); monkeypatch.setenv("MCRT_SAH_COST_TRI", "0.001")
    ignored = mcrt.host_build_bvh(tri, tm)[0]
    assert ignored.tobytes() == base.tobytes()                  # no MCRT_TUNING: the knobs are not even looked at
    monkeypatch.setenv("MCRT_TUNING", "1")
    tuned = mcrt.host_build_bvh(tri, tm)[0]
    assert len(tuned) < len(base) // 2                          # eight cheap triangles per leaf: far fewer nodes
