#!/usr/bin/env python3
"""Would a closest-hit bound known BEFORE the walk pay?  Counted on the CPU.

A wavefront of the walk steps 45 lanes on 13 distinct nodes (profiles/round4/exp_refill_threshold.txt): the rays of a bundle -- same
scan-line, same reflect/refract history -- mostly hit the same triangles.  A closest-hit bound seeded from a neighbour is EXACT under
the contract (any real hit is an upper bound; the id rule still decides ties).  Per bounce, on whole frames, the closest-hit queries are
walked over the product's BVH4 in the GPU's order
    none     as today
    perfect  the query's own answer as the initial bound (what no seed can beat)
    leader   the answer of the bundle's first ray, tested against this ray's own segment first
and BVH4 node visits / triangle tests per query are counted.  Bundle = same scan-line, same sequence of triangles hit so far, same side
of the last triangle (reflected / refracted); its leader = the lowest sample index; the leader itself walks unseeded.

    python tools/seed_count.py [workload=random1m|liver|sphere] [rays=1024] [frames=1]     -> JSON (profiles/round5/seed_count_*.json)
"""
import ctypes as C, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mcray_tracing_amd as m
from oracle import orc

workload = sys.argv[1] if len(sys.argv) > 1 else "random1m"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 1
E = 128
if workload == "random1m":
    cfg, meshes = m.synth.random_scene(1_000_000, 8, 12345)
elif workload == "liver":
    cfg, meshes = m.synth.liver_scene(5)
else:
    cfg, meshes = m.synth.sphere_scene(5)
sd = m.scene_io.build_scene(cfg, meshes)
tr = m.Transducer(E, position=cfg["transducerPosition"], angles_deg=cfg["transducerAngles"])
nodes, btri, n4, _ = m.host_build_bvh4(sd.tri, sd.tri_mesh)
osc = orc.OracleScene(sd.tri, sd.tri_mesh, sd.meshes, sd.materials, sd.start_mat, sd.spacing, bvh=(nodes, btri))
osc.set_bvh4(n4)
tex = orc.texture(256)
p = orc.default_params(n_elements=E, n_samples=S)
L = orc.lib()
L.orc_seed_count.restype = None
L.orc_seed_count.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
tri_n = np.cross(sd.tri[:, 3:6] - sd.tri[:, 0:3], sd.tri[:, 6:9] - sd.tri[:, 0:3]).astype(np.float64)

B = p.max_depth
acc = {mode: np.zeros((B, 2), np.float64) for mode in ("none", "perfect", "leader")}
nq = np.zeros(B, np.int64); n_seeded = np.zeros(B, np.int64); n_seed_hit = np.zeros(B, np.int64); n_same = np.zeros(B, np.int64); n_bundles = np.zeros(B, np.int64)
t0 = time.time()
for f in range(frames):
    o = osc.trace_frame(p, tr.pos, tr.dir, tex, frame_id=f, use_bvh=2, n_threads=os.cpu_count(), want_segs=True, want_ref=False, want_fix=False)
    segs, hits, cnt = o["segs"], o["hits"], o["seg_count"]
    # bundle key per (e, s, b): hash of the triangles hit on bounces < b and of the side of the last one the ray left on
    key = np.zeros((E, S), np.uint64)
    for b in range(B):
        live = cnt > b
        if not live.any():
            break
        if b > 0:
            prev = hits[:, :, b - 1].astype(np.int64)
            side = (np.einsum("esk,esk->es", segs["dir"][:, :, b].astype(np.float64), tri_n[np.maximum(prev, 0)]) > 0).astype(np.uint64)
            key = (key * np.uint64(0x9E3779B97F4A7C15) + (prev.astype(np.uint64) * np.uint64(2) + side + np.uint64(1))) & np.uint64(0xFFFFFFFFFFFFFFFF)
        es = np.argwhere(live)
        e_idx, s_idx = es[:, 0], es[:, 1]
        k = key[e_idx, s_idx] ^ (e_idx.astype(np.uint64) << np.uint64(50))
        order = np.lexsort((s_idx, k))
        ks = k[order]
        first = np.concatenate([[True], ks[1:] != ks[:-1]])
        leader_pos = np.maximum.accumulate(np.where(first, np.arange(len(ks)), 0))
        leader_of = np.empty(len(ks), np.int64); leader_of[order] = order[leader_pos]          # index (into es) of each query's leader
        own = hits[e_idx, s_idx, b]
        seed = np.where(leader_of == np.arange(len(ks)), -1, own[leader_of]).astype(np.int32)
        q = np.ascontiguousarray(segs[e_idx, s_idx, b])
        n = len(q)
        for mode, name in ((0, "none"), (1, "perfect"), (2, "leader")):
            out = np.zeros((n, 2), np.uint32); tri = np.zeros(n, np.int32)
            L.orc_seed_count(C.byref(osc.c), C.byref(p), q.ctypes.data, n, mode, seed.ctypes.data, out.ctypes.data, tri.ctypes.data, os.cpu_count())
            assert np.array_equal(tri, own), "mode %s found another triangle on bounce %d" % (name, b)
            acc[name][b] += out.sum(0)
        nq[b] += n; n_seeded[b] += int((seed >= 0).sum()); n_same[b] += int(((seed >= 0) & (seed == own)).sum()); n_bundles[b] += int(first.sum())

rows = []
for b in range(B):
    if nq[b] == 0:
        continue
    r = {"bounce": b, "queries": int(nq[b]), "bundles": int(n_bundles[b]), "rays_per_bundle": float(nq[b] / n_bundles[b]),
         "seeded_share": float(n_seeded[b] / nq[b]), "seed_is_the_answer_share": float(n_same[b] / nq[b])}
    for name in acc:
        r["nodes_" + name] = float(acc[name][b, 0] / nq[b]); r["tris_" + name] = float(acc[name][b, 1] / nq[b])
    rows.append(r)
tot = {name: acc[name].sum(0) / nq.sum() for name in acc}
# bounce 0 is walked once per scan-line by the product already: the figure that matters is bounces >= 1
d = nq[1:].sum()
deep = {name: acc[name][1:].sum(0) / d for name in acc}
print(json.dumps({"workload": workload, "scan_lines": E, "rays": S, "frames": frames, "seconds": round(time.time() - t0, 1),
                  "per_bounce": rows,
                  "all_bounces": {k: {"nodes": float(v[0]), "tris": float(v[1])} for k, v in tot.items()},
                  "bounces_ge1": {k: {"nodes": float(v[0]), "tris": float(v[1])} for k, v in deep.items()},
                  "visits_removed_bounces_ge1": {"perfect": float(1 - deep["perfect"][0] / deep["none"][0]), "leader": float(1 - deep["leader"][0] / deep["none"][0])}}, indent=1))
