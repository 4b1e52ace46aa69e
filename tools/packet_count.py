#!/usr/bin/env python3
"""north_star's literal traversal design -- "one wavefront per ray packet" -- COUNTED on the CPU before it is built.

64 consecutive rays of a bounce's queue (queue order: scan-line major, the sample paths of one reflect / refract history are neighbours; reflected
before refracted inside a wavefront as k_shade's compaction leaves them) walk the product's BVH4 together: one shared stack, a node is visited when
ANY ray passes that child's box with its own closest fraction, every ray tests every triangle of a visited leaf.  Per bounce: nodes a packet visits
against the sum (the lane walk's work) and the maximum (the lane walk's chain) of its rays' solo visits.

    python tools/packet_count.py [workload=random1m|liver|sphere] [rays=1024] [order=1]     -> JSON (profiles/round5/packet_count_*.json)
"""
import ctypes as C, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mcray_tracing_amd as m
from oracle import orc

workload = sys.argv[1] if len(sys.argv) > 1 else "random1m"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
order = int(sys.argv[3]) if len(sys.argv) > 3 else 1
pure = len(sys.argv) > 4 and sys.argv[4] == "pure"       # packets cut from PURE bundles (the queue sorted by reflect / refract history: what a sorting compaction would give)
wgkey = len(sys.argv) > 4 and sys.argv[4] == "wgkey"     # 256-ray blocks grouped by (triangle hit last, decision): largest group first (what a workgroup-local grouping in k_shade would give)
stray = len(sys.argv) > 4 and sys.argv[4] == "stray"     # 256-ray blocks ordered: rays that hit their wavefront's dominant triangle (reflected, then refracted), then the strays
segdec = len(sys.argv) > 4 and sys.argv[4] == "segdec"   # round 6: the WHOLE (scan-line, frame) segment partitioned by its rays' reflect / refract decisions so far (no triangle ids): what a segmented partition after k_shade would give
blk3 = len(sys.argv) > 4 and sys.argv[4] == "blk3"       # k_shade's 256-ray blocks sorted by the class of the last three reflect / refract decisions (a counting sort inside the workgroup)
dec = None
E, W = 128, 64
cfg, meshes = {"random1m": lambda: m.synth.random_scene(1_000_000, 8, 12345), "liver": lambda: m.synth.liver_scene(5), "sphere": lambda: m.synth.sphere_scene(5)}[workload]()
sd = m.scene_io.build_scene(cfg, meshes)
tr = m.Transducer(E, position=cfg["transducerPosition"], angles_deg=cfg["transducerAngles"])
nodes, btri, n4, _ = m.host_build_bvh4(sd.tri, sd.tri_mesh)
osc = orc.OracleScene(sd.tri, sd.tri_mesh, sd.meshes, sd.materials, sd.start_mat, sd.spacing, bvh=(nodes, btri)); osc.set_bvh4(n4)
tex = orc.texture(256)
p = orc.default_params(n_elements=E, n_samples=S)
L = orc.lib()
L.orc_packet_count.restype = None
L.orc_packet_count.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
tri_n = np.cross(sd.tri[:, 3:6] - sd.tri[:, 0:3], sd.tri[:, 6:9] - sd.tri[:, 0:3]).astype(np.float64)
t0 = time.time()
o = osc.trace_frame(p, tr.pos, tr.dir, tex, frame_id=0, use_bvh=2, n_threads=os.cpu_count(), want_segs=True, want_ref=False, want_fix=False)
segs, hits, cnt = o["segs"], o["hits"], o["seg_count"]
rows = []
key = np.zeros((E, S), np.uint64)
for b in range(1, p.max_depth):
    live = cnt > b
    if not live.any(): break
    es = np.argwhere(live); e_idx, s_idx = es[:, 0], es[:, 1]
    # k_shade's compaction inside a wavefront of the PREVIOUS queue: reflected rays first, then refracted (approximated on the path order: blocks of 64 paths)
    prev = hits[e_idx, s_idx, b - 1]
    refl = (np.einsum("nk,nk->n", segs["dir"][e_idx, s_idx, b].astype(np.float64), tri_n[np.maximum(prev, 0)]) *
            np.einsum("nk,nk->n", segs["dir"][e_idx, s_idx, b - 1].astype(np.float64), tri_n[np.maximum(prev, 0)]) < 0)
    blk = (e_idx * S + s_idx) // 64
    side = (np.einsum("esk,esk->es", segs["dir"][:, :, b].astype(np.float64), tri_n[np.maximum(hits[:, :, b - 1].astype(np.int64), 0)]) > 0).astype(np.uint64)
    key = (key * np.uint64(0x9E3779B97F4A7C15) + (hits[:, :, b - 1].astype(np.int64).astype(np.uint64) * np.uint64(2) + side + np.uint64(1))) & np.uint64(0xFFFFFFFFFFFFFFFF)
    d_now = np.zeros((E, S), np.uint64); d_now[e_idx, s_idx] = refl.astype(np.uint64)
    dec = d_now if dec is None else (((dec << np.uint64(1)) | d_now) & np.uint64(7))
    dec_all = d_now if b == 1 else ((dec_all << np.uint64(1)) | d_now)
    if pure: order_idx = np.lexsort((s_idx, key[e_idx, s_idx], e_idx))
    elif segdec: order_idx = np.lexsort((s_idx, dec_all[e_idx, s_idx], e_idx))
    elif wgkey:
        pos = e_idx * S + s_idx; b256 = pos // 256
        k2 = prev.astype(np.int64) * 2 + refl.astype(np.int64)
        # group size within the block (largest first), then key, then sample
        comb = b256.astype(np.int64) * (1 << 40) + k2
        uq, inv, cnts = np.unique(comb, return_inverse=True, return_counts=True)
        order_idx = np.lexsort((s_idx, k2, -cnts[inv], b256))
    elif stray:
        pos = e_idx * S + s_idx; w64 = pos // 64
        first = np.concatenate([[True], w64[1:] != w64[:-1]])
        dom = prev[np.maximum.accumulate(np.where(first, np.arange(len(pos)), 0))]          # the triangle the wavefront's first live ray hit
        order_idx = np.lexsort((s_idx, ~refl, prev != dom, pos // 256))
    elif blk3: order_idx = np.lexsort((s_idx, dec[e_idx, s_idx], (e_idx * S + s_idx) // 256))
    else: order_idx = np.lexsort((s_idx, ~refl, (e_idx * S + s_idx) // 256))          # (round 5: reflected-first over the workgroup's 256 rays)
    q = np.ascontiguousarray(segs[e_idx, s_idx, b][order_idx]); own = hits[e_idx, s_idx, b][order_idx]
    n = len(q); n_pack = (n + W - 1) // W
    out = np.zeros((n_pack, 6), np.uint32); tri = np.zeros(n, np.int32)
    L.orc_packet_count(C.byref(osc.c), C.byref(p), q.ctypes.data, n, W, order, out.ctypes.data, tri.ctypes.data, os.cpu_count())
    assert np.array_equal(tri, own), "the packet walk found another triangle on bounce %d" % b
    full = out[:, 5] == W
    # packets whose rays all hit the SAME triangle last and took the same decision (what k_shade can see): their share and their cost
    k2s = (prev.astype(np.int64) * 2 + refl.astype(np.int64))[order_idx]
    padn = n_pack * W - n
    k2p = np.concatenate([k2s, np.full(padn, k2s[-1])]).reshape(n_pack, W)
    is_pure = (k2p.min(1) == k2p.max(1))
    r = {"bounce": b, "rays": int(n), "packets": int(n_pack),
         "packet_nodes_mean": float(out[:, 0].mean()), "packet_leaves_mean": float(out[:, 1].mean()), "packet_triangles_mean": float(out[:, 2].mean()),
         "solo_nodes_per_ray": float(out[:, 3].sum() / out[:, 5].sum()), "solo_nodes_max_in_packet_mean": float(out[:, 4].mean()),
         "packet_nodes_over_solo_mean_ray": float(out[:, 0].mean() / (out[:, 3].sum() / out[:, 5].sum())),
         "packet_nodes_over_longest_ray": float((out[:, 0] / np.maximum(out[:, 4], 1)).mean()),
         "packet_nodes_p50_p90_p99": [float(np.percentile(out[:, 0], x)) for x in (50, 90, 99)],
         "one_key_packets_share": float(is_pure.mean()), "one_key_packet_nodes_mean": float(out[is_pure, 0].mean()) if is_pure.any() else None,
         "one_key_packet_solo_nodes_per_ray": float(out[is_pure, 3].sum() / max(out[is_pure, 5].sum(), 1)) if is_pure.any() else None,
         "other_packets_solo_nodes_per_ray": float(out[~is_pure, 3].sum() / max(out[~is_pure, 5].sum(), 1)) if (~is_pure).any() else None}
    rows.append(r); sys.stderr.write(json.dumps(r) + "\n")
tot_pack = sum(r["packet_nodes_mean"] * r["packets"] for r in rows); tot_solo = sum(r["solo_nodes_per_ray"] * r["rays"] for r in rows)
print(json.dumps({"queue": "sorted into pure bundles (scan-line, history)" if pure else "every scan-line's rays partitioned by their reflect / refract decisions so far (no triangle ids)" if segdec else "256-ray blocks sorted by the last three decisions" if blk3 else "as k_shade's compaction leaves it (reflected first inside a 256-ray block)", "workload": workload, "scan_lines": E, "rays": S, "packet": W, "order": "first hitting ray" if order else "smallest t_near of the packet", "seconds": round(time.time() - t0, 1),
                  "per_bounce": rows,
                  "bounces_ge1": {"packet_node_visits_per_ray": tot_pack / sum(r["rays"] for r in rows), "solo_node_visits_per_ray": tot_solo / sum(r["rays"] for r in rows),
                                  "wave_level_node_steps_packet_over_lane_walk_at_full_lanes": tot_pack * 64 / tot_solo}}, indent=1))
