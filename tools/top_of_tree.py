#!/usr/bin/env python3
"""How much of the walk's node traffic a STATIC table of the tree's hottest nodes would serve (CPU only): the closest-hit queries of whole
frames of the headline workload walked over the product's BVH4 (the oracle's analysis walk, as tools/bvh_width.py), visits counted per node.
Reports the share of inner-node visits that go to the N most visited nodes (the best any static table of N nodes can do).

    python tools/top_of_tree.py [frames=1] [rays=1024] [workload=random1m]       -> JSON on stdout (profiles/round4/top_of_tree.json)
"""
import ctypes as C, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mcray_tracing_amd as m
from oracle import orc

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
workload = sys.argv[3] if len(sys.argv) > 3 else "random1m"
E = 128
if workload == "random1m":
    cfg, meshes = m.synth.random_scene(1_000_000, 8, 12345)
elif workload == "liver":
    cfg, meshes = m.synth.liver_scene(5)
else:
    cfg, meshes = m.synth.sphere_scene(5)
sd = m.scene_io.build_scene(cfg, meshes)
tr = m.Transducer(E, position=cfg["transducerPosition"], angles_deg=cfg["transducerAngles"])
nodes, btri, depth = m.host_build_bvh(sd.tri, sd.tri_mesh)
osc = orc.OracleScene(sd.tri, sd.tri_mesh, sd.meshes, sd.materials, sd.start_mat, sd.spacing, bvh=(nodes, btri))
tex = orc.texture(256)
p = orc.default_params(n_elements=E, n_samples=S)
L = orc.lib()
L.orc_wide_build.restype = C.c_void_p; L.orc_wide_build.argtypes = [C.c_void_p, C.c_uint32, C.c_int]
L.orc_wide_free.argtypes = [C.c_void_p]; L.orc_wide_nodes.argtypes = [C.c_void_p]; L.orc_wide_nodes.restype = C.c_uint32
L.orc_wide_visits.argtypes = [C.c_void_p, C.c_void_p]
L.orc_wide_count.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int]
segs_all, tri_all = [], []
for f in range(frames):
    o = osc.trace_frame(p, tr.pos, tr.dir, tex, frame_id=f, use_bvh=1, n_threads=os.cpu_count(), want_segs=True, want_ref=False, want_fix=False)
    cnt = o["seg_count"]; B = o["segs"].shape[2]
    live = np.arange(B)[None, None, :] < cnt[:, :, None]
    segs_all.append(o["segs"][live]); tri_all.append(o["hits"][live])
segs = np.ascontiguousarray(np.concatenate(segs_all)); want = np.concatenate(tri_all)
n = segs.shape[0]
w = L.orc_wide_build(C.byref(osc.c), 4, 0)
nn = int(L.orc_wide_nodes(w))
visits = np.zeros(nn, np.uint32)
L.orc_wide_visits(w, visits.ctypes.data)
out = np.zeros((n, 4), np.uint32); tri = np.zeros(n, np.int32)
L.orc_wide_count(w, C.byref(osc.c), C.byref(p), segs.ctypes.data, n, out.ctypes.data, tri.ctypes.data, os.cpu_count())
assert np.array_equal(tri, want)
total = int(visits.sum()); assert total == int(out[:, 0].sum())
hot = np.sort(visits)[::-1].astype(np.float64).cumsum() / total
res = {"workload": workload, "frames": frames, "queries": int(n), "nodes_in_tree": nn, "inner_node_visits_per_query": total / n,
       "share_of_visits_to_the_N_most_visited_nodes": {str(N): float(hot[min(N, nn) - 1]) for N in (16, 64, 128, 256, 512, 1024, 2048, 4096, 16384)}}
L.orc_wide_order.argtypes = [C.c_void_p, C.c_int, C.c_void_p]; L.orc_wide_order.restype = C.c_uint32
for mode, name in ((0, "first_N_nodes_breadth_first"), (1, "first_N_nodes_largest_box_first")):
    order = np.zeros(nn, np.uint32)
    assert L.orc_wide_order(w, mode, order.ctypes.data) == nn
    cum = visits[order].astype(np.float64).cumsum() / total
    res["share_of_visits_to_the_" + name] = {str(N): float(cum[min(N, nn) - 1]) for N in (16, 64, 128, 192, 256, 512, 1024, 2048, 4096, 16384)}
print(json.dumps(res, indent=1))
L.orc_wide_free(w)
