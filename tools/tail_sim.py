#!/usr/bin/env python3
"""Would handing the heavy rays out FIRST cut the lane walk's launch tail?  A wave-level model of the persistent walk, on the CPU.

Counted per-ray costs (the oracle's counting walk: nodes + 2 x triangles) of the headline workload's bounces 2-4 run through a model of k_trace_lane: wavefronts of 64
lanes in lockstep, a lane steps one node per iteration, idle lanes take the next rays of the queue when 16 of a wavefront's lanes are idle, idle lanes share
what is left of their wavefront once the queue is dry.  Makespan (and the iteration the queue runs dry at) for the queue in scan-line order (the product), with
the scan-lines / the (scan-line, history) bundles ranked heaviest first from ANOTHER frame's counts (what a measurement of the previous pass could give), and
with every ray's own cost known (LPT: what no predictor can beat).  Result (profiles/round5/exp_line_order.txt): ranking lines or bundles per bounce gives
4-6 %, one ranking for all bounces nothing (measured on the GPU: nothing, tools/variants/round5_line_order.patch), LPT per ray 9-15 % -- the long rays are
spread over most lines, and a ray's own length is not known before it is walked.
    python tools/tail_sim.py"""
import os, sys, numpy as np, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mcray_tracing_amd as m
from oracle import orc
E, S = 128, 256
cfg, meshes = m.synth.random_scene(1_000_000, 8, 12345)
sd = m.scene_io.build_scene(cfg, meshes)
tr = m.Transducer(E, position=cfg["transducerPosition"], angles_deg=cfg["transducerAngles"])
nodes, btri, n4, _ = m.host_build_bvh4(sd.tri, sd.tri_mesh)
osc = orc.OracleScene(sd.tri, sd.tri_mesh, sd.meshes, sd.materials, sd.start_mat, sd.spacing, bvh=(nodes, btri)); osc.set_bvh4(n4)
tex = orc.texture(256); p = orc.default_params(n_elements=E, n_samples=S)
L = orc.lib(); L.orc_seed_count.restype = None
L.orc_seed_count.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
def frame(f):
    o = osc.trace_frame(p, tr.pos, tr.dir, tex, frame_id=f, use_bvh=2, n_threads=8, want_segs=True, want_ref=False, want_fix=False)
    segs, cnt, hits = o["segs"], o["seg_count"], o["hits"]; per = {}
    tri_n = np.cross(sd.tri[:, 3:6] - sd.tri[:, 0:3], sd.tri[:, 6:9] - sd.tri[:, 0:3]).astype(np.float64)
    hist = np.zeros((E, S), np.int64)
    for b in range(1, 6):
        prev = np.maximum(hits[:, :, b - 1].astype(np.int64), 0)
        din = np.einsum("esk,esk->es", segs["dir"][:, :, b - 1].astype(np.float64), tri_n[prev]); dout = np.einsum("esk,esk->es", segs["dir"][:, :, b].astype(np.float64), tri_n[prev])
        hist = hist * 2 + ((din > 0) != (dout > 0)).astype(np.int64)          # 1 = reflected at the end of bounce b-1
        es = np.argwhere(cnt > b); q = np.ascontiguousarray(segs[es[:, 0], es[:, 1], b]); n = len(q)
        out = np.zeros((n, 2), np.uint32); tri = np.zeros(n, np.int32)
        L.orc_seed_count(C.byref(osc.c), C.byref(p), q.ctypes.data, n, 0, None, out.ctypes.data, tri.ctypes.data, 8)
        per[b] = (es[:, 0].copy(), es[:, 1].copy(), out[:, 0].astype(np.int64) + 2 * out[:, 1].astype(np.int64), hist[es[:, 0], es[:, 1]].copy())   # cost: nodes + 2 x triangles
    return per
A, Bf = frame(0), frame(1)
def simulate(cost, lanes=4096, refill=16):
    """wave-level model of the persistent lane walk: waves of 64 lanes in lockstep, a lane steps one node per iteration, idle lanes take the next
    rays of the queue when >= refill of the wave's lanes are idle (or the wave is empty)"""
    W = lanes // 64; n = len(cost); nxt = 0
    rem = np.zeros((W, 64), np.int64); it = 0; dry_at = None; ends = np.zeros(W, np.int64); done = np.zeros(W, bool)
    while True:
        idle = rem <= 0
        ni = idle.sum(1)
        for w in np.nonzero((ni >= refill) & ~done)[0]:
            if nxt < n:
                k = min(int(ni[w]), n - nxt); idx = np.nonzero(idle[w])[0][:k]
                rem[w, idx] = cost[nxt:nxt + k]; nxt += k
                if nxt >= n and dry_at is None: dry_at = it
            elif ni[w] == 64:
                done[w] = True; ends[w] = it
            else:
                # queue dry: subtree adoption -- idle lanes share the remaining work of the wave's busy lanes (perfectly)
                tot = rem[w][rem[w] > 0].sum(); rem[w][:] = 0; rem[w][:1] = 0
                share = int(np.ceil(tot / 64)); rem[w][:] = share
        if done.all(): break
        rem -= 1; it += 1
    return it, dry_at
for b in (2, 3, 4):
    lineA, sA, cA, hA = A[b]; line, smp, cost, hB = Bf[b]
    lmaxA = np.array([cA[lineA == e].max() if (lineA == e).any() else 0 for e in range(E)])
    def queue(order_lines):
        rank = np.empty(E, np.int64); rank[order_lines] = np.arange(E)
        o = np.lexsort((smp, rank[line])); return cost[o]
    def bundle_queue(nc):
        # (line, history) bundles ranked by the OTHER frame's max cost; unseen bundles take their line's max; nc classes (0: fully sorted), scan order inside a class
        keyA = lineA * 1024 + hA; keyB = line * 1024 + hB
        tab = {}
        for k, c in zip(keyA, cA): tab[k] = max(tab.get(k, 0), c)
        pred = np.array([tab.get(k, lmaxA[k // 1024]) for k in keyB])
        if nc == 0:
            o = np.lexsort((smp, line, -pred))
        else:
            r = np.argsort(np.argsort(-pred, kind="stable"), kind="stable") * nc // len(pred)
            o = np.lexsort((smp, line, r))
        return cost[o]
    ident = np.arange(E)
    allmax = np.zeros(E)
    for bb in A:
        la, _, ca, _h = A[bb]
        for e in range(E):
            v = ca[la == e]
            if len(v): allmax[e] = max(allmax[e], v.max())
    heavy_first = np.argsort(-lmaxA); heavy_one = np.argsort(-allmax)
    lmeanA = np.array([cA[lineA == e].mean() if (lineA == e).any() else 0 for e in range(E)])
    res = {"scan order": simulate(queue(ident)), "heavy lines first (other frame)": simulate(queue(heavy_first)), "LPT per ray": simulate(np.sort(cost)[::-1]), "one ranking (max over bounces)": simulate(queue(heavy_one)), "by line mean": simulate(queue(np.argsort(-lmeanA))),
           "8 classes of lines by max, scan order inside": simulate(queue(np.array(sorted(range(E), key=lambda e: (np.argsort(np.argsort(-lmaxA))[e] * 8 // E, e))))),
           "bundle classes": simulate(bundle_queue(8)), "bundle classes 16": simulate(bundle_queue(16)), "bundle LPT (no classes)": simulate(bundle_queue(0)),
           "random ray order": simulate(np.random.default_rng(1).permutation(cost))}
    print("bounce", b, "rays", len(cost), "ideal (sum/lanes)", int(cost.sum() / 4096), {k: v for k, v in res.items()})
