#!/usr/bin/env python3
"""Could the walk's launch tail be cut by handing the heavy scan-lines out FIRST?  Counted on the CPU.

The lane walk's launch ends one longest-ray after its queue runs dry: with ~8 rays per lane in a 20-frame launch, whatever is
claimed in the last eighth of the queue is still being walked then.  If the walk length of a ray were predictable from its scan-line (the
geometry a line crosses is the same in every frame of a pass), the lines could be queued heaviest first and the tail would be made of the
lightest lines' rays.  Per bounce, frame A ranks the lines (by the p99 of their rays' node visits), frame B (other random numbers) is the test:
    nodes_max / p999 / p99 of   all rays   |   the rays of the lightest eighth of the lines as frame A ranked them   |   as frame B itself would
    python tools/tail_predict.py [workload=random1m|liver|sphere] [rays=256]      -> JSON
"""
import ctypes as C, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mcray_tracing_amd as m
from oracle import orc

workload = sys.argv[1] if len(sys.argv) > 1 else "random1m"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
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
B = p.max_depth
t0 = time.time()

def frame_nodes(f):
    o = osc.trace_frame(p, tr.pos, tr.dir, tex, frame_id=f, use_bvh=2, n_threads=os.cpu_count(), want_segs=True, want_ref=False, want_fix=False)
    segs, cnt = o["segs"], o["seg_count"]
    per = []
    for b in range(B):
        live = cnt > b
        es = np.argwhere(live)
        if len(es) == 0:
            per.append(None); continue
        q = np.ascontiguousarray(segs[es[:, 0], es[:, 1], b]); n = len(q)
        out = np.zeros((n, 2), np.uint32); tri = np.zeros(n, np.int32)
        L.orc_seed_count(C.byref(osc.c), C.byref(p), q.ctypes.data, n, 0, None, out.ctypes.data, tri.ctypes.data, os.cpu_count())
        per.append((es[:, 0].copy(), out[:, 0].astype(np.int64)))
    return per

A, Bf = frame_nodes(0), frame_nodes(1)

def q3(x):
    return {"max": int(x.max()), "p999": float(np.percentile(x, 99.9)), "p99": float(np.percentile(x, 99)), "mean": float(x.mean()), "rays": int(len(x))}

# ONE ranking for every bounce (what a permutation of the scan-lines in k_init can give): frame A's per-line max over bounces >= 1
def line_stat_all(fr_all, fn):
    out = np.zeros(E)
    for b in range(1, B):
        if fr_all[b] is None: continue
        line, nd = fr_all[b]
        for e in range(E):
            v = nd[line == e]
            if len(v): out[e] = max(out[e], fn(v))
    return out
rank_one = np.argsort(line_stat_all(A, np.max))
if os.environ.get("ORDER_OUT"):
    # heaviest first, dealt round-robin to the 8 XCD sub-queues (contiguous eighths of the queue): slot = x * per + j holds the line of rank j * 8 + x
    heavy_first = rank_one[::-1]
    order = [int(heavy_first[j * 8 + x]) for x in range(8) for j in range(E // 8)]
    open(os.environ["ORDER_OUT"], "w").write(" ".join(map(str, order)) + "\n")
    open(os.environ["ORDER_OUT"] + ".plain", "w").write(" ".join(str(int(v)) for v in heavy_first) + "\n")
rows = []
for b in range(1, B):
    if A[b] is None or Bf[b] is None or len(Bf[b][1]) < 2000:
        continue
    def line_stat(fr, fn):
        line, nd = fr
        return np.array([fn(nd[line == e]) if (line == e).any() else 0.0 for e in range(E)])
    stat = {"p99": lambda v: np.percentile(v, 99), "max": np.max, "mean": np.mean}[os.environ.get("RANK_BY", "p99")]
    rankA = np.argsort(line_stat(A[b], stat))          # lightest first
    rankB = np.argsort(line_stat(Bf[b], lambda v: np.percentile(v, 99)))
    line, nd = Bf[b]
    def lightest(rank, share):
        # the lightest lines holding `share` of frame B's rays
        cnt = np.bincount(line, minlength=E)[rank]; k = int(np.searchsorted(np.cumsum(cnt), share * len(nd))) + 1
        return nd[np.isin(line, rank[:k])], k
    la, ka = lightest(rankA, 0.125); lb, kb = lightest(rankB, 0.125); lo, ko = lightest(rank_one, 0.125)
    corr = float(np.corrcoef(line_stat(A[b], lambda v: np.percentile(v, 99)), line_stat(Bf[b], lambda v: np.percentile(v, 99)))[0, 1])
    rows.append({"bounce": b, "all": q3(nd), "lightest_eighth_ranked_by_other_frame": dict(q3(la), lines=ka), "lightest_eighth_ranked_by_itself": dict(q3(lb), lines=kb), "lightest_eighth_one_ranking_for_all_bounces": dict(q3(lo), lines=ko),
                 "line_p99_correlation_between_frames": corr})
print(json.dumps({"workload": workload, "scan_lines": E, "rays": S, "seconds": round(time.time() - t0, 1), "per_bounce": rows}, indent=1))
