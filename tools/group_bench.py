#!/usr/bin/env python3
"""mcrt_group_* on the box: the headline workload traced through a GROUP (the C-ABI's several-GPUs path) against a single context.
On the one-GPU box the ranks share GPU 0, so this measures what the group's plumbing costs -- the ranks' host threads, the per-rank
block buffers, the peer copies and the interleaving kernel, the double-buffered hand-over to the root -- not xGMI; on a node with
several GPUs pass their ids.

    python tools/group_bench.py [devices=0,0] [frames_in_flight=32] [passes=8]      -> JSON on stdout
"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mcray_tracing_amd as m

devices = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0,0").split(",")]
F = int(sys.argv[2]) if len(sys.argv) > 2 else 32
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 8
per_rank_lines, S = 128, 1024
cfg, meshes = m.synth.random_scene(1_000_000, 8, 12345)
sd = m.scene_io.build_scene(cfg, meshes)
psf = m.Psf()


def run(devs, E):
    tr = m.Transducer(E, position=cfg["transducerPosition"], angles_deg=cfg["transducerAngles"])
    grp = m.Group(devs)
    grp.set_params(n_elements=E, n_samples=S, frequency=tr.frequency)
    grp.upload_scene(sd); grp.upload_texture(None, 256); grp.set_transducer(tr.pos, tr.dir)
    R = grp.root.params.n_rows
    bufs = [grp.root.alloc(F * E * R * 4) for _ in range(2)]
    imgs = [grp.root.alloc(F * 400 * 500 * 4) for _ in range(2)]

    def region(first):
        for k in range(passes):
            i = k & 1
            grp.trace_frames(first + k * F, F, bufs[i])
            grp.root.convolve_frames(bufs[i], F, E, R, psf.axial_kernel, psf.lateral_kernel)      # on the ROOT's stream: overlaps the ranks' next pass
            grp.root.envelope_frames(bufs[i], F, E, R)
            grp.root.scan_convert_frames(bufs[i], F, E, R, imgs[i])
        grp.synchronize()
    region(1000)
    t = []
    for rep in range(5):
        t0 = time.perf_counter(); region(0); t.append(time.perf_counter() - t0)
    dt = sorted(t)[len(t) // 2]
    tr_ms, cp_ms = grp.last_pass_ms()
    out = {"devices": devs, "scan_lines": E, "rays_per_scan_line": S, "frames_in_flight": F, "passes": passes,
           "ms_per_frame": dt / (passes * F) * 1e3, "rays_per_s": E * S * passes * F / dt,
           "last_pass_trace_ms_per_rank": [round(float(x), 3) for x in tr_ms], "last_pass_copy_ms_per_rank": [round(float(x), 3) for x in cp_ms]}
    for d in bufs + imgs:
        grp.root.free(d)
    grp.close()
    return out


res = {"what": "whole B-mode frames through mcrt_group_trace_frames + post-processing on the root context, double-buffered; weak scaling (%d scan-lines per rank)" % per_rank_lines,
       "one_rank": run(devices[:1], per_rank_lines), "group": run(devices, per_rank_lines * len(devices))}
if len(set(devices)) == 1 and len(devices) > 1:
    res["same_work_one_rank"] = run(devices[:1], per_rank_lines * len(devices))       # the same frame traced by ONE context: what sharing a GPU among ranks costs
print(json.dumps(res, indent=1))
