#!/usr/bin/env python3
"""Build time and walk cost of the two BVH builders on the 1 M-triangle workload."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mcray_tracing_amd as m
import torch

n_tri = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
cfg, meshes = m.synth.random_scene(n_tri, 8, 12345)
sd = m.scene_io.build_scene(cfg, meshes)
E, S, F = 128, 1024, 16
tr = m.Transducer(E, position=cfg["transducerPosition"], angles_deg=cfg["transducerAngles"])
for builder in ("sah", "lbvh"):
    t0 = time.perf_counter()
    sim = m.Simulator(sd, tr, n_samples=S, bvh_builder=builder)
    sim.ctx.synchronize()
    t_create = time.perf_counter() - t0
    t0 = time.perf_counter()
    sim.ctx.upload_scene(sd); sim.ctx.synchronize()
    t_upload = time.perf_counter() - t0
    d_tri = torch.from_numpy(sd.tri).cuda()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        sim.ctx.update_triangles(d_tri); sim.ctx.synchronize()
        ts.append(time.perf_counter() - t0)
    tr_ = []
    for _ in range(5):
        t0 = time.perf_counter()
        sim.ctx.refit_triangles(d_tri); sim.ctx.synchronize()
        tr_.append(time.perf_counter() - t0)
    n4, ms = sim.ctx.L and (sim.ctx.get_bvh4()[0].shape[0], sim.ctx.get_bvh4()[1])
    rf = torch.empty((F, E, sim.R), dtype=torch.float32, device="cuda")
    for f in range(2): sim.ctx.trace_frames(f * F, F, rf)
    sim.ctx.enable_stats(True); sim.ctx.get_stats(reset=True)
    sim.ctx.trace_frames(0, F, rf); st = sim.ctx.get_stats(reset=True); sim.ctx.enable_stats(False)
    sim.ctx.synchronize()
    t0 = time.perf_counter()
    for f in range(4): sim.ctx.trace_frames(100 + f * F, F, rf)
    sim.ctx.synchronize()
    dt = (time.perf_counter() - t0) / (4 * F)
    print("%-5s upload_scene %.3f s   update_triangles (device pointer) min %.2f ms   refit min %.2f ms   BVH4 nodes %d  max_stack %d   nodes/query %.1f  tris/query %.1f   %.3f ms/frame  %.1f M rays/s"
          % (builder, t_upload, 1e3 * min(ts), 1e3 * min(tr_), n4, ms, st["nodes_visited"] / st["queries"], st["tris_tested"] / st["queries"], 1e3 * dt, E * S / dt / 1e6))
    sim.close()
