#!/usr/bin/env python3
"""Experiment: two contexts on one GPU, passes alternating between them on two streams (does the head of one pass fill the
drain of the other?).  usage: two_ctx.py [frames_in_flight] [contexts]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mcray_tracing_amd as m
import torch
F = int(sys.argv[1]) if len(sys.argv) > 1 else 16
NC = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfg, meshes = m.synth.random_scene(1_000_000, 8, 12345)
sd = m.scene_io.build_scene(cfg, meshes)
E, S = 128, 1024
tr = m.Transducer(E, position=cfg["transducerPosition"], angles_deg=cfg["transducerAngles"])
ctxs, streams, rfs = [], [], []
for i in range(NC):
    c = m.Context(0); c.set_params(n_elements=E, n_samples=S, frequency=tr.frequency)
    c.upload_scene(sd); c.upload_texture(None, 256); c.set_transducer(tr.pos, tr.dir)
    st = torch.cuda.Stream(); c.set_stream(st.cuda_stream)
    ctxs.append(c); streams.append(st); rfs.append(torch.empty((F, E, c.params.n_rows), dtype=torch.float32, device="cuda"))
def run(passes, first):
    for p in range(passes):
        k = p % NC
        ctxs[k].trace_frames(first + p * F, F, rfs[k])
run(2 * NC, 0); torch.cuda.synchronize()
P = 8
t0 = time.perf_counter(); run(P, 1000); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("contexts %d, %d frames in flight each: %.3f ms/frame  %.1f M rays/s" % (NC, F, 1e3 * dt / (P * F), E * S * P * F / dt / 1e6))
