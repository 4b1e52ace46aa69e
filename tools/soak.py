#!/usr/bin/env python3
"""Soak: the same passes traced over and over (1 M-triangle scene, 128 x 1024 rays per frame), every repeat compared bit for bit with the
first -- races in the walk's hand-overs, the closest-hit words or the accumulation would show as a differing image or a watchdog error.
usage: soak.py [seconds per pass size]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mcray_tracing_amd as m
import torch
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
cfg, meshes = m.synth.random_scene(1_000_000, 8, 12345)
sd = m.scene_io.build_scene(cfg, meshes)
E, S = 128, 1024
tr = m.Transducer(E, position=cfg["transducerPosition"], angles_deg=cfg["transducerAngles"])
sim = m.Simulator(sd, tr, n_samples=S)
ctx = sim.ctx
bad = 0
for F in (1, 3, 20, 128):
    rf = torch.empty((F, E, sim.R), dtype=torch.float32, device="cuda")
    ctx.trace_frames(7, F, rf); ctx.synchronize()
    want = rf.view(torch.int32).clone()
    t0, n = time.time(), 0
    while time.time() - t0 < budget:
        for _ in range(max(1, 64 // F)):
            ctx.trace_frames(7, F, rf)
        ctx.synchronize()                      # (raises on a watchdog / stack error)
        n += max(1, 64 // F)
        if not torch.equal(rf.view(torch.int32), want):
            bad += 1
            print("MISMATCH at pass size %d after %d passes: %d words differ" % (F, n, int((rf.view(torch.int32) != want).sum())))
    print("pass size %3d: %6d passes (%d frames) in %.0f s, all identical: %s" % (F, n, n * F, time.time() - t0, bad == 0))
sim.close()
sys.exit(1 if bad else 0)
