#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of k_trace from a -DMCRT_STAMP build (MCRT_LIB=.../libmcrt_hip_stamp.so)."""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mcray_tracing_amd as m
rays = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cfg, meshes = m.synth.random_scene(1_000_000, 8, 12345)
sd = m.scene_io.build_scene(cfg, meshes)
tr = m.Transducer(128, position=cfg["transducerPosition"], angles_deg=cfg["transducerAngles"])
sim = m.Simulator(sd, tr, n_samples=rays)
out = (C.c_uint64 * 16)()
for f in range(3):
    sim.trace(f)
sim.ctx.synchronize()
sim.ctx.L.mcrt_debug_stamps(sim.ctx.h, out, 1)
for f in range(5):
    sim.trace(f)
sim.ctx.synchronize()
sim.ctx.L.mcrt_debug_stamps(sim.ctx.h, out, 1)
v = [int(x) for x in out]
names = ["refill cyc", "phase1 cyc", "phase2 cyc", "phase1 iters", "phase2 iters", "outer iters", "active lanes p1 (sum)", "active lanes p2 (sum)", "waves"]
for n, x in zip(names, v): print("%-24s %16d" % (n, x))
tot = v[0] + v[1] + v[2]
print("shares: refill %.1f%%  phase1 %.1f%%  phase2 %.1f%%" % (100 * v[0] / tot, 100 * v[1] / tot, 100 * v[2] / tot))
print("cycles per phase-1 iteration %.0f (avg active lanes %.1f/64); per phase-2 iteration %.0f (avg parked lanes %.1f/64)" % (v[1] / max(v[3], 1), v[6] / max(v[3], 1), v[2] / max(v[4], 1), v[7] / max(v[4], 1)))
print("per wave: %.0f cycles, %.1f node iterations, %.1f leaf iterations" % (tot / v[8], v[3] / v[8], v[4] / v[8]))
