#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares and the per-bounce launch timeline of k_trace from a -DMCRT_STAMP build
(MCRT_LIB=.../libmcrt_hip_stamp.so).  usage: stamps.py [rays] [frames_in_flight]"""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mcray_tracing_amd as m
import torch
rays = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
F = int(sys.argv[2]) if len(sys.argv) > 2 else 16
cfg, meshes = m.synth.random_scene(1_000_000, 8, 12345)
sd = m.scene_io.build_scene(cfg, meshes)
tr = m.Transducer(128, position=cfg["transducerPosition"], angles_deg=cfg["transducerAngles"])
sim = m.Simulator(sd, tr, n_samples=rays)
ctx = sim.ctx
rf = torch.empty((F, 128, sim.R), dtype=torch.float32, device="cuda")
out = (C.c_uint64 * 200)()
hist = (C.c_uint64 * 2560)()
for f in range(3):
    ctx.trace_frames(f * F, F, rf, 0, 128)
ctx.synchronize()
ctx.L.mcrt_debug_stamps(ctx.h, out, 1)
ctx.L.mcrt_debug_tail_histograms(ctx.h, hist, 1)
ctx.trace_frames(100, F, rf, 0, 128)          # ONE pass: the timeline slots hold min/max over what ran since the reset
ctx.synchronize()
ctx.L.mcrt_debug_stamps(ctx.h, out, 1)
ctx.L.mcrt_debug_tail_histograms(ctx.h, hist, 1)
v = [int(x) for x in out]
if v[8]:      # (a -DMCRT_STAMP_LITE build carries only the timeline)
    names = ["refill cyc", "phase1 cyc", "phase2 cyc", "phase1 iters", "phase2 iters", "outer iters", "active lanes p1 (sum)", "active lanes p2 (sum)", "waves"]
    for n, x in zip(names, v): print("%-24s %16d" % (n, x))
    tot = v[0] + v[1] + v[2] + v[130] + v[131]
    if v[132]:
        print("refill rounds %d (%.1f per wave, %.1f lanes each): report + claim %.0f cycles, ray load + set-up %.0f cycles, hand-over + rest %.0f cycles per round" % (
            v[132], v[132] / v[8], v[133] / v[132], v[130] / v[132], v[131] / v[132], v[0] / v[132]))
    print("shares: refill %.1f%%  phase1 %.1f%%  phase2 %.1f%%" % (100 * (v[0] + v[130] + v[131]) / tot, 100 * v[1] / tot, 100 * v[2] / tot))
    print("cycles per phase-1 iteration %.0f (avg active lanes %.1f/64); per phase-2 iteration %.0f (avg parked lanes %.1f/64)" % (v[1] / max(v[3], 1), v[6] / max(v[3], 1), v[2] / max(v[4], 1), v[7] / max(v[4], 1)))
    print("per wave: %.0f cycles, %.1f node iterations, %.1f leaf iterations" % (tot / v[8], v[3] / v[8], v[4] / v[8]))
    print("k_march: %d waves, loop iterations %.1f per wave; step iterations %.1f per wave with %.2f of 16 quads active; iterations with a finishing quad %.1f; refill rounds %.1f"
          % (v[14], v[9] / max(v[14], 1), v[10] / max(v[14], 1), v[11] / max(v[10], 1), v[12] / max(v[14], 1), v[13] / max(v[14], 1)))
    if v[124]: print("k_march cycle shares: hand-out and finished segments %.1f%%, advance %.1f%%, voxel + gathers %.1f%%, rows and bins %.1f%% (of the loop's %.0f cycles per wavefront)" % tuple([100.0 * v[120 + k] / v[124] for k in range(4)] + [v[124] / max(v[14], 1)]))
    print("lane walk, per node-step iteration: %.1f lanes stepping on %.1f distinct nodes, %.1f parked on a leaf, %.1f without a walk; subtrees adopted %d" % (v[6] / max(v[3], 1), v[59] / max(v[3], 1), v[56] / max(v[3], 1), v[57] / max(v[3], 1), v[58]))
M = (1 << 64) - 1
lane = True
print("bounce   launch us   queue empty at us (share of launch)   wavefronts   mean wave lifetime us   mean time after the queue ran dry us | mean start us  longest life us | node-step iterations per wave: mean, most")
for b in range(10):
    s0, s1, e, life = v[16 + 4 * b: 20 + 4 * b]
    if e == 0: continue
    start, empty = M - s0, (M - s1) if s1 else None
    dur = (e - start) / 100.0
    em = (empty - start) / 100.0 if empty is not None else float("nan")
    waves = v[61 + 2 * b] if lane else 5120
    print("%4d   %10.1f   %10.1f (%.0f%%)   %8d   %10.1f   %10.1f" % (b, dur, em, 100 * em / dur if dur else 0, waves, life / 100.0 / max(waves, 1), (v[60 + 2 * b] / 100.0 / max(waves, 1)) if lane else float("nan")), end="")
    if lane and waves: print("   | %8.1f %10.1f | %8.1f %8d" % ((v[80 + b] / waves - start) / 100.0, v[90 + b] / 100.0, v[100 + b] / waves, v[110 + b]))
    else: print()
H = np.array([int(x) for x in hist], np.int64).reshape(10, 256)
if H.sum():
    print("\ntail of each walk launch (wavefronts by 20 us bins on their own clock): when they END / when they FIND THE QUEUE DRY; lanes still walking at that moment")
    for b in range(10):
        end, dry, infl, since = H[b, :64], H[b, 64:128], H[b, 128:192], H[b, 192:256]
        if not end.sum(): continue
        def pct(h, q):
            c = np.cumsum(h); return 20.0 * (int(np.searchsorted(c, q * c[-1])) + 1) if c[-1] else float("nan")
        print("bounce %d: end p10 / p50 / p90 / p99 / last = %4.0f / %4.0f / %4.0f / %4.0f / %4.0f us   dry p10 / p50 / p90 / p99 / last = %4.0f / %4.0f / %4.0f / %4.0f / %4.0f us   node-step iterations last claim -> dry p50 / p90 / p99 / last = %3.0f / %3.0f / %3.0f / %3.0f   last claim -> dry p50 / p90 / p99 / last = %4.0f / %4.0f / %4.0f / %4.0f us"
              % (b, pct(end, .1), pct(end, .5), pct(end, .9), pct(end, .99), 20.0 * (np.nonzero(end)[0].max() + 1),
                 pct(dry, .1), pct(dry, .5), pct(dry, .9), pct(dry, .99), 20.0 * (np.nonzero(dry)[0].max() + 1) if dry.sum() else float("nan"),
                 pct(infl, .5) * 0.4, pct(infl, .9) * 0.4, pct(infl, .99) * 0.4, 8.0 * (np.nonzero(infl)[0].max() + 1) if infl.sum() else float("nan"), pct(since, .5), pct(since, .9), pct(since, .99), 20.0 * (np.nonzero(since)[0].max() + 1) if since.sum() else float("nan")))
    if os.environ.get("STAMPS_DUMP"):
        for b in (1, 5, 9):
            print("bounce %d end  :" % b, " ".join(str(int(x)) for x in H[b, :64]))
            print("bounce %d dry  :" % b, " ".join(str(int(x)) for x in H[b, 64:128]))
if sum(v[160:176]):
    print("\nthe LAST wavefront of a workgroup outlives the second-last by (20 us bins, all bounces):", " ".join(str(x) for x in v[160:176]))
    print("... of bounce 1's workgroups that end after 940 us:", " ".join(str(x) for x in v[140:156]))
if lane: sys.exit(0)
h = v[60:77]
tot_h = sum(h) or 1
print("node visits per walk (bounces >= 1), log2 bins: share of walks / share of visits (bin midpoint estimate)")
est = [h[k] * (1.5 * 2 ** (k - 1) if k > 0 else 0) for k in range(len(h))]
te = sum(est) or 1
for k, x in enumerate(h):
    if x: print("  < %6d : %6.2f%%  %6.2f%%" % (2 ** k, 100 * x / tot_h, 100 * est[k] / te))
