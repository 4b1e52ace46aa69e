"""Diagnostic (round 6): k_path's per-bounce cycle stamps from a -DMCRT_PATH_STAMP build (MCRT_LIB=.../libmcrt_hip_<variant>.so): walk and shade cycles per
wavefront and bounce, outer-loop iterations, lanes alive -- and the time of one frame at a time."""
import os, sys, ctypes as C
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/bench.py') else os.getcwd())
import mcray_tracing_amd as m, torch
cfg, meshes = m.synth.random_scene(1_000_000, 8, 12345)
sd = m.scene_io.build_scene(cfg, meshes)
tr = m.Transducer(128, position=cfg["transducerPosition"], angles_deg=cfg["transducerAngles"])
sim = m.Simulator(sd, tr, n_samples=1024); ctx = sim.ctx
rf = torch.empty((1, 128, sim.R), dtype=torch.float32, device="cuda")
out = (C.c_uint64 * 200)()
for f in range(3): ctx.trace_frames(f, 1, rf, 0, 128)
ctx.synchronize(); ctx.L.mcrt_debug_stamps(ctx.h, out, 1)
N = 8
for f in range(N): ctx.trace_frames(100 + f, 1, rf, 0, 128)
ctx.synchronize(); ctx.L.mcrt_debug_stamps(ctx.h, out, 1)
v = [int(x) for x in out]
print("bounce  waves  alive/wave  walk cyc/wave  max walk cyc  shade cyc/wave  iters/wave  max iters  cyc/iter")
for b in range(10):
    w = max(v[48 + b], 1)
    print("%4d  %7.0f  %8.1f  %12.0f  %12d  %12.0f  %10.1f  %8d  %8.0f" % (b, v[48 + b] / N, v[80 + b] / w, v[b] / w, v[64 + b], v[16 + b] / w, v[32 + b] / w, v[96 + b], v[b] / max(v[32 + b], 1)))
import time
ctx.synchronize(); t0 = time.perf_counter()
for f in range(200): ctx.trace_frames(1000 + f, 1, rf, 0, 128)
ctx.synchronize(); print("one frame at a time: %.4f ms per frame (trace + accumulate, no post-processing; stamped build)" % ((time.perf_counter() - t0) / 200 * 1e3))
print("sum of mean walk %.0f cyc, of mean shade %.0f cyc; sum of max walk %d" % (sum(v[b] / max(v[48 + b], 1) for b in range(10)), sum(v[16 + b] / max(v[48 + b], 1) for b in range(10)), sum(v[64:74])))
