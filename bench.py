#!/usr/bin/env python3
"""bench.py -- rays/sec + B-mode frames/sec of the ray-tracing hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload random1m|sphere|liver]
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one whole B-mode frame of the hot path over synthetic input already resident in HBM
(main.cpp:102-148): clear -> trace (BVH closest-hit + interface sampling) -> RF accumulation ->
[RCCL all-gather of the scan-line blocks when N > 1] -> PSF convolution -> envelope -> scan
conversion (400 x 500 image).  Weak scaling by default (each rank traces
`--scanlines` = 128 scan-lines x 1024 sample paths; the frame has 128*N scan-lines);
`--scanlines-total E` fixes the frame instead (strong scaling: BASELINE C4 = 256 x 8192,
C5 = 512 x 16384 sharded over 2/4/8 GPUs).

Frames are traced `--frames-in-flight` at a time (mcrt_trace_frames: every launch carries that many
consecutive frames' rays; images are bit-identical to one-at-a-time tracing); K steps are cut into
the fewest, evenly sized passes.  Passes are DOUBLE-BUFFERED: the all-gather + PSF convolution of
pass k run on a second stream beside the trace of pass k+1.  The timed region is EXACTLY K steps
between barrier + synchronize pairs; it is repeated (at least 5 times, at least ~0.6 s in total)
and the MEDIAN repeat is reported.  `--frames-in-flight 1` is the strict latency mode, also
reported as `one_frame_at_a_time`; `sweep` is the same pass with a different probe pose in every
frame (mcrt_trace_frames_poses: the moving probe of transducer.h:82-118).  `per_rank` carries every
rank's trace / gather / post-processing time per step (HIP events on the streams they run on).

The JSON line carries
  roofline      the dominant kernel (the BVH walk) against the guide's architectural VALU issue rate (`frac`), its
                lane-level share, the measured HBM share (`hbm.measured_frac`: fabric bytes from rocprofv3 PMC
                passes of this same command, taken live in child processes at the same pass size; `traffic`),
                the cache-served algorithmic rate, the vector-memory-pipe roof, the self-calibrated roof of
                earlier rounds (`calibrated`), and the same figures for k_march and k_shade (`kernels`);
  cpu_baseline  the oracle (a port of the reference algorithm) on this box's host cores;
  parity_check  frame 0 of the timed workload against the oracle, bit for bit.
"""
import argparse
import csv
import glob
import gc
import json
import os
import shutil
import statistics
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
N_SIMD = 1024           # 256 CUs x 4 SIMD-32
PMC_ROUND = next((r for r in ("round6", "round5", "round4") if os.path.exists(os.path.join(ROOT, "profiles", r, "pmc_bench.json"))), "round4")      # the newest committed single-GPU passes: what N > 1 runs fall back to
MIN_SPLIT_PASS = 48     # N > 1: a timed region is cut into two passes (gather + post of the first hidden behind the second) only if each has this many frames
TRACE_KERNELS = ("k_trace_lane<false", "k_trace_lane_wide", "k_trace_packet")   # the walk of the timed build, pooled: the lane walk's five-wavefront form (k_trace_lane_wide: the default since round 6) and its four-wavefront form (large trees, CU-masked streams) and bounce 1's ray packets
KERNEL_FAMILIES = {"walk": TRACE_KERNELS, "march": ("k_march<false",), "shade": ("k_shade<false",)}
ARCH_IPC = 0.5          # MI355X_MICROARCH.md: a wave64 VALU instruction issues in 2 cycles on the SIMD-32 -> 0.5 instructions per cycle and SIMD


def build_workload(m, name):
    if name == "random1m":
        cfg, meshes = m.synth.random_scene(1_000_000, 8, 12345)
        label = "1M random triangles (8 meshes, PCG64 12345)"
    elif name == "random16m":
        # the STREAMING regime (north_star's ">= 40 % of HBM during traversal" can only be judged where the BVH does not sit on-die):
        # 16 M triangles = 1.5 GB of records + ~0.5 GB of walked nodes, past the 256 MiB Infinity Cache; same total triangle area as random1m
        cfg, meshes = m.synth.random_scene(16_000_000, 8, 12345, edge=0.025)
        label = "16M random triangles (8 meshes, PCG64 12345, edge 0.025)"
    elif name == "sphere":
        cfg, meshes = m.synth.sphere_scene(5)
        label = "examples/sphere (icosphere 20480 tris + box)"
    elif name == "liver":
        cfg, meshes = m.synth.liver_scene(5)
        label = "ircad11-like liver (11 organs, ~225k tris)"
    else:
        raise SystemExit("unknown workload " + name)
    return cfg, m.scene_io.build_scene(cfg, meshes), label


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=128)
    ap.add_argument("--workload", default="random1m")
    ap.add_argument("--scanlines", type=int, default=128, help="scan-lines per GPU (weak scaling)")
    ap.add_argument("--scanlines-total", type=int, default=0, help="scan-lines of the whole frame, sharded over the GPUs (strong scaling); overrides --scanlines")
    ap.add_argument("--rays", type=int, default=1024, help="sample paths per scan-line")
    ap.add_argument("--rows", type=int, default=465)
    ap.add_argument("--tex-n", type=int, default=256, help="texture edge in voxels (256 = the reference; smaller only for cache experiments)")
    ap.add_argument("--frames-in-flight", type=int, default=128,
                    help="frames traced per pass (mcrt_trace_frames): a step is still ONE frame, but every kernel launch then carries the "
                         "rays of this many consecutive frames (1 = strict one-frame-at-a-time latency mode)")
    ap.add_argument("--bvh", default=None, choices=["sah", "lbvh"], help="BVH builder: host binned SAH (default; random16m: the device LBVH) or the device LBVH")
    ap.add_argument("--min-time", type=float, default=0.6, help="the K-step timed region is repeated until this many seconds are covered (>= 5 repeats)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency-leg", action="store_true", help="skip the extra one-frame-at-a-time and moving-probe measurements (profiling runs)")
    ap.add_argument("--no-pmc", action="store_true", help="do not take the live rocprofv3 PMC passes (HBM traffic, VALU instructions of k_trace)")
    ap.add_argument("--no-overlap", action="store_true", help="gather + PSF on the trace stream (no double buffering)")
    ap.add_argument("--no-split", action="store_true", help="N > 1: never cut a timed region into two passes")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)   # the process rocprofv3 profiles: warm-up + one K-step region, nothing else
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL); gloo only for plumbing checks")
    ap.add_argument("--same-gpu", action="store_true", help="plumbing check on a 1-GPU box: every rank uses GPU 0")
    ap.add_argument("--check-gather", action="store_true", help="rank 0 re-traces the last pass alone (all scan-lines in one process) and compares it with "
                                                                "the sharded + gathered + convolved frames bit for bit (`gather_check` in the JSON)")
    return ap.parse_args()


def passes(count, F):
    """split `count` frames into the fewest passes of at most F frames, as even as possible (20 with F=16 -> 10 + 10)"""
    if count <= 0:
        return []
    n_pass = -(-count // F)
    base, rem = divmod(count, n_pass)
    return [base + (1 if i < rem else 0) for i in range(n_pass)]


class Pipeline:
    """trace -> gather -> PSF -> envelope -> scan conversion over double-buffered passes.  The context's trace kernels run on
    `s_trace`; the collective and the post-processing of a finished pass on `s_post`, ordered by events, so they overlap the
    next pass's trace.  Every stage of every pass is bracketed by HIP events on the stream it runs on (`times()`)."""

    OUT_ROWS, OUT_COLS = 400, 500           # rf_image's scan-converted image (main.cpp:33 via rfimage.h:183-215)

    def __init__(self, torch, dist, ctx, psf, rank, world, E, e0, e1, R, F, backend, overlap, poses=None):
        from mcray_tracing_amd.dist import gather_rf
        self.torch, self.dist, self.ctx, self.psf, self.gather_rf = torch, dist, ctx, psf, gather_rf
        self.rank, self.world, self.E, self.e0, self.e1, self.R, self.F, self.backend = rank, world, E, e0, e1, R, F, backend
        self.s_trace = torch.cuda.Stream()
        self.s_post = torch.cuda.Stream() if overlap else self.s_trace
        self.buf = [torch.zeros((F, e1 - e0, R), dtype=torch.float32, device="cuda") for _ in range(2)]
        self.bmode = [torch.zeros((F, self.OUT_ROWS, self.OUT_COLS), dtype=torch.float32, device="cuda") for _ in range(2)] if rank == 0 else None
        self.ev_traced = [torch.cuda.Event() for _ in range(2)]
        self.ev_posted = [torch.cuda.Event() for _ in range(2)]
        self.frames = None          # the last pass's gathered + convolved + enveloped images [nf][E][R] (rank 0)
        self.images = None          # ... and its scan-converted B-mode images [nf][400][500] (rank 0)
        self.last = (0, 0)          # (first frame id, frames) of the last pass
        self.k = 0
        self.poses = poses          # (pos, dir) device tensors [F][E][3]: a probe pose per frame of a pass (the moving-probe leg)
        self.marks = []             # per pass: five timing events (trace start/end on s_trace; gather start, gather end, post end on s_post)
        self.timing = False
        ctx.set_stream(self.s_trace.cuda_stream)

    def post(self, frames, nf, out):
        """main.cpp:146-148 on the nf gathered frames [nf][E][R] (in place) -> out [nf][400][500], on the current stream of the context"""
        self.ctx.convolve_frames(frames, nf, self.E, self.R, self.psf.axial_kernel, self.psf.lateral_kernel)
        self.ctx.envelope_frames(frames, nf, self.E, self.R)
        self.ctx.scan_convert_frames(frames, nf, self.E, self.R, out, out_rows=self.OUT_ROWS, out_cols=self.OUT_COLS)

    def run_pass(self, frame, nf):
        torch, i = self.torch, self.k & 1
        self.k += 1
        rf = self.buf[i]
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)] if self.timing else None
        self.s_trace.wait_event(self.ev_posted[i])              # the pass that used this buffer two passes ago is done with it
        self.ctx.set_stream(self.s_trace.cuda_stream)
        if ev: ev[0].record(self.s_trace)
        if self.poses is not None:
            self.ctx.trace_frames_poses(frame, self.poses[0][:nf], self.poses[1][:nf], rf, self.e0, self.e1)
        else:
            self.ctx.trace_frames(frame, nf, rf, self.e0, self.e1)
        if ev: ev[1].record(self.s_trace)
        self.ev_traced[i].record(self.s_trace)
        self.s_post.wait_event(self.ev_traced[i])
        with torch.cuda.stream(self.s_post):
            # ONE collective per pass (RCCL gather over xGMI of the [nf][E/N][R] blocks to rank 0), then the rest of the B-mode frames there
            if ev: ev[2].record(self.s_post)
            frames = self.gather_rf(rf[:nf], self.E, self.R, self.dist if self.world > 1 else None, root=0 if self.world > 1 else None)   # to rank 0 only: it alone post-processes
            if ev: ev[3].record(self.s_post)
            if self.rank == 0:
                self.ctx.set_stream(self.s_post.cuda_stream)
                self.post(frames, nf, self.bmode[i])
                self.ctx.set_stream(self.s_trace.cuda_stream)
                self.images = self.bmode[i][:nf]
            if ev: ev[4].record(self.s_post)
            self.frames = frames
        if ev: self.marks.append(ev)
        self.last = (frame, nf)
        self.ev_posted[i].record(self.s_post)

    def run_steps(self, first, count, F=None):
        f = first
        for nf in passes(count, F or self.F):
            self.run_pass(f, nf)
            f += nf

    def sync(self):
        """barrier + synchronize: the bracket around a timed region.  At its END the clock is read after `drain()` and BEFORE this
        (the maximum over the ranks is taken afterwards, so the barrier's own latency is not part of anybody's time)."""
        self.torch.cuda.synchronize()
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()
        self.ctx.synchronize()          # (raises if a persistent kernel's watchdog abandoned a launch: a number must not come from a broken frame)

    def drain(self):
        self.torch.cuda.synchronize()   # everything this rank enqueued -- its passes, the gather, rank 0's post-processing -- has finished

    def times(self, steps):
        """ms per step of this rank's trace / gather / post-processing over the passes recorded since `timing` was switched on"""
        t = {"trace_ms": 0.0, "gather_ms": 0.0, "post_ms": 0.0}
        for ev in self.marks:
            t["trace_ms"] += ev[0].elapsed_time(ev[1]); t["gather_ms"] += ev[2].elapsed_time(ev[3]); t["post_ms"] += ev[3].elapsed_time(ev[4])
        self.marks = []
        return {k: v / max(1, steps) for k, v in t.items()}


def main():
    args = parse_args()
    if args.bvh is None:
        args.bvh = "lbvh" if args.workload == "random16m" else "sah"
    import numpy as np
    import torch
    import mcray_tracing_amd as m

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ..." % (args.gpus, args.gpus))
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if args.same_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
        seen = dist.get_world_size()
        assert seen == world, "the process group has %d ranks, WORLD_SIZE says %d" % (seen, world)

    from mcray_tracing_amd.dist import shard_range
    S, R = args.rays, args.rows
    strong = args.scanlines_total > 0
    E = args.scanlines_total if strong else args.scanlines * world
    e0, e1 = shard_range(rank, world, E)
    E_local = e1 - e0
    cfg, sd, label = build_workload(m, args.workload)
    tr = m.Transducer(E, position=cfg["transducerPosition"], angles_deg=cfg["transducerAngles"])
    ctx = m.Context(local_rank)
    ctx.set_params(n_elements=E, n_samples=S, n_rows=R, frequency=tr.frequency, tex_n=args.tex_n)
    t0 = time.time()
    ctx.set_bvh_builder(args.bvh)
    ctx.upload_scene(sd)
    t_bvh = time.time() - t0
    ctx.upload_texture(None, args.tex_n)
    ctx.set_transducer(tr.pos, tr.dir)
    psf = m.Psf(freq=tr.frequency)
    F = max(1, min(args.frames_in_flight, 256))
    K, W = args.steps, args.warmup
    if world > 1 and not args.no_split and K >= 2 * MIN_SPLIT_PASS:
        # N > 1: a timed region of ONE pass has its gather + rank-0 post-processing exposed at the end (nothing to overlap them with).  Cutting
        # the region into two passes hides them behind the second pass's trace -- but smaller passes pay more in launch tails: measured on
        # one MI355X (profiles/round4/exp_pass_split.txt), 20 steps as 10 + 10 cost 0.500 ms per step against 0.415 as one pass (+20 %), 64 as
        # 32 + 32 0.388 against 0.366 (+6 %), while the exposed gather + post of a 20-frame pass of 8 x 128 scan-lines is ~4 % of it
        # (33 MB over seven xGMI links, three memory-bound kernels).  So the region is cut only where a half still fills the GPU.
        F = min(F, -(-K // 2))
    pipe = Pipeline(torch, dist, ctx, psf, rank, world, E, e0, e1, R, F, args.backend, not args.no_overlap)

    if args.pmc_child:
        # the process the PMC passes profile: TWO K-step regions (the first as warm-up), so that every profiled launch has a pass
        # size of the timed region -- the per-launch averages are then those of the launches the live kernel time is taken over
        pipe.run_steps(1000, K); pipe.sync()
        pipe.run_steps(0, K); pipe.sync()
        ctx.close()
        return

    # ---- counted algorithmic work of the K timed frames (counting build of the same kernels, untimed) ----
    # (the counting build walks every ray once, i.e. exactly the oracle's node/triangle visits: the timed kernel may cut the
    #  rays of a small bounce into pieces, whose extra visits are overhead, not algorithmic work)
    ctx.enable_stats(True); ctx.get_stats(reset=True)
    pipe.run_steps(0, K); pipe.sync()
    st = ctx.get_stats(reset=True)
    ctx.enable_stats(False)
    # Algorithmic bytes (adapted to the BVH4 nodes the walk reads: 64 B with half-float boxes for the lane-per-ray
    # walk, 128 B for the quad walk).  The dominant kernel is the walk, launched once per bounce and pass: per closest-hit query
    # nodes*64 B + triangles*48 B + the 32-B ray read and 32-B hit record written.
    node_bytes = 64
    trace_bytes_frame = (st["nodes_visited"] * node_bytes + st["tris_tested"] * 48 + st["queries"] * 64) / K
    # the rest of the frame, for the record: 64-B segment written + read, 8-B texture gather per RF step, RF block + bins
    other_bytes_frame = (st["segments"] * 128 + st["rf_steps"] * 8) / K + E_local * R * (4 + 8)

    # ---- the timed region: EXACTLY K steps between barrier + synchronize pairs, repeated; the median repeat is reported ----
    # (warm-up: at least W steps, rounded up to whole timed regions, so that the first timed repeat finds every buffer at its size)
    W_run = -(-max(W, 1) // K) * K
    ctx.enable_timing(True)                                    # (before the warm-up: its HIP events are created there, not in the first timed repeat)
    pipe.timing = True
    for w0 in range(0, W_run, K):
        pipe.run_steps(1000 + w0, K)
    pipe.sync()
    ctx.kernel_time(reset=True); pipe.times(1)
    reps, total, n_rep, k_sum, k_n = [], 0.0, 0, 0.0, 0
    stage = {"trace_ms": 0.0, "gather_ms": 0.0, "post_ms": 0.0}
    gc.collect(); gc.disable()                                 # (as timeit does: a full collection of the interpreter's heap is tens of ms, a whole repeat at this size)
    while n_rep < 5 or (total < args.min_time and n_rep < 200):
        pipe.sync()
        t0 = time.perf_counter()
        pipe.run_steps(0, K)
        pipe.drain()
        dt = time.perf_counter() - t0
        pipe.sync()
        dt_t = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        if world > 1:
            dist.all_reduce(dt_t, op=dist.ReduceOp.MAX)       # MAX over ranks (every rank then takes the same loop decisions)
        dt = float(dt_t.item())
        reps.append(dt); total += dt; n_rep += 1
        a_ms, a_n = ctx.kernel_time(reset=True)                # harvested per repeat (outside the clock), so the same HIP events serve every repeat
        k_sum += a_ms * a_n; k_n += a_n
        for k_, v_ in pipe.times(1).items():
            stage[k_] += v_
    gc.enable()
    k_ms = k_sum / max(k_n, 1)                                 # HIP events around every k_trace launch, on the launching stream
    ctx.enable_timing(False)
    pipe.timing = False
    mine = dict({k_: v_ / (K * n_rep) for k_, v_ in stage.items()}, rank=rank, scan_lines=[e0, e1])
    per_rank = [mine]
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    dt = statistics.median(reps)
    launches_per_frame = k_n / (K * n_rep)                     # max_depth bounces / frames per pass
    alg_bytes = trace_bytes_frame / launches_per_frame

    if rank == 0:
        gather_check = None
        if args.check_gather:
            # the sharded path against one process tracing every scan-line: same frames, bit for bit (PSF included)
            f0, nf = pipe.last
            torch.cuda.synchronize()                               # (rank 0 only: no collective here)
            alone = torch.zeros((nf, E, R), dtype=torch.float32, device="cuda")
            alone_img = torch.zeros((nf, pipe.OUT_ROWS, pipe.OUT_COLS), dtype=torch.float32, device="cuda")
            ctx.set_stream(pipe.s_trace.cuda_stream)
            ctx.trace_frames(f0, nf, alone, 0, E)
            pipe.post(alone, nf, alone_img)
            torch.cuda.synchronize()
            gather_check = {"equal": bool(torch.equal(alone.view(torch.int32), pipe.frames.view(torch.int32)) and torch.equal(alone_img.view(torch.int32), pipe.images.view(torch.int32))),
                            "frames": nf, "first_frame": f0, "scan_lines": E, "ranks": world, "backend": args.backend,
                            "nonzero": int(torch.count_nonzero(alone).item()), "nonzero_bmode": int(torch.count_nonzero(alone_img).item())}
        rays = E * S * K
        alg_gbs = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        pass_sizes = passes(K, F)
        out = {
            "metric": "rays/sec (Monte-Carlo sample paths traced + accumulated into whole B-mode frames: PSF, envelope, scan conversion)",
            "value": rays / dt, "unit": "rays/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": dt / K * 1e3, "frames_per_sec": K / dt, "higher_is_better": True,
            "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %dx%d rays/GPU, depth 10, %d rows" % (label, E_local, S, R),
                       "scan_lines_total": E, "rays_per_scan_line": S, "triangles": int(sd.n_tri), "parallelism": "scanline-shard x%d" % world,
                       "frames_in_flight": F, "passes_per_timed_region": pass_sizes, "last_pass_gather_and_post_exposed": world > 1, "gather": "none (one GPU)" if world == 1 else "RCCL gather of the scan-line blocks to rank 0, one per pass, double-buffered against the next pass's trace", "timed_region_repeats": n_rep,
                       "timed_seconds_total": total, "repeat_ms_per_step_min_median_max": [min(reps) / K * 1e3, dt / K * 1e3, max(reps) / K * 1e3], "slowest_repeat": reps.index(max(reps)),
                       "overlap_gather_psf_with_next_trace": not args.no_overlap, "bvh_builder": args.bvh, "bvh_build_s": round(t_bvh, 3),
                       "warmup_steps_run": W_run, "step": "clear, trace, accumulate, [all-gather], PSF, envelope, scan conversion to %dx%d" % (pipe.OUT_ROWS, pipe.OUT_COLS)},
            "ranks_seen": world, "per_rank": per_rank,
        }
        per_frame = {k: v / K for k, v in st.items()}
        pmc = None
        if world == 1 and not args.no_pmc:
            pmc = live_pmc(args)
        if pmc is None:
            pmc = committed_pmc(args, pass_sizes, per_frame["queries"] / launches_per_frame)
        roof = roofline_from(pmc, k_ms, alg_gbs, alg_bytes)
        roof.update({"kernel": "the walk, one launch per bounce, pooled: k_trace_lane_wide (a lane per ray, five wavefronts per SIMD: every launch since round 6; k_trace_lane<false>, the four-wavefront form, for trees past half the Infinity Cache) and k_trace_packet (bounce 1 of passes of >= 262144 paths: a wavefront per ray packet)", "kernel_ms": k_ms, "launches": k_n,
                     "launches_per_frame": launches_per_frame, "algorithmic_bytes_per_launch": alg_bytes, "algorithmic_GBps_cache_served": alg_gbs, "node_bytes": node_bytes,
                     "trace_bytes_per_frame": trace_bytes_frame, "other_stage_bytes_per_frame": other_bytes_frame, "per_frame": per_frame})
        if world == 1 and not args.no_latency_leg:
            # ---- the bounce's OTHER kernels: one more region with k_shade's and k_march's launches bracketed too (untimed: the timed
            # region carries the walk's events only), and the same region on a context whose kernels do not overlap (each alone on the GPU)
            ctx.enable_timing(2); ctx.kernel_times(reset=True)
            pipe.run_steps(4000, K); pipe.sync()
            t_over = ctx.kernel_times(reset=True)
            ctx.enable_timing(False)
            t_alone = None
            prev_env = {k_: os.environ.get(k_) for k_ in ("MCRT_TUNING", "MCRT_NO_OVERLAP")}      # (a caller's own MCRT_TUNING=1 + knobs must survive this leg)
            try:
                os.environ["MCRT_TUNING"] = "1"; os.environ["MCRT_NO_OVERLAP"] = "1"
                ctx2 = m.Context(local_rank)
            finally:
                for k_, v_ in prev_env.items():
                    if v_ is None: os.environ.pop(k_, None)
                    else: os.environ[k_] = v_
            try:
                ctx2.set_params(n_elements=E, n_samples=S, n_rows=R, frequency=tr.frequency, tex_n=args.tex_n)
                ctx2.set_bvh_builder(args.bvh); ctx2.upload_scene(sd); ctx2.upload_texture(None, args.tex_n); ctx2.set_transducer(tr.pos, tr.dir)
                ctx2.set_stream(pipe.s_trace.cuda_stream)
                f = 5000
                for rep in range(2):
                    if rep == 1: ctx2.enable_timing(2); ctx2.kernel_times(reset=True)
                    for nf in pass_sizes:
                        ctx2.trace_frames(f, nf, pipe.buf[0], e0, e1); f += nf
                    ctx2.synchronize()
                t_alone = ctx2.kernel_times(reset=True)
            finally:
                ctx2.close(); ctx.set_stream(pipe.s_trace.cuda_stream)
            clock = valu_calibration()["clock_ghz"]
            other = (pmc or {}).get("other_kernels", {})
            lpf = launches_per_frame                      # launches per frame of every per-bounce kernel
            roof["kernels"] = {
                "k_trace_lane": {"what": "the walk (this block's top level): duration beside k_march / alone", "ms_per_launch_overlapped": t_over["walk"][0], "ms_per_launch_alone": t_alone["walk"][0],
                                 "dilation_beside_the_rest": t_over["walk"][0] / t_alone["walk"][0] if t_alone["walk"][0] else None},
                "k_march": kernel_block("k_march", "RF accumulation (main.cpp:106-144): per RF step one 8-byte texture gather + the sequential advance of point, time, intensity that reproduces the reference's float recurrence bit for bit (both lanes of a pair repeat it: half of its ~63 lane-instructions per step)",
                                        t_over["march"][0], t_alone["march"][0], other.get("march"),
                                        (per_frame["rf_steps"] * 8.0 + per_frame["segments"] * 48.0) / lpf, clock,
                                        note="algorithmic bytes = RF steps x 8 B (texture gather) + segments x 48 B (march record); runs on a low-priority side stream beside the NEXT bounce's walk"),
                "k_shade": kernel_block("k_shade", "interface physics + queue compaction (scene.cpp:122-165, ray.cpp:11-97), one lane per ray",
                                        t_over["shade"][0], t_alone["shade"][0], other.get("shade"), per_frame["queries"] * 212.0 / lpf, clock,
                                        note="algorithmic bytes = 212 B per ray (state in and out 96, march record 48, winning triangle 3 x 16, keys / queue / counts 20)"),
            }
        roof["salu"] = salu_block(pmc)
        out["roofline"] = roof
        if world == 1 and F > 1 and not args.no_latency_leg:
            # the same workload strictly one frame at a time (each launch carries one frame's rays), for the record
            pipe.run_steps(2000, min(W, 16), F=1)
            lat = []
            for _ in range(3):
                pipe.sync(); t1 = time.perf_counter()
                pipe.run_steps(0, K, F=1)
                pipe.sync(); lat.append(time.perf_counter() - t1)
            dt1 = statistics.median(lat)
            out["one_frame_at_a_time"] = {"value": E * S * K / dt1, "unit": "rays/s", "ms_per_step": dt1 / K * 1e3, "frames_per_sec": K / dt1}
            # the moving probe: every frame of a pass with its own pose (a 30-degree sweep about the probe's axis over the pass, transducer.h:82-118)
            sweep = [m.Transducer(E, position=cfg["transducerPosition"], angles_deg=np.asarray(cfg["transducerAngles"], np.float64) + np.array([30.0 * f / F - 15.0, 0.0, 0.0]))
                     for f in range(F)]
            pipe.poses = (torch.from_numpy(np.stack([t.pos for t in sweep])).cuda(), torch.from_numpy(np.stack([t.dir for t in sweep])).cuda())
            pipe.run_steps(3000, K)
            lat = []
            for _ in range(3):
                pipe.sync(); t1 = time.perf_counter()
                pipe.run_steps(0, K)
                pipe.sync(); lat.append(time.perf_counter() - t1)
            pipe.poses = None
            dts = statistics.median(lat)
            out["sweep"] = {"value": E * S * K / dts, "unit": "rays/s", "ms_per_step": dts / K * 1e3, "frames_per_sec": K / dts, "frames_in_flight": F,
                            "what": "the timed region with a different probe pose in every frame of a pass (mcrt_trace_frames_poses)"}
        if world == 1 and not args.no_latency_leg and pipe.F >= 2:
            # the STEADY STATE, for the record beside the driver's pass size: whole passes of min(128, frames-in-flight) frames, outside
            # the timed region like the two legs above -- what a caller who keeps the GPU fed sees per frame (tails of a launch amortised over more rays)
            P = min(128, pipe.F)
            pipe.run_steps(6000, P); pipe.sync()               # (work buffers grow to the pass size here, not inside the clock)
            lat = []
            for _ in range(5):
                pipe.sync(); t1 = time.perf_counter()
                pipe.run_steps(0, P)
                pipe.drain(); lat.append(time.perf_counter() - t1)
            pipe.sync()
            dtp = statistics.median(lat)
            out["pass%d" % P] = {"value": E * S * P / dtp, "unit": "rays/s", "ms_per_step": dtp / P * 1e3, "frames_per_sec": P / dtp, "frames_in_flight": P, "repeats": len(lat),
                                 "ms_per_step_min_max": [min(lat) / P * 1e3, max(lat) / P * 1e3],
                                 "what": "one pass of %d frames per repeat (whole B-mode frames: trace, accumulate, PSF, envelope, scan conversion), median of %d" % (P, len(lat))}
        if world > 1 and not args.no_cpu_baseline:
            # N > 1: the oracle on rank 0's OWN shard (seconds), so that a scaling line carries oracle evidence and not only gather_check
            ctx.set_stream(pipe.s_trace.cuda_stream)
            ctx.trace_frames(0, pass_sizes[0], pipe.buf[0], e0, e1)
            torch.cuda.synchronize(); ctx.synchronize()
            out["parity_check"] = shard_parity(m, sd, tr, ctx, S, R, pipe.buf[0][0].cpu().numpy(), e0, e1)
        if world == 1 and not args.no_cpu_baseline:
            # frame 0 of the timed workload, traced the way the timed region traces it (first pass), before the PSF
            ctx.set_stream(pipe.s_trace.cuda_stream)
            ctx.trace_frames(0, pass_sizes[0], pipe.buf[0], e0, e1)
            pipe.sync()
            rf0 = pipe.buf[0][0].cpu().numpy()                 # [E][R]
            out["cpu_baseline"], out["parity_check"] = cpu_baseline(m, sd, tr, ctx, S, R, rf0)
        if gather_check is not None:
            out["gather_check"] = gather_check
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------ roofline
def valu_calibration():
    """profiles/round4/valu_roof.json (tools/valu_roof.hip on the MI355X, round 4): the VALU issue ceiling in wave-instructions per cycle and
    SIMD.  It depends on the instruction class (0.25-0.28 for fp32 fma / min3 / compare-select streams, 0.40-0.43 for plain integer adds),
    so the roof the walk is priced against is the one measured for ITS mix: a register-only replica of the node step the kernel runs TODAY
    (12 v_cndmask plane picks, 24 v_fma_mix_f32, 4 x max / max3 / min / min3 / compare, key build and ranking, branch-free push offsets,
    child pick, address: 91 instructions) at the walk's occupancy of 4 wavefronts per SIMD, with the clock the chip held meanwhile."""
    try:
        with open(os.path.join(ROOT, "profiles", "round4", "valu_roof.json")) as f:
            d = json.load(f)
        mix = [r for r in d["results"] if r["class"].startswith("BVH4 LANE node-step mix") and r["waves_per_simd"] == 4][0]
        best = max(r["simd_ipc"] for r in d["results"] if not r["class"].startswith("node fetch"))
        return {"ipc_per_simd": mix["simd_ipc"], "clock_ghz": mix["clock_ghz"], "class": mix["class"], "waves_per_simd": 4,
                "best_class_ipc_per_simd": best, "source": "profiles/round4/valu_roof.json"}
    except Exception:
        # MI355X_MICROARCH.md: a wave64 VALU instruction issues in 2 cycles on the SIMD-32 once >= 2 waves share a SIMD; 2.4 GHz max clock
        return {"ipc_per_simd": 0.5, "clock_ghz": 2.4, "source": "MI355X_MICROARCH.md (no calibration file)"}


def tcp_access_cost():
    """profiles/round4/tcp_access_cost.json: cycles of a CU's vector memory pipe per access TCP_TOTAL_CACHE_ACCESSES counts (one per lane; a
    uniform adjacent quad counts once), measured for every sharing pattern -- the CHEAPEST pattern is the floor of what an access costs"""
    try:
        with open(os.path.join(ROOT, "profiles", "round4", "tcp_access_cost.json")) as f:
            return float(json.load(f)["min_cycles_per_counted_access"]), "profiles/round4/tcp_access_cost.json"
    except Exception:
        return 1.32, "default (no calibration file)"


def roofline_from(pmc, k_ms, alg_gbs, alg_bytes):
    """The walk against its roofs, GUIDE-ANCHORED numbers first.
    The path has no dense contraction (no MFMA) and its tree is served from L1 / L2 / Infinity Cache (fabric traffic is a few per cent of the HBM
    peak at 1 M triangles), so neither of the contract's two roofs binds; the roof that does is on the CU:
      frac      = wave-level VALU instructions per second (SQ_INSTS_VALU of the launch / its HIP-event duration) over the ARCHITECTURAL issue rate,
                  1024 SIMDs x 0.5 instructions per cycle (MI355X_MICROARCH.md: 2 cycles per wave64 instruction) x the clock the calibration run held;
                  frac_lane_level = frac x the share of lanes active in those instructions
      hbm       = the contract's HBM view, kept beside it: algorithmic bytes per launch over the launch's duration (a cache-served rate: it may exceed
                  the HBM peak) and the MEASURED fabric bytes per launch (`traffic`; separate --pmc passes, gfx950 correction) over the same duration
      second_roof = the CU's vector memory pipe (time the launch's counted cache accesses need at the cheapest measured cost per access)
      calibrated  = rounds 2-4's self-calibrated roof (the issue rate of a register-only replica of the kernel's own node step): secondary"""
    cal = valu_calibration()
    clock = cal["clock_ghz"]
    peak = N_SIMD * ARCH_IPC * clock                                   # G wave-instructions / s, architectural
    r = {"bound": "valu", "unit": "Ginstr/s", "peak": peak,
         "peak_is": "architectural VALU issue rate: 1024 SIMDs x 0.5 wave64 instructions per cycle (MI355X_MICROARCH.md) x %.3f GHz (the clock held during tools/valu_roof.hip)" % clock,
         "bound_note": "neither HBM nor MFMA binds this kernel (cache-resident tree, no dense contraction); the contract's HBM figures are under `hbm`, `traffic` is the measured fabric bytes per launch",
         "kernel_ms_is": "HIP-event time of the walk's launches on the stream they run on, with k_march running beside them on its side stream (contention included)"}
    hbm = {"peak_GBps": HBM_PEAK_GBS, "algorithmic_bytes_per_launch": alg_bytes, "algorithmic_GBps_cache_served": alg_gbs, "algorithmic_over_hbm_peak": alg_gbs / HBM_PEAK_GBS}
    if pmc and pmc.get("valu_instructions_per_launch") and k_ms > 0:
        ach = pmc["valu_instructions_per_launch"] / (k_ms * 1e-3) / 1e9
        cal_peak = N_SIMD * cal["ipc_per_simd"] * clock
        r.update({"achieved": ach, "frac": ach / peak, "frac_vs_architectural": ach / peak, "derived": bool(pmc.get("derived"))})
        if pmc.get("lane_utilisation") is not None:
            r["valu_lane_utilisation"] = pmc["lane_utilisation"]
            r["frac_lane_level"] = ach / peak * pmc["lane_utilisation"]
        r["calibrated"] = {"what": "rounds 2-4's roof: the issue rate of a register-only replica of this kernel's CURRENT node step (tools/valu_roof.hip) -- 'this code minus its memory stalls', not a hardware peak",
                           "peak": cal_peak, "frac": ach / cal_peak, "frac_vs_best_class": ach / (N_SIMD * cal.get("best_class_ipc_per_simd", 0.43) * clock), "peak_source": cal}
    else:
        r.update({"achieved": None, "frac": None, "frac_vs_architectural": None})
    if pmc and pmc.get("tcp_lane_accesses_per_launch") and k_ms > 0:
        # the OTHER roof of the walk: the CU's vector memory pipe.  Cost model (round 4, fetch_roof_same under --pmc): every access the
        # counter counts costs at least `cost` cycles of its CU's pipe, whatever the sharing pattern; the launch cannot be shorter than its accesses
        # x cost / (256 CUs x clock).  The accesses are counted on the walk's FOUR-wavefront form (what the walk needs); where the timed launches ran
        # in the five-wavefront form (every launch since round 6: its refill code spills, coalesced scratch accesses the counter counts per lane) the as-run count
        # is reported beside it and the pair (needed accesses, as-run duration) is labelled as mixed.
        cost, src = tcp_access_cost()
        need, as_run = pmc["tcp_lane_accesses_per_launch"], pmc.get("tcp_lane_accesses_per_launch_as_run")
        t_ms = need * cost / 256.0 / (clock * 1e6)
        same_form = as_run is None or abs(as_run - need) <= 0.02 * need
        r["second_roof"] = {"what": "vector memory pipe (TCP) of the CUs: time the launch's counted cache accesses need at the cheapest measured cost per access",
                            "accesses_per_launch": need, "accesses_per_launch_as_run": as_run, "cycles_per_access_floor": cost, "clock_ghz": clock,
                            "pipe_ms": t_ms, "kernel_ms": k_ms, "frac": t_ms / k_ms,
                            "consistent_pair": bool(same_form),
                            "pair_note": ("accesses and duration are of the same (four-wavefront) form" if same_form else
                                          "MIXED: accesses counted on the four-wavefront form, duration of the five-wavefront form the timed launches took; as-run accesses x cost / duration = %.3f is an upper bound (spill accesses are served a wavefront at a time)" % (as_run * cost / 256.0 / (clock * 1e6) / k_ms)),
                            "source": "TCP_TOTAL_CACHE_ACCESSES of the walk's launches; cost: " + src}
        if r.get("frac") is not None:
            calf = r["calibrated"]["frac"]
            r["binding_roof"] = "vector memory pipe (second_roof)" if t_ms / k_ms > calf else "valu issue (calibrated mix)"
            # sensitivity builds (profiles/round4/exp_sensitivity.txt): one more load per node step costs the walk 3 %, ten more instructions 2 % -- the
            # nearer roof is not a wall on its own, the walk sits at the knee of the two
            r["binding_roof_note"] = "compared on the self-calibrated scale (pipe time share vs calibrated VALU share); both within a few per cent of the launch; marginal costs in profiles/round4/exp_sensitivity.txt"
    if pmc and pmc.get("traffic_bytes_per_launch") is not None and k_ms > 0:
        r["traffic"] = pmc["traffic_bytes_per_launch"]
        hbm["measured_GBps"] = pmc["traffic_bytes_per_launch"] / (k_ms * 1e-3) / 1e9
        hbm["measured_frac"] = hbm["measured_GBps"] / HBM_PEAK_GBS
        hbm["traffic_over_algorithmic"] = pmc["traffic_bytes_per_launch"] / alg_bytes if alg_bytes else None
        if pmc.get("fetch_size_kib") is not None:
            # profiles/round4/fetch_size_calibration.json: the walk's scattered 64-byte node reads are ONE fabric request each, counted as 64 bytes;
            # the guide's doubling (right for coalesced 128-byte requests) is an upper bound for them.  Both readings are reported.
            one = (pmc["fetch_size_kib"] + pmc["write_size_kib"]) * 1024.0
            hbm["traffic_one_unit_per_request"] = one
            hbm["measured_frac_one_unit_per_request"] = one / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        r["hbm_measured_frac"] = hbm["measured_frac"]
    else:
        r["traffic"] = None; r["hbm_measured_frac"] = None
    r["hbm"] = hbm
    r["pmc"] = pmc
    return r


def salu_block(pmc):
    """The scalar pipe: a CU's four SIMDs share ONE scalar ALU, which issues at most one instruction per cycle.  Per kernel:
    SQ_INSTS_SALU per launch over the launch's busy CU cycles (SQ_BUSY_CU_CYCLES, summed over the 256 CUs) = scalar instructions per cycle and CU,
    i.e. the share of that pipe's issue slots the kernel fills while it runs alone under --pmc; and scalar per vector instruction."""
    def one(d):
        if not d or not d.get("salu_instructions_per_launch") or not d.get("busy_cu_cycles_per_launch"):
            return None
        per_cu_cycle = d["salu_instructions_per_launch"] / (d["busy_cu_cycles_per_launch"] * 256.0)
        return {"salu_instructions_per_launch": d["salu_instructions_per_launch"], "busy_cycles_per_cu": d["busy_cu_cycles_per_launch"],
                "salu_per_cycle_and_cu": per_cu_cycle, "frac_of_scalar_issue": per_cu_cycle / 1.0,
                "salu_per_valu_instruction": d["salu_instructions_per_launch"] / d["valu_instructions_per_launch"] if d.get("valu_instructions_per_launch") else None}
    if not pmc:
        return None
    o = {"what": "scalar ALU issue: SQ_INSTS_SALU per launch / busy CU cycles (one scalar ALU per CU, one instruction per cycle); kernels run one at a time under --pmc",
         "peak": 1.0, "unit": "scalar instructions per cycle and CU", "walk": one(pmc)}
    for fam, d in (pmc.get("other_kernels") or {}).items():
        o["k_" + fam] = one(d)
    return o if any(v for k_, v in o.items() if k_ in ("walk", "k_march", "k_shade")) else None


def kernel_block(name, what, t_ms, t_alone_ms, pmc_k, alg_bytes, clock, note=None):
    """one of the bounce's other kernels in the walk's terms: duration per launch beside the rest of the pipeline and alone (dilation), its VALU
    issue share of the architectural rate over its own duration, lane-level share, algorithmic and measured bytes"""
    peak = N_SIMD * ARCH_IPC * clock
    o = {"what": what, "ms_per_launch_overlapped": t_ms, "ms_per_launch_alone": t_alone_ms,
         "dilation_beside_the_rest": (t_ms / t_alone_ms) if t_ms and t_alone_ms else None,
         "algorithmic_bytes_per_launch": alg_bytes}
    if note: o["note"] = note
    for label, t in (("overlapped", t_ms), ("alone", t_alone_ms)):
        if pmc_k and t:
            ach = pmc_k["valu_instructions_per_launch"] / (t * 1e-3) / 1e9
            o["valu_frac_vs_architectural_" + label] = ach / peak
            if alg_bytes: o["algorithmic_GBps_" + label] = alg_bytes / (t * 1e-3) / 1e9
            if pmc_k.get("traffic_bytes_per_launch") is not None: o["hbm_measured_frac_" + label] = pmc_k["traffic_bytes_per_launch"] / (t * 1e-3) / 1e9 / HBM_PEAK_GBS
    if pmc_k:
        o.update({"valu_instructions_per_launch": pmc_k["valu_instructions_per_launch"], "valu_lane_utilisation": pmc_k.get("lane_utilisation"),
                  "tcp_lane_accesses_per_launch": pmc_k.get("tcp_lane_accesses_per_launch"), "traffic": pmc_k.get("traffic_bytes_per_launch")})
    return o


def _pmc_rows(d, family="walk"):
    """counter values of one kernel family's launches (the walk: both of its kernels), per counter"""
    per = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if any(k in row["Kernel_Name"] for k in KERNEL_FAMILIES[family]):
                per.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    if not per:
        return {}, None
    return per, " + ".join(k + ">" if k.endswith("false") else k for k in KERNEL_FAMILIES[family])


def live_pmc(args):
    """rocprofv3 --pmc passes of THIS command (same workload, same pass sizes) in child processes; per k_trace launch:
    fabric-side bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB (gfx950: FETCH_SIZE reports half the bytes of wide reads,
    MI355X_MICROARCH.md, HBM section; separate passes because the TCC counters do not fit one) and SQ_INSTS_VALU."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    child = [sys.executable, os.path.abspath(__file__), "--pmc-child", "--steps", str(args.steps), "--warmup", str(min(args.warmup, args.frames_in_flight)),
             "--workload", args.workload, "--scanlines", str(args.scanlines), "--scanlines-total", str(args.scanlines_total), "--rays", str(args.rays),
             "--rows", str(args.rows), "--tex-n", str(args.tex_n), "--frames-in-flight", str(args.frames_in_flight), "--bvh", args.bvh]
    got = {}
    other = {}
    tmp = tempfile.mkdtemp(prefix="mcrt_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for name, ctrs in (("sq", ["SQ_INSTS_VALU", "SQ_BUSY_CU_CYCLES", "SQ_WAVES", "SQ_INSTS_VMEM_RD", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU"]),
                           ("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"]), ("tcp", ["TCP_TOTAL_CACHE_ACCESSES_sum"]), ("tcp_all", ["TCP_TOTAL_CACHE_ACCESSES_sum"])):
            d = os.path.join(tmp, name)
            # (the walk's ACCESSES are counted on its four-wavefront form: the five-wavefront form, which the timed launches take, adds the spill
            #  traffic of its refill code -- coalesced 4-byte scratch accesses that the counter counts per lane but the pipe serves a wavefront
            #  at a time, so the floor price per counted access does not apply to them; "tcp_all" counts that form as it runs)
            env_pass = dict(env, MCRT_TUNING="1", MCRT_WIDE_FROM="4294967295") if name == "tcp" else env
            r = subprocess.run([exe, "--pmc"] + ctrs + ["--output-format", "csv", "-d", d, "--"] + child, cwd="/tmp", env=env_pass,
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=240)
            rows, kname = _pmc_rows(d)
            if r.returncode != 0 or not rows:
                sys.stderr.write("bench.py: PMC pass '%s' failed (rc %s): %s\n" % (name, r.returncode, (r.stdout or b"")[-600:].decode(errors="replace")))
                return None
            got["kernel"] = kname
            for c, v in rows.items():
                got[c + "@all" if name == "tcp_all" else c] = (sum(v) / len(v), len(v))
            if name != "tcp":                                   # the other two kernels of a bounce, from the same passes (as run)
                for fam in ("march", "shade"):
                    rows_f, _ = _pmc_rows(d, fam)
                    for c, v in rows_f.items():
                        other.setdefault(fam, {})[c] = (sum(v) / len(v), len(v))
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    try:
        return {"source": "live: rocprofv3 --pmc child passes of this command (same pass sizes), per launch of the walk kernel", "kernel": got["kernel"],
                "launches_profiled": got["SQ_INSTS_VALU"][1],
                "valu_instructions_per_launch": got["SQ_INSTS_VALU"][0], "busy_cu_cycles_per_launch": got["SQ_BUSY_CU_CYCLES"][0] / 256.0,
                "vmem_read_instructions_per_launch": got["SQ_INSTS_VMEM_RD"][0], "salu_instructions_per_launch": got["SQ_INSTS_SALU"][0],
                "lane_utilisation": got["SQ_THREAD_CYCLES_VALU"][0] / (64.0 * got["SQ_ACTIVE_INST_VALU"][0]) if got["SQ_ACTIVE_INST_VALU"][0] else None,
                "tcp_lane_accesses_per_launch": got["TCP_TOTAL_CACHE_ACCESSES_sum"][0],
                "tcp_lane_accesses_counted_on": "the walk's four-wavefront form (MCRT_WIDE_FROM off for this pass): the accesses the walk needs",
                "tcp_lane_accesses_per_launch_as_run": got.get("TCP_TOTAL_CACHE_ACCESSES_sum@all", (None, 0))[0],      # (with the five-wavefront form's coalesced spill traffic)
                "fetch_size_kib": got["FETCH_SIZE"][0], "write_size_kib": got["WRITE_SIZE"][0],
                "traffic_bytes_per_launch": (2.0 * got["FETCH_SIZE"][0] + got["WRITE_SIZE"][0]) * 1024.0,
                "other_kernels": {fam: {"launches_profiled": o["SQ_INSTS_VALU"][1], "valu_instructions_per_launch": o["SQ_INSTS_VALU"][0],
                                        "busy_cu_cycles_per_launch": o["SQ_BUSY_CU_CYCLES"][0] / 256.0, "salu_instructions_per_launch": o.get("SQ_INSTS_SALU", (None, 0))[0],
                                        "lane_utilisation": o["SQ_THREAD_CYCLES_VALU"][0] / (64.0 * o["SQ_ACTIVE_INST_VALU"][0]) if o["SQ_ACTIVE_INST_VALU"][0] else None,
                                        "tcp_lane_accesses_per_launch": o["TCP_TOTAL_CACHE_ACCESSES_sum"][0],
                                        "traffic_bytes_per_launch": (2.0 * o["FETCH_SIZE"][0] + o["WRITE_SIZE"][0]) * 1024.0}
                                  for fam, o in other.items() if "SQ_INSTS_VALU" in o and "FETCH_SIZE" in o and "WRITE_SIZE" in o and "TCP_TOTAL_CACHE_ACCESSES_sum" in o}}
    except KeyError:
        return None


def committed_pmc(args, pass_sizes, queries_per_launch):
    """Fallback when no live pass can be taken (N > 1: rocprofv3 cannot wrap one rank of a torchrun job from inside it): the
    committed single-GPU passes of the same per-GPU workload.  The per-launch counters are used as they are when the pass sizes
    match; otherwise the walk's VALU instructions are DERIVED as the committed instructions per closest-hit query x the queries this
    run counted per launch (labelled `derived`), and the fabric bytes are left out."""
    try:
        with open(os.path.join(ROOT, "profiles", PMC_ROUND, "pmc_bench.json")) as f:
            d = json.load(f)
        key = d.get("config_key")
        if key[:1] != [args.workload] or key[4] != args.rows:            # (the instructions per query are the scene's, to a few per cent whatever the ray count)
            return None
        p = dict(d["pmc"])
        if key == [args.workload, args.scanlines, args.scanlines_total, args.rays, args.rows, args.gpus, pass_sizes]:
            p["source"] = "file: profiles/%s/pmc_bench.json (%s)" % (PMC_ROUND, d.get("taken_at", "?"))
            return p
        per_query = d["valu_instructions_per_query"]
        return {"source": "derived: profiles/" + PMC_ROUND + "/pmc_bench.json instructions per closest-hit query (%.1f, N = 1) x the %.0f queries per launch counted in this run" % (per_query, queries_per_launch),
                "derived": True, "kernel": p.get("kernel"), "valu_instructions_per_launch": per_query * queries_per_launch,
                "lane_utilisation": p.get("lane_utilisation"), "traffic_bytes_per_launch": None}
    except Exception:
        return None


# ------------------------------------------------------------------------------------------------ CPU baseline + parity
def usable_cores():
    """host cores this process may actually use: the affinity mask, cut by the cgroup CPU quota (os.cpu_count() reports the
    machine's threads, which a container's quota can be far below)"""
    n = os.cpu_count() or 1
    info = {"cpu_count": n}
    try:
        n = min(n, len(os.sched_getaffinity(0))); info["affinity"] = len(os.sched_getaffinity(0))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    q = float(txt[0]) / float(txt[1]); info["cgroup_quota_cpus"] = q; n = max(1, min(n, int(q + 0.5)))
            else:
                q = float(txt[0])
                if q > 0:
                    per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()); info["cgroup_quota_cpus"] = q / per; n = max(1, min(n, int(q / per + 0.5)))
            break
        except Exception:
            continue
    return n, info


def cpu_baseline(m, sd, tr, ctx, S, R, rf0):
    """The oracle (a port of the reference algorithm; the reference binary itself needs Bullet + OpenCV and cannot be
    built) timed on this box's host cores on a bounded sample of the same workload: whole frames, tasks = (scan-line x block
    of samples) so that every core has work, walking the same BVH as the GPU; at least 5 s.  Its frame 0 doubles as the
    parity check of the timed workload: the GPU's frame-0 RF image (fixed-point contract) must equal it bit for bit."""
    import numpy as np
    from oracle import orc
    cores, core_info = usable_cores()
    nodes, btri, _ = ctx.get_bvh()
    osc = orc.OracleScene(sd.tri, sd.tri_mesh, sd.meshes, sd.materials, sd.start_mat, sd.spacing, bvh=(nodes, btri))
    osc.set_bvh4(ctx.get_bvh4()[0])
    tex = orc.texture(256)
    E = tr.n_elements
    p = orc.default_params(n_elements=E, n_samples=S, n_rows=R)
    kw = dict(use_bvh=2, n_threads=cores, want_hits=False, want_ref=False)
    osc.trace_frame(p, tr.pos, tr.dir, tex, frame_id=99, e_begin=0, e_end=min(E, 8), **kw)   # untimed: thread pool, page faults
    t0 = time.perf_counter(); c0 = sum(os.times()[:2])
    frames, o0 = 0, None
    while True:
        o = osc.trace_frame(p, tr.pos, tr.dir, tex, frame_id=frames, **kw)
        if frames == 0:
            o0 = o
        frames += 1
        dt = time.perf_counter() - t0
        if dt >= 5.0 or frames >= 100000:
            break
    busy = (sum(os.times()[:2]) - c0) / dt                      # CPU seconds per wall second: the cores the sample really kept busy
    want = o0["rf"]                                             # [R][E]
    got = np.ascontiguousarray(rf0.T)
    parity = {"rf_bit_exact": bool(np.array_equal(got.view(np.uint32), want.view(np.uint32))), "frame": 0, "scan_lines": int(E),
              "paths": int(E * S), "what": "fixed-point RF image of frame 0 of the timed workload (traced in the timed pass size), GPU vs oracle, before the PSF"}
    # one thread, how the reference itself runs (scene.cpp:74 has its OpenMP pragma commented out)
    n1 = min(E, 8)
    t1 = time.perf_counter()
    osc.trace_frame(p, tr.pos, tr.dir, tex, frame_id=0, e_begin=0, e_end=n1, use_bvh=2, n_threads=1, want_hits=False, want_ref=False)
    dt1 = time.perf_counter() - t1
    base = {"value": E * S * frames / dt, "unit": "rays/s", "cores": cores, "kind": "port", "host": core_info, "cores_kept_busy": round(busy, 1),
            "single_thread": {"value": n1 * S / dt1, "unit": "rays/s", "cores": 1, "sample": "%d scan-lines x %d rays, one thread" % (n1, S), "seconds": dt1},
            "sample": "%d whole frame(s) of %dx%d rays, same workload, %d OpenMP threads (trace + accumulate, no PSF)" % (frames, E, S, cores),
            "seconds": dt}
    return base, parity


def shard_parity(m, sd, tr, ctx, S, R, rf0, e0, e1):
    """rank 0's shard [e0, e1) of frame 0 against the oracle, bit for bit (the N > 1 line's parity evidence; no timing)"""
    import numpy as np
    from oracle import orc
    cores, _ = usable_cores()
    nodes, btri, _ = ctx.get_bvh()
    osc = orc.OracleScene(sd.tri, sd.tri_mesh, sd.meshes, sd.materials, sd.start_mat, sd.spacing, bvh=(nodes, btri))
    osc.set_bvh4(ctx.get_bvh4()[0])
    p = orc.default_params(n_elements=tr.n_elements, n_samples=S, n_rows=R)
    t0 = time.perf_counter()
    o = osc.trace_frame(p, tr.pos, tr.dir, orc.texture(256), frame_id=0, e_begin=e0, e_end=e1, use_bvh=2, n_threads=cores, want_hits=False, want_ref=False)
    want = o["rf"]                                               # [R][e1 - e0]
    got = np.ascontiguousarray(rf0.T)
    return {"rf_bit_exact": bool(want.shape == got.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32))), "frame": 0, "rank": 0, "scan_lines": [int(e0), int(e1)],
            "paths": int((e1 - e0) * S), "oracle_seconds": time.perf_counter() - t0,
            "what": "fixed-point RF image of rank 0's scan-line shard of frame 0 (traced in the timed pass size), GPU vs oracle, before the gather and the PSF"}


if __name__ == "__main__":
    main()
