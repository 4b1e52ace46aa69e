#!/usr/bin/env python3
"""bench.py -- rays/sec + B-mode frames/sec of the ray-tracing hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload random1m|sphere|liver]
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one frame of the hot path over synthetic input already resident in HBM:
clear -> trace (BVH closest-hit + interface sampling) -> RF accumulation -> [RCCL all-gather of the
scan-line blocks when N > 1] -> PSF convolution.  Each rank traces 128 scan-lines x 1024 sample paths
(weak scaling: the frame has 128*N scan-lines).  By default 32 consecutive frames are in flight per
pass (mcrt_trace_frames: every launch carries 32 frames' rays; images are bit-identical to
one-at-a-time tracing); `--frames-in-flight 1` is the strict latency mode, also reported in the
JSON as `one_frame_at_a_time`.  The JSON line carries the live roofline figure of the
dominant kernel (k_trace: counted algorithmic bytes / HIP-event kernel time) and a CPU baseline (the
oracle = port of the reference algorithm, timed on this box's host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def build_workload(m, name):
    if name == "random1m":
        cfg, meshes = m.synth.random_scene(1_000_000, 8, 12345)
        label = "synthetic 1M random triangles (8 meshes, PCG64 seed 12345)"
    elif name == "sphere":
        cfg, meshes = m.synth.sphere_scene(5)
        label = "examples/sphere (generated icosphere 20480 tris + box)"
    elif name == "liver":
        cfg, meshes = m.synth.liver_scene(5)
        label = "ircad11-like synthetic liver scene (11 procedural organs, ~225k tris)"
    else:
        raise SystemExit("unknown workload " + name)
    return cfg, m.scene_io.build_scene(cfg, meshes), label


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--workload", default="random1m")
    ap.add_argument("--scanlines", type=int, default=128, help="scan-lines per GPU")
    ap.add_argument("--rays", type=int, default=1024, help="sample paths per scan-line")
    ap.add_argument("--rows", type=int, default=465)
    ap.add_argument("--tex-n", type=int, default=256, help="texture edge in voxels (256 = the reference; smaller only for cache experiments)")
    ap.add_argument("--frames-in-flight", type=int, default=32,
                    help="frames traced per pass (mcrt_trace_frames): a step is still ONE frame, but every kernel launch then carries the "
                         "rays of this many consecutive frames (1 = strict one-frame-at-a-time latency mode)")
    ap.add_argument("--bvh", default="sah", choices=["sah", "lbvh"], help="BVH builder: host binned SAH (default) or the device LBVH")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency-leg", action="store_true", help="skip the extra one-frame-at-a-time measurement (profiling runs)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL); gloo only for plumbing checks")
    ap.add_argument("--same-gpu", action="store_true", help="plumbing check on a 1-GPU box: every rank uses GPU 0")
    args = ap.parse_args()

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

    E_local, S, R = args.scanlines, args.rays, args.rows
    E = E_local * world
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
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)

    from mcray_tracing_amd.dist import shard_range
    assert E % world == 0
    e0, e1 = shard_range(rank, world, E)
    F = max(1, args.frames_in_flight)
    rf_local = torch.zeros((F, E_local, R), dtype=torch.float32, device="cuda")

    def step_batch(frame, nf):
        """nf consecutive frames (nf = 1 unless --frames-in-flight): trace -> gather -> PSF convolution of each frame"""
        ctx.trace_frames(frame, nf, rf_local, e0, e1)
        if world > 1:
            # ONE RCCL all-gather over xGMI per pass: every rank contributes its [nf][E/N][R] block; frame f of the result is the
            # concatenation of the ranks' scan-line blocks, made contiguous for the convolution
            gathered = torch.empty((world, nf, E_local, R), dtype=torch.float32, device="cuda")
            if args.backend == "nccl":
                dist.all_gather_into_tensor(gathered, rf_local[:nf].contiguous())
            else:
                parts = [torch.empty((nf, E_local, R)) for _ in range(world)]
                dist.all_gather(parts, rf_local[:nf].cpu())
                gathered = torch.stack(parts).cuda()
            frames = gathered.permute(1, 0, 2, 3).reshape(nf, E, R).contiguous()
        else:
            frames = rf_local
        if rank == 0:
            ctx.convolve_frames(frames, nf, E, R, psf.axial_kernel, psf.lateral_kernel)      # all nf images in one launch per pass

    def passes(count):
        """split `count` frames into the fewest passes of at most F frames, as even as possible (20 with F=16 -> 10 + 10)"""
        if count <= 0:
            return []
        n_pass = -(-count // F)
        base, rem = divmod(count, n_pass)
        return [base + (1 if i < rem else 0) for i in range(n_pass)]

    def run_steps(first, count):
        f = first
        for nf in passes(count):
            step_batch(f, nf)
            f += nf

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- counted algorithmic bytes of the timed frames (instrumented build of the same kernel, untimed) ----
    # (the counting build walks every ray once, i.e. exactly the oracle's node/triangle visits: the timed kernel may cut the
    #  rays of a small bounce into pieces, whose extra visits are overhead, not algorithmic bytes)
    ctx.enable_stats(True); ctx.get_stats(reset=True)
    f = 0
    for nf in passes(args.steps):
        ctx.trace_frames(f, nf, rf_local, e0, e1)
        f += nf
    st = ctx.get_stats(reset=True)
    ctx.enable_stats(False)
    # Algorithmic bytes (adapted to the 128-B BVH4 nodes).  The dominant kernel is k_trace, launched once per
    # bounce: per closest-hit query nodes*128 B + triangles*48 B + the 32-B ray read and 32-B hit record written.
    trace_bytes_frame = (st["nodes_visited"] * 128 + st["tris_tested"] * 48 + st["queries"] * 64) / args.steps
    # the rest of the frame, for the record: 64-B segment written + read, 8-B texture gather per RF step, RF block + bins
    other_bytes_frame = (st["segments"] * 128 + st["rf_steps"] * 8) / args.steps + E_local * R * (4 + 8)

    run_steps(1000, args.warmup)
    ctx.enable_timing(True); ctx.kernel_time(reset=True)
    sync()
    t0 = time.perf_counter()
    run_steps(0, args.steps)
    sync()
    dt = time.perf_counter() - t0
    k_ms, k_n = ctx.kernel_time(reset=True)
    ctx.enable_timing(False)
    launches_per_frame = k_n / args.steps                   # max_depth bounces (x groups) / frames per pass
    alg_bytes = trace_bytes_frame / launches_per_frame

    dt_t = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(dt_t, op=dist.ReduceOp.MAX)
    dt = float(dt_t.item())

    if rank == 0:
        rays = E * S * args.steps
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        out = {
            "metric": "rays/sec (Monte-Carlo sample paths traced + accumulated + PSF-convolved per second)",
            "value": rays / dt, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "frames_per_sec": args.steps / dt, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s; %d scan-lines x %d rays per GPU, %d RF rows, max depth 10" % (label, E_local, S, R),
                       "scan_lines_total": E, "rays_per_scan_line": S, "triangles": int(sd.n_tri), "parallelism": "scanline-shard x%d" % world, "frames_in_flight": F,
                       "bvh_builder": args.bvh, "bvh_build_s": round(t_bvh, 3)},
            "roofline": {"bound": "hbm", "kernel": "k_trace", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(args),
                         "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": k_ms, "launches": k_n,
                         "launches_per_frame": launches_per_frame, "trace_bytes_per_frame": trace_bytes_frame,
                         "other_stage_bytes_per_frame": other_bytes_frame,
                         "per_launch": {k: v / args.steps for k, v in st.items()}},
        }
        if world == 1 and F > 1 and not args.no_latency_leg:
            # the same workload strictly one frame at a time (each launch carries one frame's rays), for the record
            for f in range(args.warmup):
                step_batch(2000 + f, 1)
            sync(); t1 = time.perf_counter()
            for f in range(args.steps):
                step_batch(f, 1)
            sync(); dt1 = time.perf_counter() - t1
            out["one_frame_at_a_time"] = {"value": E * S * args.steps / dt1, "unit": "rays/s", "ms_per_step": dt1 / args.steps * 1e3,
                                          "frames_per_sec": args.steps / dt1}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(m, sd, tr, ctx, S, R)
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def pmc_traffic(args):
    """HBM-side bytes per k_trace launch from the committed rocprofv3 PMC passes of this same command
    (profiles/round1/pmc_k_trace.json: (2 x FETCH_SIZE + WRITE_SIZE) KiB, gfx950 correction applied); null for other workloads."""
    if (args.workload, args.scanlines, args.rays, args.rows, args.gpus, args.frames_in_flight) != ("random1m", 128, 1024, 465, 1, 32):
        return None
    try:
        with open(os.path.join(ROOT, "profiles", "round1", "pmc_k_trace.json")) as f:
            return json.load(f)["derived"]["traffic_bytes_per_k_trace_launch"]
    except Exception:
        return None


def cpu_baseline(m, sd, tr, ctx, S, R):
    """The oracle (a port of the reference algorithm; the reference binary itself needs Bullet + OpenCV and cannot be
    built) timed on this box's host cores on a bounded sample of the same workload: the first scan-lines of frame 0,
    OpenMP over scan-lines, walking the same BVH as the GPU."""
    from oracle import orc
    cores = os.cpu_count() or 1
    nodes, btri, _ = ctx.get_bvh()
    osc = orc.OracleScene(sd.tri, sd.tri_mesh, sd.meshes, sd.materials, sd.start_mat, sd.spacing, bvh=(nodes, btri))
    osc.set_bvh4(ctx.get_bvh4()[0])
    tex = orc.texture(256)
    n_el = min(tr.n_elements, max(cores, 8))
    p = orc.default_params(n_elements=tr.n_elements, n_samples=S, n_rows=R)
    osc.trace_frame(p, tr.pos, tr.dir, tex, frame_id=99, e_begin=0, e_end=min(n_el, 16), use_bvh=2, n_threads=cores, want_hits=False)   # untimed: thread pool, page faults
    t0 = time.perf_counter()
    osc.trace_frame(p, tr.pos, tr.dir, tex, frame_id=0, e_begin=0, e_end=n_el, use_bvh=2, n_threads=cores, want_hits=False)
    dt = time.perf_counter() - t0
    # keep the sample between ~10 and 30 s of CPU work
    reps = 1
    while dt * cores * reps < 10.0 and reps < 64:
        reps *= 2
    if reps > 1:
        t0 = time.perf_counter()
        for i in range(reps):
            osc.trace_frame(p, tr.pos, tr.dir, tex, frame_id=i, e_begin=0, e_end=n_el, use_bvh=2, n_threads=cores, want_hits=False)
        dt = time.perf_counter() - t0
    # (a) of BASELINE.md's plan: one thread, how the reference itself runs (scene.cpp:74 has its OpenMP pragma commented out)
    n1 = min(tr.n_elements, 16)
    t0 = time.perf_counter()
    osc.trace_frame(p, tr.pos, tr.dir, tex, frame_id=0, e_begin=0, e_end=n1, use_bvh=2, n_threads=1, want_hits=False)
    dt1 = time.perf_counter() - t0
    return {"value": n_el * S * reps / dt, "unit": "rays/s", "cores": cores, "kind": "port",
            "single_thread": {"value": n1 * S / dt1, "unit": "rays/s", "cores": 1, "sample": "%d scan-lines x %d rays, one thread" % (n1, S), "seconds": dt1},
            "sample": "%d scan-lines x %d rays x %d frame(s) of the same workload, OpenMP over scan-lines (trace + RF accumulation, no PSF)" % (n_el, S, reps),
            "seconds": dt}


if __name__ == "__main__":
    main()
