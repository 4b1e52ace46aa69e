"""The N>1 path on CPU: world_size-2 `gloo` processes shard the scan-lines exactly as bench.py does on GPUs (same
shard_range / gather_rf helpers), with the oracle standing in for the kernel, and must reproduce the single-process frame
bit for bit -- including the PSF convolution, envelope and scan conversion applied to the gathered image on rank 0."""
import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, E, S, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import mcray_tracing_amd as m
    from mcray_tracing_amd.dist import shard_range, gather_rf
    from oracle import orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg, meshes = m.synth.sphere_scene(2)
    sd = m.scene_io.build_scene(cfg, meshes)
    tr = m.Transducer(E, position=cfg["transducerPosition"], angles_deg=cfg["transducerAngles"])
    tex = orc.texture(16)
    osc = orc.OracleScene(sd.tri, sd.tri_mesh, sd.meshes, sd.materials, sd.start_mat, sd.spacing)
    p = orc.default_params(n_elements=E, n_samples=S, tex_n=16)
    e0, e1 = shard_range(rank, world, E)
    o = osc.trace_frame(p, tr.pos, tr.dir, tex, frame_id=2, e_begin=e0, e_end=e1, want_hits=False, want_ref=False)
    local = torch.from_numpy(np.ascontiguousarray(o["rf"].T))          # [ne][R], the device layout
    full = gather_rf(local, E, p.n_rows, dist)
    # the F frames of a pass go through ONE collective (what bench.py does): frames 2 and 3 as [F][ne][R]
    o3 = osc.trace_frame(p, tr.pos, tr.dir, tex, frame_id=3, e_begin=e0, e_end=e1, want_hits=False, want_ref=False)
    both = gather_rf(torch.stack([local, torch.from_numpy(np.ascontiguousarray(o3["rf"].T))]), E, p.n_rows, dist)
    assert both.shape == (2, E, p.n_rows) and torch.equal(both[0], full)
    # gathered to ONE rank (what bench.py does for N > 1: only rank 0 post-processes): the same frames there, nothing elsewhere
    for root in (0, 1):
        at_root = gather_rf(torch.stack([local, torch.from_numpy(np.ascontiguousarray(o3["rf"].T))]), E, p.n_rows, dist, root=root)
        assert (at_root is None) == (rank != root)
        if rank == root:
            assert torch.equal(at_root, both)
    if rank == 0:
        np.save(out_path, full.numpy())
        np.save(out_path + ".f3.npy", both[1].numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("E", [8, 7])
def test_two_rank_scanline_sharding_matches_single_process(tmp_path, orc, mcrt, E):
    import torch.multiprocessing as mp
    S = 16
    out = str(tmp_path / "gathered.npy")
    port = 29500 + (os.getpid() % 2000) + E
    mp.spawn(_worker, args=(2, port, E, S, out), nprocs=2, join=True)
    got = np.load(out)                                                   # [E][R]
    cfg, meshes = mcrt.synth.sphere_scene(2)
    sd = mcrt.scene_io.build_scene(cfg, meshes)
    tr = mcrt.Transducer(E, position=cfg["transducerPosition"], angles_deg=cfg["transducerAngles"])
    tex = orc.texture(16)
    osc = orc.OracleScene(sd.tri, sd.tri_mesh, sd.meshes, sd.materials, sd.start_mat, sd.spacing)
    p = orc.default_params(n_elements=E, n_samples=S, tex_n=16)
    ref = osc.trace_frame(p, tr.pos, tr.dir, tex, frame_id=2, want_hits=False, want_ref=False)["rf"]   # [R][E]
    assert np.array_equal(got.T.view(np.uint32), ref.view(np.uint32))
    ref3 = osc.trace_frame(p, tr.pos, tr.dir, tex, frame_id=3, want_hits=False, want_ref=False)["rf"]
    assert np.array_equal(np.load(out + ".f3.npy").T.view(np.uint32), ref3.view(np.uint32))
    ax, lat = orc.psf()
    assert np.array_equal(orc.convolve(np.ascontiguousarray(got.T), ax, lat).view(np.uint32), orc.convolve(ref, ax, lat).view(np.uint32))
    # ... and the rest of the B-mode frame rank 0 computes from the gathered image (main.cpp:146-148), as bench.py's step does
    bmode = lambda img: orc.scan_convert(orc.envelope(orc.convolve(img, ax, lat)), out_rows=100, out_cols=125)
    a, b = bmode(np.ascontiguousarray(got.T)), bmode(ref)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and np.count_nonzero(a) > 100


def test_shard_range_covers_every_scanline(mcrt):
    from mcray_tracing_amd.dist import shard_range
    for E in (1, 7, 8, 128, 512, 513):
        for G in (1, 2, 3, 4, 8):
            cover = []
            for r in range(G):
                b, e = shard_range(r, G, E)
                assert 0 <= b <= e <= E
                cover += list(range(b, e))
            assert cover == list(range(E))
