"""Multi-GPU sharding of a frame: scan-lines (RF columns) are independent (main.cpp:128,139 use ray_i as the column), so
rank g traces the contiguous block [g*E/G, (g+1)*E/G) with the scene replicated, and ONE collective -- an all-gather of the
[E/G][R] float blocks (RCCL over xGMI; `nccl` backend on ROCm) -- reassembles the scan-line-major image.  The PSF's lateral
pass needs 12 columns of halo (rfimage.h:113-118), so convolution runs on the gathered image."""
import torch


def shard_range(rank, world, n_elements):
    """contiguous scan-line block of `rank`; the last ranks take one fewer when E % world != 0"""
    base, rem = divmod(n_elements, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def gather_rf(rf_local, n_elements, n_rows, dist=None, group=None):
    """rf_local: [ne_local][R] tensor (device of the backend).  Returns the full [E][R] image on every rank."""
    if dist is None or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return rf_local
    world = dist.get_world_size(group)
    if dist.get_backend(group) == "gloo" and rf_local.is_cuda:      # plumbing checks only: stage through the host
        return gather_rf(rf_local.cpu(), n_elements, n_rows, dist, group).to(rf_local.device)
    sizes = [shard_range(r, world, n_elements) for r in range(world)]
    full = torch.empty((n_elements, n_rows), dtype=rf_local.dtype, device=rf_local.device)
    if n_elements % world == 0:
        dist.all_gather_into_tensor(full, rf_local.contiguous(), group=group)
    else:
        parts = [full[b:e] for b, e in sizes]
        ne_max = max(e - b for b, e in sizes)
        bufs = [torch.empty((ne_max, n_rows), dtype=rf_local.dtype, device=rf_local.device) for _ in range(world)]
        mine = torch.zeros((ne_max, n_rows), dtype=rf_local.dtype, device=rf_local.device)
        mine[: rf_local.shape[0]] = rf_local
        dist.all_gather(bufs, mine, group=group)
        for p, b in zip(parts, bufs):
            p.copy_(b[: p.shape[0]])
    return full
