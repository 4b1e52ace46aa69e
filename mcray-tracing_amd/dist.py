"""Multi-GPU sharding of a frame: scan-lines (RF columns) are independent (main.cpp:128,139 use ray_i as the column), so
rank g traces the contiguous block [g*E/G, (g+1)*E/G) with the scene replicated, and ONE collective -- an all-gather of the
[E/G][R] float blocks (RCCL over xGMI; `nccl` backend on ROCm) -- reassembles the scan-line-major image.  The PSF's lateral
pass needs 12 columns of halo (rfimage.h:113-118), so convolution runs on the gathered image."""
import torch

_GATHER_TO_ROOT_OK = True      # cleared when the backend refuses dist.gather (every rank then falls back to the all-gather, in the same call)


def shard_range(rank, world, n_elements):
    """contiguous scan-line block of `rank`; the last ranks take one fewer when E % world != 0"""
    base, rem = divmod(n_elements, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def gather_rf(rf_local, n_elements, n_rows, dist=None, group=None, root=None):
    """rf_local: this rank's scan-line block, [ne_local][R] for one frame or [F][ne_local][R] for the F frames of a pass
    (tensor on the backend's device).  Returns the full image(s), [E][R] or [F][E][R] -- ONE collective whatever F is, enqueued on
    the CURRENT torch stream.  root=None: an all-gather, every rank gets the frames; root=r: a gather to rank r only (what the
    B-mode pipeline needs: one rank post-processes), the other ranks get None and send their block once instead of receiving
    world - 1 of them."""
    if dist is None or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return rf_local
    world = dist.get_world_size(group)
    if dist.get_backend(group) == "gloo" and rf_local.is_cuda:      # plumbing checks only: stage through the host
        full = gather_rf(rf_local.cpu(), n_elements, n_rows, dist, group, root)
        return None if full is None else full.to(rf_local.device)
    batched = rf_local.dim() == 3
    loc = rf_local if batched else rf_local.unsqueeze(0)            # [F][ne][R]
    F = loc.shape[0]
    sizes = [shard_range(r, world, n_elements) for r in range(world)]
    ne_max = max(e - b for b, e in sizes)
    if n_elements % world == 0:
        mine = loc.contiguous()
    else:                                                           # ragged shards: pad to the largest block
        mine = torch.zeros((F, ne_max, n_rows), dtype=loc.dtype, device=loc.device)
        mine[:, : loc.shape[1]] = loc
    global _GATHER_TO_ROOT_OK
    stacked = None
    if root is not None and _GATHER_TO_ROOT_OK:
        me = dist.get_rank(group)
        flat = torch.empty((world, F, ne_max, n_rows), dtype=loc.dtype, device=loc.device) if me == root else None
        try:
            dist.gather(mine, list(flat.unbind(0)) if me == root else None, dst=dist.get_global_rank(group, root) if group is not None else root, group=group)
        except (RuntimeError, NotImplementedError, ValueError) as ex:
            # a backend without gather raises on EVERY rank before anything is sent: all of them take the all-gather below, now and from here on
            _GATHER_TO_ROOT_OK = False
            import sys
            sys.stderr.write("mcray_tracing_amd.dist: dist.gather refused (%s): falling back to all_gather_into_tensor\n" % str(ex).splitlines()[0])
        else:
            if me != root:
                return None
            stacked = flat
    if stacked is None:
        flat = torch.empty((world * F, ne_max, n_rows), dtype=loc.dtype, device=loc.device)      # rank blocks concatenated along dim 0
        dist.all_gather_into_tensor(flat, mine, group=group)
        stacked = flat.view(world, F, ne_max, n_rows)
        if root is not None and dist.get_rank(group) != root:
            return None
    if n_elements % world == 0:
        full = stacked.permute(1, 0, 2, 3).reshape(F, n_elements, n_rows)      # frame-major, rank blocks concatenated (a copy)
    else:
        full = torch.cat([stacked[r, :, : e - b] for r, (b, e) in enumerate(sizes)], dim=1)
    full = full.contiguous()
    return full if batched else full[0]
