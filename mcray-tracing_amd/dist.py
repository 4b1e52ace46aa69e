"""Multi-GPU sharding of a frame: scan-lines (RF columns) are independent (main.cpp:128,139 use ray_i as the column), so
rank g traces the contiguous block [g*E/G, (g+1)*E/G) with the scene replicated, and ONE collective -- an all-gather of the
[E/G][R] float blocks (RCCL over xGMI; `nccl` backend on ROCm) -- reassembles the scan-line-major image.  The PSF's lateral
pass needs 12 columns of halo (rfimage.h:113-118), so convolution runs on the gathered image."""
import torch

_GATHER_TO_ROOT = {}           # (backend name, group) -> does it gather to a root?  Decided ONCE per backend and group, by ALL its ranks together (_gather_to_root_ok)


def _unsupported(ex):
    """is this the backend saying "I have no such collective" (raised on every rank before anything is sent) -- as opposed to a failure of
    this rank, which must not be papered over: the peers would sit in a different collective"""
    if isinstance(ex, NotImplementedError):
        return True
    msg = str(ex).lower()
    return isinstance(ex, RuntimeError) and any(k in msg for k in ("not support", "unsupported", "not implemented", "does not implement"))


def _gather_to_root_ok(dist, group, like):
    """Can the group's backend gather to a root?  Probed with a 1-element gather the first time a backend is used, and the answer is the
    MINIMUM over the ranks (an all-reduce), so every rank takes the same branch from then on -- a rank-local surprise can no longer send
    one rank into an all-gather while its peers wait in a gather.  Anything but "unsupported" is re-raised -- AFTER the all-reduce, which
    every rank enters whatever its probe did (a rank that raised before it left its peers blocked there until the backend timed out)."""
    be = (dist.get_backend(group), id(group) if group is not None else None)      # per (backend, group): another group may span other devices
    if be not in _GATHER_TO_ROOT:
        me, world = dist.get_rank(group), dist.get_world_size(group)
        ok, failure = 1, None
        probe = torch.zeros(1, dtype=torch.float32, device=like.device)
        try:
            dist.gather(probe, [torch.zeros_like(probe) for _ in range(world)] if me == 0 else None,
                        dst=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        except Exception as ex:           # whatever went wrong, this rank still takes part in the all-reduce below: its peers are waiting there
            ok = 0
            if _unsupported(ex):
                import sys
                sys.stderr.write("mcray_tracing_amd.dist: %s has no gather (%s): all_gather_into_tensor instead\n" % (be[0], str(ex).splitlines()[0]))
            else:
                failure = ex
        flag = torch.tensor([ok], dtype=torch.int32, device=like.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if failure is not None:           # a failure of THIS rank is not "unsupported": it propagates, after the collective every rank entered
            raise failure
        _GATHER_TO_ROOT[be] = bool(int(flag.item()))
    return _GATHER_TO_ROOT[be]


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
    stacked = None
    if root is not None and _gather_to_root_ok(dist, group, mine):
        me = dist.get_rank(group)
        flat = torch.empty((world, F, ne_max, n_rows), dtype=loc.dtype, device=loc.device) if me == root else None
        dist.gather(mine, list(flat.unbind(0)) if me == root else None, dst=dist.get_global_rank(group, root) if group is not None else root, group=group)
        if me != root:           # (an error here is this rank's own and propagates: no silent change of collective)
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
