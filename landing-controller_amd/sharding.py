"""Batch sharding of the drop-state sweep over the GPUs of one node.

Batch members are independent NLPs (the reference solves them in a serial ``for``,
generate_data/generate_training_data_automated.m:38), so the path shards with no data-path collective:
rank r solves the contiguous members [r*B/G, (r+1)*B/G).  The only communication is one all-gather of
the solved trajectories and status words (RCCL over xGMI when the backend is "nccl"; the same code runs
on "gloo" for the CPU tests).
"""
import torch
import torch.distributed as dist


def shard_range(total, world, rank):
    """contiguous split; the first ``total % world`` ranks get one extra member"""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_solutions(x_local, status_local, out_x=None, out_status=None, equal_shards=None):
    """all-gather x* [b_r, nx] and status [b_r] of the ranks' shards -> ([sum b_r, nx], [sum b_r]) in rank order.

    Shards produced by shard_range() may be ragged (the first ``total % world`` ranks hold one more member).
    ``equal_shards=True`` promises equal sizes and skips the size exchange (bench.py: 1024 members per GPU);
    otherwise the sizes are all-gathered first, every shard is padded to the largest one for the single
    all_gather_into_tensor, and the padding rows are dropped afterwards."""
    if not (dist.is_available() and dist.is_initialized()):
        return x_local, status_local
    if dist.get_world_size() == 1 and out_x is None:      # (with output buffers a single rank still goes through the collective:
        return x_local, status_local                       # bench.py --force-dist, the 1-GPU smoke test of the RCCL path)
    world = dist.get_world_size()
    b = x_local.shape[0]
    if equal_shards is None:
        equal_shards = out_x is not None and out_x.shape[0] == world * b
    if equal_shards:
        if out_x is None:
            out_x = torch.empty((world * b,) + tuple(x_local.shape[1:]), dtype=x_local.dtype, device=x_local.device)
        if out_status is None:
            out_status = torch.empty((world * b,), dtype=status_local.dtype, device=status_local.device)
        dist.all_gather_into_tensor(out_x, x_local.contiguous())
        dist.all_gather_into_tensor(out_status, status_local.contiguous())
        return out_x, out_status
    sizes = torch.zeros(world, dtype=torch.int64, device=x_local.device)
    dist.all_gather_into_tensor(sizes, torch.tensor([b], dtype=torch.int64, device=x_local.device))
    sizes = [int(v) for v in sizes.tolist()]
    bmax = max(sizes)
    xp = torch.zeros((bmax,) + tuple(x_local.shape[1:]), dtype=x_local.dtype, device=x_local.device)
    sp = torch.full((bmax,), -1, dtype=status_local.dtype, device=status_local.device)      # -1 = padding row
    xp[:b] = x_local; sp[:b] = status_local
    gx = torch.empty((world * bmax,) + tuple(x_local.shape[1:]), dtype=x_local.dtype, device=x_local.device)
    gs = torch.empty((world * bmax,), dtype=status_local.dtype, device=status_local.device)
    dist.all_gather_into_tensor(gx, xp)
    dist.all_gather_into_tensor(gs, sp)
    if all(n == bmax for n in sizes):
        return gx, gs
    keep = torch.cat([torch.arange(r * bmax, r * bmax + n, device=x_local.device) for r, n in enumerate(sizes)])
    return gx.index_select(0, keep), gs.index_select(0, keep)


def solved_count(status_local):
    """global number of converged members (all-reduce of a count; metric only)"""
    n = (status_local == 0).sum().to(torch.float64).reshape(1)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(n, op=dist.ReduceOp.SUM)
    return float(n.item())
