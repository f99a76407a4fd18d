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


def gather_solutions(x_local, status_local, out_x=None, out_status=None):
    """all-gather x* [b, nx] and status [b] of equally sized shards -> ([world*b, nx], [world*b])"""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return x_local, status_local
    world = dist.get_world_size()
    if out_x is None:
        out_x = torch.empty((world * x_local.shape[0],) + tuple(x_local.shape[1:]), dtype=x_local.dtype, device=x_local.device)
    if out_status is None:
        out_status = torch.empty((world * status_local.shape[0],), dtype=status_local.dtype, device=status_local.device)
    dist.all_gather_into_tensor(out_x, x_local.contiguous())
    dist.all_gather_into_tensor(out_status, status_local.contiguous())
    return out_x, out_status


def solved_count(status_local):
    """global number of converged members (all-reduce of a count; metric only)"""
    n = (status_local == 0).sum().to(torch.float64).reshape(1)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(n, op=dist.ReduceOp.SUM)
    return float(n.item())
