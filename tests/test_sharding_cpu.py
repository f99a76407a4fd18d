"""world_size-2 gloo test of the multi-GPU path's host logic (no GPU): contiguous sharding of the batch
and the single all-gather of solved trajectories + status words that bench.py / callers use."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import lc


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, total, nx, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sh = lc("sharding")
    lo, hi = sh.shard_range(total, world, rank)
    # every rank "solves" its shard: x*[i] = member index, status = index % 3
    idx = torch.arange(lo, hi, dtype=torch.float64)
    x = idx[:, None].repeat(1, nx)
    st = (torch.arange(lo, hi) % 3).to(torch.int32)
    X, S = sh.gather_solutions(x, st)
    n = sh.solved_count(st)
    q.put((rank, lo, hi, X.numpy(), S.numpy(), n))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_gather_world2():
    world, total, nx = 2, 16, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, total, nx, q)) for r in range(world)]
    for p in ps: p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps: p.join(60)
    want_x = np.arange(total, dtype=float)[:, None].repeat(nx, 1)
    want_s = (np.arange(total) % 3).astype(np.int32)
    assert sorted((r[1], r[2]) for r in res) == [(0, 8), (8, 16)]
    for r in res:
        assert np.array_equal(r[3], want_x) and np.array_equal(r[4], want_s)
        assert r[5] == float((want_s == 0).sum())


def test_shard_and_gather_world2_ragged():
    """7 members over 2 ranks (4 + 3): the gather must pad, exchange once and drop the padding rows"""
    world, total, nx = 2, 7, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, total, nx, q)) for r in range(world)]
    for p in ps: p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps: p.join(60)
    want_x = np.arange(total, dtype=float)[:, None].repeat(nx, 1)
    want_s = (np.arange(total) % 3).astype(np.int32)
    assert sorted((r[1], r[2]) for r in res) == [(0, 4), (4, 7)]
    for r in res:
        assert np.array_equal(r[3], want_x) and np.array_equal(r[4], want_s)


def test_shard_range_covers_ragged_totals():
    sh = lc("sharding")
    for total in (0, 1, 7, 8192, 1000):
        for world in (1, 2, 3, 8):
            spans = [sh.shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
