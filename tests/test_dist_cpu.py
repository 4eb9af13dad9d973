"""Batch-shard plumbing on CPU: world_size=2 over gloo (the GPU path uses the same code over RCCL).
Covers shard ranges, the one-time weight-arena broadcast (only rank 0 prepares weights), and that no data-path
collective exists after it."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from smelter_amd import dist as sd
from smelter_amd import zoo
from smelter_amd.graph import ONNXGraph


def test_shard_range_partitions_the_batch():
    for gb in (512, 64, 7, 1):
        for world in (1, 2, 4, 8):
            rs = [sd.shard_range(gb, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == gb
            assert all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
            sizes = [hi - lo for lo, hi in rs]
            assert max(sizes) - min(sizes) <= 1
    assert [sd.shard_range(512, r, 8) for r in (0, 7)] == [(0, 64), (448, 512)]      # BASELINE config 5


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w, _ = sd.init_from_env("gloo")
    model = zoo.mobilenet_v2(hw=(32, 32))           # every rank parses the structure (7 MB of weights: the CPU suite stays fast)
    g = ONNXGraph(model)
    calls = {"pack": 0}
    orig = g.packWeights

    def counting(b):
        calls["pack"] += 1
        return orig(b)
    g.packWeights = counting
    arena = sd.broadcast_weight_arena(g, 4, device=torch.device("cpu"))
    lo, hi = sd.shard_range(10, r, w)
    q.put((r, sd.arena_checksum(arena), int(arena.numel()), calls["pack"], (lo, hi)))
    dist.barrier()
    dist.destroy_process_group()


def test_weight_arena_broadcast_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (r0, c0, n0, packs0, s0), (r1, c1, n1, packs1, s1) = res
    assert (r0, r1) == (0, 1)
    assert n0 == n1 > 5e6 and c0 == c1                 # byte-identical weights on both ranks
    assert packs0 == 1 and packs1 == 0                 # only the source rank runs the CPU weight preparation
    assert s0 == (0, 5) and s1 == (5, 10)
    ref = ONNXGraph(zoo.mobilenet_v2(hw=(32, 32))).packWeights(4)
    assert sd.arena_checksum(torch.from_numpy(ref)) == c0


def test_single_process_path_needs_no_process_group():
    g = ONNXGraph(zoo.toy())
    arena = sd.broadcast_weight_arena(g, 1, device=torch.device("cpu"))
    assert np.array_equal(arena.numpy(), g.packWeights(1))
