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
    # the execution plan travels the same way (rank 0's tile picks; here: the host-side default plan with two picks written in, as an autotuner would)
    mine = None
    if r == 0:
        mine = g.defaultPlan(4)
        mine[3 + 1], mine[3 + 2] = 7, 13
    plan = sd.broadcast_plan(mine, device=torch.device("cpu"))
    q.put((r, sd.arena_checksum(arena), int(arena.numel()), calls["pack"], (lo, hi), plan.tobytes()))
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
    (r0, c0, n0, packs0, s0, p0), (r1, c1, n1, packs1, s1, p1) = res
    assert p0 == p1 and len(p0) > 12                   # the plan bytes are identical on both ranks ...
    words = np.frombuffer(p0, np.int32)
    assert words[0] == 0x4e414c50 and 0 < words[2] < len(words) - 3 and words[4] == 7 and words[5] == 13   # ... and are rank 0's (header, one word per step, the persistent runs, a fingerprint)
    assert (r0, r1) == (0, 1)
    assert n0 == n1 > 5e6 and c0 == c1                 # byte-identical weights on both ranks
    assert packs0 == 1 and packs1 == 0                 # only the source rank runs the CPU weight preparation
    assert s0 == (0, 5) and s1 == (5, 10)
    ref = ONNXGraph(zoo.mobilenet_v2(hw=(32, 32))).packWeights(4)
    assert sd.arena_checksum(torch.from_numpy(ref)) == c0


def test_default_plan_is_host_only_and_self_describing():
    from smelter_amd import ops
    g = ONNXGraph(zoo.resnet50(hw=(32, 32)))
    p = g.defaultPlan(8)
    assert p.dtype == np.int32 and p[0] == 0x4e414c50 and p[1] == ops.num_conv_configs()
    steps, runs = int(p[2]), int(p[3 + p[2]])
    assert len(p) == 3 + steps + 1 + 3 * runs + 1      # header, one word per step, the persistent runs (first step, steps, on), the step list's fingerprint
    assert (p[3:3 + steps] == -1).all()                 # no autotuning happened: every step on its heuristic
    assert np.array_equal(sd.broadcast_plan(p), p)      # no process group: the plan comes back as it is


def test_single_process_path_needs_no_process_group():
    g = ONNXGraph(zoo.toy())
    arena = sd.broadcast_weight_arena(g, 1, device=torch.device("cpu"))
    assert np.array_equal(arena.numpy(), g.packWeights(1))
