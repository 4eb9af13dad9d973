"""Batch-shard data parallelism: one process per GPU, weights broadcast ONCE over RCCL/xGMI, no per-iteration
collective (BASELINE.json north_star).  The reference has no multi-device code at all
(/root/reference/Sources/Smelter/ONNXGraph.swift:169 takes a single MTLDevice); inference over a batch has no
cross-sample dependency, so the batch axis shards into independent contiguous image ranges.

Flow:  every rank parses the ONNX bytes (cheap, structure only) -> rank 0 alone does the weight preparation
(BN fold, fp16 cast, OHWI, packing) into one arena -> `torch.distributed.broadcast` of that arena
(backend "nccl" == RCCL on ROCm; "gloo" in the CPU tests) -> every rank binds the received arena.
"""
from __future__ import annotations

import os
from typing import Tuple

import numpy as np


def shard_range(global_batch: int, rank: int, world: int) -> Tuple[int, int]:
    """contiguous image range [lo, hi) of `rank`; remainders go to the lowest ranks"""
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def init_from_env(backend: str = None):
    """torch.distributed init from RANK / WORLD_SIZE / MASTER_* (torchrun contract).  Returns (rank, world, local)."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if os.environ.get("SMELTER_FORCE_LOCAL0"):        # dry run of the multi-rank path on a one-GPU box: every rank uses device 0
        local = 0
    backend = backend or os.environ.get("SMELTER_DIST_BACKEND")
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def broadcast_weight_arena(graph, batch: int, device=None, src: int = 0):
    """Returns a uint8 tensor holding the packed weight arena on every rank.  Only `src` runs the CPU weight
    preparation; the others receive the bytes through one broadcast."""
    import torch
    import torch.distributed as dist
    nbytes = graph.weightBytes(batch)
    rank = dist.get_rank() if dist.is_initialized() else 0
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    if rank == src:
        arena = torch.from_numpy(graph.packWeights(batch)).to(device)
    else:
        arena = torch.empty(nbytes, dtype=torch.uint8, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(arena, src=src)
    return arena


def broadcast_plan(plan, src: int = 0, device=None):
    """Every rank returns rank `src`'s execution plan (int32 numpy array; `plan` is ignored elsewhere and may be None).  Two small broadcasts: length, words."""
    import torch
    import torch.distributed as dist
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return np.asarray(plan, np.int32)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    rank = dist.get_rank()
    n = torch.tensor([len(plan) if rank == src else 0], dtype=torch.int64, device=device)
    dist.broadcast(n, src=src)
    t = torch.from_numpy(np.ascontiguousarray(np.asarray(plan, np.int32))).to(device) if rank == src else torch.empty(int(n.item()), dtype=torch.int32, device=device)
    dist.broadcast(t, src=src)
    return t.cpu().numpy()


def arena_checksum(arena) -> int:
    """byte-exact checksum used by tests to show every rank holds identical weights"""
    import zlib
    a = np.ascontiguousarray(arena.detach().cpu().numpy())
    return (zlib.crc32(memoryview(a).cast("B")) << 32) | (zlib.adler32(memoryview(a).cast("B")) & 0xFFFFFFFF)   # no temporaries: the arena is 51 MB


def compile_sharded(graph, per_rank_batch: int, local_device: int = 0, src: int = 0):
    """rank-local executor bound to the broadcast arena (no weight upload on non-source ranks).  Rank `src` compiles first — autotuning if the graph's
    Configuration asks for it — and its execution plan is broadcast: every rank runs the identical kernels with the identical tile configurations, so a
    shard's outputs are bit-identical to the one-GPU run of the same images."""
    import torch.distributed as dist
    if not dist.is_initialized():
        src = 0                                       # one process: it is the source whatever the caller named
    arena = broadcast_weight_arena(graph, per_rank_batch, src=src)
    rank = dist.get_rank() if dist.is_initialized() else 0
    ex = graph.hipGraph(local_device, per_rank_batch, weight_arena=arena, upload_weights=False) if rank == src else None
    plan = broadcast_plan(ex.plan if ex is not None else None, src=src)
    if ex is None:
        ex = graph.hipGraph(local_device, per_rank_batch, weight_arena=arena, upload_weights=False, plan=plan)
    ex.shard_arena = arena                            # (what this rank's kernels read: bench.py checksums it per rank)
    return ex


def shard_fingerprint(ex) -> dict:
    """what a batch-shard job compares across ranks: the bytes of the weight arena this rank's kernels read and the execution plan it runs"""
    import zlib
    return {"arena": arena_checksum(ex.shard_arena) if getattr(ex, "shard_arena", None) is not None else None, "plan": zlib.crc32(np.ascontiguousarray(ex.plan).tobytes())}
