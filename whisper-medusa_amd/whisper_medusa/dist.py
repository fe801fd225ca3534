"""Data-parallel sharding of audio streams over the GPUs of one node.

The path shards naturally: every stream's encode + decode is independent (the reference itself is
batch-1, model.py:1451), so there is NO collective on the data path.  The only exchange is one-time:
rank 0 reads / packs the checkpoint and the packed blob is broadcast to the other ranks with RCCL over
xGMI (``torch.distributed`` backend "nccl" == RCCL on ROCm); results (ragged int32 token lists, a few
KB) are gathered on the host side with ``all_gather_object``.  One process per GPU.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist


def shard_streams(n_streams: int, rank: int, world: int) -> List[int]:
    """Stream s -> GPU s mod world (round-robin keeps ragged-length work balanced)."""
    return list(range(rank, n_streams, world))


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialise the process group from torchrun's env (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:                   # WM_DIST_BACKEND=gloo: smoke-test the N > 1 path with several ranks on ONE GPU
            backend = os.environ.get("WM_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def broadcast_blob(blob: Optional[torch.Tensor], offsets: Optional[np.ndarray], device, src: int = 0,
                   chunk_bytes: int = 512 << 20) -> Tuple[torch.Tensor, np.ndarray]:
    """Rank ``src`` owns (blob, offsets); every rank returns its own device copy.

    The blob (3.1 GB bf16 for large-v2 + 10 heads) goes out as a few large broadcasts: ring
    collectives over point-to-point xGMI are per-link bound (~153 GB/s), so a handful of >=512 MB
    messages amortise latency while keeping the pipeline full."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return blob.to(device), offsets
    rank = dist.get_rank()
    meta = [None]
    if rank == src:
        meta = [(int(blob.numel()), offsets.tolist())]
    dist.broadcast_object_list(meta, src=src)
    nbytes, offs = meta[0]
    if rank == src:
        buf = blob.to(device)
    else:
        buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
    for lo in range(0, nbytes, chunk_bytes):
        dist.broadcast(buf[lo: lo + chunk_bytes], src=src)
    return buf, np.asarray(offs, dtype=np.uint64)


def gather_token_lists(local: Sequence[Tuple[int, List[int]]]) -> List[List[int]]:
    """Every rank passes [(stream_index, ids), ...]; returns the full list ordered by stream index."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        items = list(local)
    else:
        parts = [None] * dist.get_world_size()
        dist.all_gather_object(parts, list(local))
        items = [it for p in parts for it in p]
    items.sort(key=lambda t: t[0])
    return [ids for _, ids in items]


def max_over_ranks(value: float, device=None) -> float:
    """MAX-reduce a scalar (bench timing: the slowest rank defines the step time)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device=None) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_floats(value: float, device=None) -> List[float]:
    """One scalar per rank, in rank order, on every rank (bench.py: tokens decoded per rank = which ranks really ran)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [float(value)]
    dev = device if dist.get_backend() == "nccl" else "cpu"
    mine = torch.tensor([value], dtype=torch.float64, device=dev)
    parts = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, mine)
    return [float(p.item()) for p in parts]


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
