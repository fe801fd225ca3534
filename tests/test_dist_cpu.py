"""The N>1 data-parallel path on CPU: two gloo processes shard streams, broadcast the packed blob
from rank 0 and gather the ragged token lists.  No GPU."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from helpers import ROOT, MedusaConfig, synth


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "whisper-medusa_amd"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from whisper_medusa import dist as wd, weights
    r, _, w = wd.init_from_env(backend="gloo")
    cfg = MedusaConfig.micro(K=4)
    blob = offs = None
    if r == 0:                                        # only rank 0 touches the checkpoint
        blob, offs = weights.build_blob(cfg, synth.synth_state_dict(cfg, seed=7))
    blob, offs = wd.broadcast_blob(blob, offs, device="cpu", chunk_bytes=1 << 16)
    mine = wd.shard_streams(7, r, w)
    local = [(s, [s] * (s + 1)) for s in mine]        # ragged per-stream results
    allv = wd.gather_token_lists(local)
    tmax = wd.max_over_ranks(1.0 + r)
    tot = wd.sum_over_ranks(len(mine))
    per_rank = wd.gather_floats(float(len(mine)))
    wd.barrier()
    q.put((r, int(blob.to(torch.int64).sum()), offs.tolist(), allv, tmax, tot, per_rank))


def test_two_rank_shard_broadcast_gather():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    cfg = MedusaConfig.micro(K=4)
    from whisper_medusa import weights
    blob, offs = weights.build_blob(cfg, synth.synth_state_dict(cfg, seed=7))
    for r, bsum, o, allv, tmax, tot, per_rank in res:
        assert per_rank == [4.0, 3.0]                                                  # rank order, on every rank
        assert bsum == int(blob.to(torch.int64).sum()) and o == offs.tolist()       # identical weights everywhere
        assert allv == [[s] * (s + 1) for s in range(7)]                               # ordered, ragged, complete
        assert tmax == 2.0 and tot == 7


def test_shard_is_a_partition():
    from whisper_medusa.dist import shard_streams
    for n, w in ((256, 8), (7, 2), (3, 8), (32, 1)):
        parts = [shard_streams(n, r, w) for r in range(w)]
        assert sorted(x for p in parts for x in p) == list(range(n))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
