"""Micro-batch concurrency on one GPU.

A decode iteration of a few dozen streams is a chain of several hundred short kernels, most of which do not fill
256 CUs; two half-size batches on two HIP streams overlap each other's launch gaps and tails.  ``ContextPool`` owns N
engine contexts (one HIP stream each, ONE shared weight blob), shards a batch of clips into N contiguous
micro-batches and drives every context from its own host thread (ctypes releases the GIL inside libwm.so).  Streams
never interact — the engine's results do not depend on batch composition — so the tokens are those of a single
context run (tests/test_gpu_parity.py::test_micro_batches_match_single_context).

Measured on MI355X, whisper-large-v2 + Medusa-Linear K=10 (tests/microbench/two_ctx.py, bench.py --micro-batches):
32 clips: 1 x 32 streams 6.9 k tokens/s, 2 x 16 streams 7.3-7.7 k; 64 clips as 2 x 32: 8.2 k; 2 clips as 2 x 1 (each
context then runs the host-driven single-stream schedule): 1.77 k vs 1.33 k batched.  It does NOT help in between
(4 x 1: 1.6 k vs 2.6 k batched, 2 x 4: 3.3 k vs 3.9 k): more host threads synchronising every iteration, and every
context re-streams the weights.  Two contexts is the useful setting.
"""
from __future__ import annotations

from concurrent.futures import ThreadPoolExecutor
from typing import List, Sequence, Tuple

import numpy as np
import torch

from .config import GenParams, MedusaConfig
from .engine import Engine


def shard_bounds(n: int, parts: int) -> List[Tuple[int, int]]:
    """Contiguous, balanced [lo, hi) ranges; empty shards are dropped (fewer clips than contexts)."""
    parts = max(1, min(parts, n))
    q, r = divmod(n, parts)
    out, lo = [], 0
    for i in range(parts):
        hi = lo + q + (1 if i < r else 0)
        out.append((lo, hi))
        lo = hi
    return out


def merge_stats(stats: Sequence[dict]) -> dict:
    """Whole-batch view of per-context statistics: counters add up, times are the slowest context's."""
    out = dict(stats[0])
    for k in ("tokens_emitted", "iterations_launched", "graph_replays", "schedule_steps"):
        out[k] = sum(s.get(k, 0) for s in stats)
    out["iterations"] = max(s["iterations"] for s in stats)
    out["accept_hist"] = np.sum([np.asarray(s["accept_hist"]) for s in stats], axis=0).tolist()
    for k in ("ms_logmel", "ms_encode", "ms_decode"):
        out[k] = max(s[k] for s in stats)
    out["micro_batches"] = len(stats)
    return out


class ContextPool:
    def __init__(self, cfg: MedusaConfig, blob: torch.Tensor, offsets: np.ndarray, contexts: int, max_batch: int,
                 dec_weight_fp8: bool = False, enc_fp8: bool = False, act_fp16: bool = False, cross_kv_fp8: bool = False):
        if contexts < 1:
            raise ValueError("contexts must be >= 1")
        self.cfg = cfg
        self.per_ctx = (max_batch + contexts - 1) // contexts
        self._mk = lambda: Engine(cfg, blob, offsets, max_batch=self.per_ctx, device=blob.device, dec_weight_fp8=dec_weight_fp8, enc_fp8=enc_fp8, act_fp16=act_fp16, cross_kv_fp8=cross_kv_fp8)
        self.engines = [self._mk() for _ in range(contexts)]
        self._ex = ThreadPoolExecutor(max_workers=contexts, thread_name_prefix="wm-ctx") if contexts > 1 else None
        self.last_stats: dict = {}

    def grow(self, contexts: int) -> None:
        """Add contexts of the same size (KV caches, scratch, captured graphs of the existing ones are kept)."""
        if contexts <= len(self.engines):
            return
        self.engines += [self._mk() for _ in range(contexts - len(self.engines))]
        if self._ex is not None:
            self._ex.shutdown(wait=True)
        self._ex = ThreadPoolExecutor(max_workers=contexts, thread_name_prefix="wm-ctx")

    def close(self):
        if self._ex is not None:
            self._ex.shutdown(wait=True)
            self._ex = None
        for e in self.engines:
            e.close()
        self.engines = []

    def _one(self, eng: Engine, x: torch.Tensor, gp: GenParams, from_wav: bool):
        with torch.cuda.device(eng.device):
            feats = eng.logmel(x) if from_wav else x
            eng.encode(feats)
            seqs = eng.decode(gp, x.shape[0])
            return seqs, eng.stats()

    def run(self, x: torch.Tensor, gp: GenParams, from_wav: bool = False) -> List[List[int]]:
        """x: input features [B, n_mels, frames] (or waveforms [B, samples] with ``from_wav``) on the pool's GPU."""
        B = x.shape[0]
        if B > self.per_ctx * len(self.engines):
            raise ValueError(f"batch of {B} exceeds the pool capacity {self.per_ctx * len(self.engines)}")
        bounds = shard_bounds(B, len(self.engines))
        if len(bounds) == 1:
            res = [self._one(self.engines[0], x, gp, from_wav)]
        else:
            futs = [self._ex.submit(self._one, self.engines[i], x[lo:hi].contiguous(), gp, from_wav)
                    for i, (lo, hi) in enumerate(bounds)]
            res = [f.result() for f in futs]
        self.last_stats = merge_stats([st for _, st in res])
        return [s for seqs, _ in res for s in seqs]
