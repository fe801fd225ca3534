"""Checkpoint -> packed HBM parameter blob ("checkpoint tooling").

Consumes a state dict with the reference checkpoint's key layout (written by
``trainer.py:45-51`` ``save_pretrained``) and produces ONE contiguous byte blob plus a table of byte
offsets in the canonical order ``csrc/wm_engine.hip`` expects (``wm_create``).  Matrices are bf16 in
the MFMA-fragment "packed" layout ``[N/16][K/32][64 lanes][8]`` (csrc/wm_common.h); LayerNorm
parameters, biases and position tables stay fp32.  On 8 GPUs rank 0 builds the blob and RCCL
broadcasts it (``dist.py``).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np
import torch

from . import frontend
from .config import MedusaConfig

ALIGN = 256


def _rup(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def pack_matrix(w: torch.Tensor) -> torch.Tensor:
    """Row-major [N, K] -> bf16 packed [N/16][K/32][64][8] (flat).  N % 16 == 0, K % 32 == 0."""
    n, k = w.shape
    assert n % 16 == 0 and k % 32 == 0, (n, k)
    t = w.to(torch.bfloat16).view(n // 16, 16, k // 32, 4, 8)      # (nt, r, kt, g, e)
    return t.permute(0, 2, 3, 1, 4).contiguous().view(-1)          # (nt, kt, g, r, e): lane = g*16 + r


def unpack_matrix(p: torch.Tensor, n: int, k: int) -> torch.Tensor:
    t = p.view(n // 16, k // 32, 4, 16, 8).permute(0, 3, 1, 2, 4).contiguous()
    return t.view(n, k)


def _pad2(w: torch.Tensor, n: int, k: int) -> torch.Tensor:
    if w.shape == (n, k):
        return w
    out = torch.zeros(n, k, dtype=w.dtype, device=w.device)
    out[: w.shape[0], : w.shape[1]] = w
    return out


def canonical_tensors(cfg: MedusaConfig, sd: Dict[str, torch.Tensor], device) -> List[torch.Tensor]:
    """The parameter list in the engine's canonical order; each entry is a flat fp32 or bf16 tensor."""
    d, dev = cfg.d_model, device
    f32 = lambda t: t.detach().to(dev, torch.float32).contiguous().view(-1)
    mat = lambda t: pack_matrix(t.detach().to(dev, torch.float32))
    zeros = lambda n: torch.zeros(n, dtype=torch.float32, device=dev)
    np32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev).view(-1)
    enc, dec = "whisper_model.model.encoder", "whisper_model.model.decoder"

    out: List[torch.Tensor] = []
    out += [np32(frontend.hann_window()), np32(frontend.dft_twiddles()), np32(frontend.slaney_mel_bank(cfg.num_mel_bins))]
    k1pad = _rup(3 * cfg.num_mel_bins, 128)
    c1 = sd[enc + ".conv1.weight"].to(dev, torch.float32).permute(0, 2, 1).reshape(d, 3 * cfg.num_mel_bins)   # k = kw*n_mels + c
    out += [mat(_pad2(c1, d, k1pad)), f32(sd[enc + ".conv1.bias"])]
    c2 = sd[enc + ".conv2.weight"].to(dev, torch.float32).permute(0, 2, 1).reshape(d, 3 * d)                   # k = kw*d + c
    out += [mat(c2), f32(sd[enc + ".conv2.bias"])]
    out += [f32(sd[enc + ".embed_positions.weight"]), f32(sd[enc + ".layer_norm.weight"]), f32(sd[enc + ".layer_norm.bias"])]
    emb = sd[dec + ".embed_tokens.weight"].to(dev, torch.float32)
    proj = sd.get("whisper_model.proj_out.weight", sd[dec + ".embed_tokens.weight"]).to(dev, torch.float32)
    vpad = _rup(cfg.vocab_size, 128)
    out += [emb.to(torch.bfloat16).contiguous().view(-1), mat(_pad2(proj, vpad, d))]
    out += [f32(sd[dec + ".embed_positions.weight"]), f32(sd[dec + ".layer_norm.weight"]), f32(sd[dec + ".layer_norm.bias"])]
    n_res = cfg.medusa_num_heads + (0 if cfg.is_block else 1)
    out += [mat(torch.cat([sd[f"medusa_heads.{k}.0.linear.weight"].to(dev, torch.float32) for k in range(n_res)], 0)),
            f32(torch.cat([sd[f"medusa_heads.{k}.0.linear.bias"].to(dev, torch.float32) for k in range(n_res)], 0))]
    kv_prefixes = [f"{dec}.layers.{i}" for i in range(cfg.decoder_layers)] + (["medusa_block"] if cfg.is_block else [])
    out += [mat(torch.cat([torch.cat([sd[p + ".encoder_attn.k_proj.weight"].to(dev, torch.float32),
                                       sd[p + ".encoder_attn.v_proj.weight"].to(dev, torch.float32)], 0) for p in kv_prefixes], 0)),
            torch.cat([torch.cat([zeros(d), f32(sd[p + ".encoder_attn.v_proj.bias"])]) for p in kv_prefixes])]

    def qkv(p):
        w = torch.cat([sd[p + ".q_proj.weight"], sd[p + ".k_proj.weight"], sd[p + ".v_proj.weight"]], 0)
        b = torch.cat([f32(sd[p + ".q_proj.bias"]), zeros(d), f32(sd[p + ".v_proj.bias"])])       # k_proj has no bias
        return [mat(w), b]

    def lin(p):
        return [mat(sd[p + ".weight"]), f32(sd[p + ".bias"])]

    def ln(p):
        return [f32(sd[p + ".weight"]), f32(sd[p + ".bias"])]

    for i in range(cfg.encoder_layers):
        p = f"{enc}.layers.{i}"
        out += ln(p + ".self_attn_layer_norm") + qkv(p + ".self_attn") + lin(p + ".self_attn.out_proj")
        out += ln(p + ".final_layer_norm") + lin(p + ".fc1") + lin(p + ".fc2")
    for p in kv_prefixes:
        out += ln(p + ".self_attn_layer_norm") + qkv(p + ".self_attn") + lin(p + ".self_attn.out_proj")
        out += ln(p + ".encoder_attn_layer_norm") + lin(p + ".encoder_attn.q_proj") + lin(p + ".encoder_attn.out_proj")
        out += ln(p + ".final_layer_norm") + lin(p + ".fc1") + lin(p + ".fc2")
    return out


def n_table_entries(cfg: MedusaConfig) -> int:
    return 19 + 12 * cfg.encoder_layers + 18 * cfg.n_kv_layers


def build_blob(cfg: MedusaConfig, sd: Dict[str, torch.Tensor], device="cpu") -> Tuple[torch.Tensor, np.ndarray]:
    """-> (uint8 blob on ``device``, uint64 offsets[n_table_entries])."""
    tensors = canonical_tensors(cfg, sd, device)
    assert len(tensors) == n_table_entries(cfg), (len(tensors), n_table_entries(cfg))
    offsets, total = [], 0
    for t in tensors:
        offsets.append(total)
        total = _rup(total + t.numel() * t.element_size(), ALIGN)
    blob = torch.zeros(total, dtype=torch.uint8, device=device)
    for off, t in zip(offsets, tensors):
        nb = t.numel() * t.element_size()
        blob[off: off + nb] = t.view(torch.uint8)
    return blob, np.asarray(offsets, dtype=np.uint64)


def load_state_dict_from_dir(path: str) -> Dict[str, torch.Tensor]:
    """Read ``model.safetensors`` (sharded or not) or ``pytorch_model.bin`` from a checkpoint directory."""
    import glob
    import os
    files = sorted(glob.glob(os.path.join(path, "*.safetensors")))
    sd: Dict[str, torch.Tensor] = {}
    if files:
        from safetensors.torch import load_file
        for f in files:
            sd.update(load_file(f))
        return sd
    binf = os.path.join(path, "pytorch_model.bin")
    if os.path.exists(binf):
        return torch.load(binf, map_location="cpu", weights_only=True)
    raise OSError(f"no model.safetensors / pytorch_model.bin under {path}")
