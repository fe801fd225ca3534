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


def pack_matrix(w: torch.Tensor, fp16: bool = False) -> torch.Tensor:
    """Row-major [N, K] -> bf16 packed [N/16][K/32][64][8] (flat).  N % 16 == 0, K % 32 == 0.
    ``fp16`` (wm_config.act_fp16, the fp16 single-plane decode contract): the bf16-rounded parameter re-expressed as fp16 — the same value
    for every |w| >= 2^-17 (bf16 carries 8 significant bits, fp16's subnormal step is 2^-24), absolute error <= 3e-8 below — returned as the
    16-bit container the blob stores (bfloat16-typed view of the fp16 bits)."""
    n, k = w.shape
    assert n % 16 == 0 and k % 32 == 0, (n, k)
    t = w.to(torch.bfloat16)
    if fp16:
        t = t.to(torch.float32).to(torch.float16).view(torch.bfloat16)
    t = t.view(n // 16, 16, k // 32, 4, 8)                          # (nt, r, kt, g, e)
    return t.permute(0, 2, 3, 1, 4).contiguous().view(-1)          # (nt, kt, g, r, e): lane = g*16 + r


def unpack_matrix(p: torch.Tensor, n: int, k: int) -> torch.Tensor:
    t = p.view(n // 16, k // 32, 4, 16, 8).permute(0, 3, 1, 2, 4).contiguous()
    return t.view(n, k)


FP8_MAX = 448.0                  # largest finite e4m3 (OCP "fn") magnitude


def quantize_rows_e4m3(w: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Per-output-row symmetric fp8 quantisation (BASELINE.json configs[4]): W[n][k] ~= scale[n] * q[n][k] with
    scale[n] = max_k |W[n][k]| / 448 and q rounded to nearest-even e4m3.  Returns (q as float8_e4m3fn [N, K], fp32 scale [N])."""
    # always on the CPU: GPU elementwise kernels turn `x / 448` into `x * (1 / 448)`, which moves a scale by one ulp and
    # flips round-to-nearest ties of q — the quantised checkpoint must not depend on where it was packed
    w = w.detach().to("cpu", torch.float32)
    amax = w.abs().amax(dim=1)
    scale = torch.where(amax > 0, amax / FP8_MAX, torch.ones_like(amax))
    q = (w / scale[:, None]).clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn)
    return q, scale


def pack_matrix_fp8(q: torch.Tensor) -> torch.Tensor:
    """float8_e4m3fn [N, K] -> uint8 in the packed fragment layout [N/16][K/32][64][8] (one byte per element)."""
    n, k = q.shape
    assert n % 16 == 0 and k % 32 == 0, (n, k)
    t = q.view(torch.uint8).view(n // 16, 16, k // 32, 4, 8)
    return t.permute(0, 2, 3, 1, 4).contiguous().view(-1)


def pack_matrix_fp8_k128(q: torch.Tensor) -> torch.Tensor:
    """float8_e4m3fn [N, K] -> uint8 in the fp8-MFMA operand layout of v_mfma_f32_16x16x128_f8f6f4,
    [N/16][Kp/128][2 halves][64 lanes][16]: lane (r = row & 15, g) of a 128-k unit holds the 32 consecutive k = 32 g .. + 31, half h
    its bytes 16 h .. + 15 (csrc/wm_encoder.hip, f8k_index).  N % 16 == 0; K is zero-padded to a multiple of 128 (Kp)."""
    n, k = q.shape
    assert n % 16 == 0, (n, k)
    kp = (k + 127) // 128 * 128
    u8 = q.view(torch.uint8)
    if kp != k:
        u8 = torch.cat([u8, torch.zeros(n, kp - k, dtype=torch.uint8, device=u8.device)], dim=1)      # e4m3 zero is 0x00
    t = u8.reshape(n // 16, 16, kp // 128, 4, 2, 16)                     # (nt, r, u, g, h, e)
    return t.permute(0, 2, 4, 3, 1, 5).contiguous().view(-1)              # (nt, u, h, g, r, e): lane = g*16 + r


def _pad2(w: torch.Tensor, n: int, k: int) -> torch.Tensor:
    if w.shape == (n, k):
        return w
    out = torch.zeros(n, k, dtype=w.dtype, device=w.device)
    out[: w.shape[0], : w.shape[1]] = w
    return out


def canonical_tensors(cfg: MedusaConfig, sd: Dict[str, torch.Tensor], device, dec_fp8: bool = False, enc_fp8: bool = False,
                      act_fp16: bool = False) -> List[torch.Tensor]:
    """The parameter list in the engine's canonical order; each entry is a flat fp32 / bf16 (/ uint8 fp8) tensor.
    ``dec_fp8``: the six matrices of every decoder layer are stored as fp8 e4m3 and their per-row scales are appended.
    ``enc_fp8``: the encoder matrices that multiply a LayerNorm output (q/k/v, fc1) and the fused cross-K/V projection are
    ALSO given as e4m3 in the fp8-MFMA layout with per-row scales (appended last: 4 entries per encoder layer + 2); their
    bf16 entries shrink to 16-byte placeholders (the engine does not read them in that mode).
    ``act_fp16``: the matrices the DECODE GEMMs stream — decoder-layer matrices (unless fp8), Medusa heads, packed vocabulary projection —
    are stored as fp16 (`pack_matrix(fp16=True)`); everything the encoder reads stays bf16."""
    d, dev = cfg.d_model, device
    f32 = lambda t: t.detach().to(dev, torch.float32).contiguous().view(-1)
    mat = lambda t: pack_matrix(t.detach().to(dev, torch.float32))
    dmat = lambda t: pack_matrix(t.detach().to(dev, torch.float32), fp16=act_fp16)          # a matrix of the decode GEMMs
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
    out += [emb.to(torch.bfloat16).contiguous().view(-1), dmat(_pad2(proj, vpad, d))]
    out += [f32(sd[dec + ".embed_positions.weight"]), f32(sd[dec + ".layer_norm.weight"]), f32(sd[dec + ".layer_norm.bias"])]
    n_res = cfg.medusa_num_heads + (0 if cfg.is_block else 1)
    out += [dmat(torch.cat([sd[f"medusa_heads.{k}.0.linear.weight"].to(dev, torch.float32) for k in range(n_res)], 0)),
            f32(torch.cat([sd[f"medusa_heads.{k}.0.linear.bias"].to(dev, torch.float32) for k in range(n_res)], 0))]
    kv_prefixes = [f"{dec}.layers.{i}" for i in range(cfg.decoder_layers)] + (["medusa_block"] if cfg.is_block else [])
    enc8: List[torch.Tensor] = []
    placeholder = lambda: torch.zeros(8, dtype=torch.bfloat16, device=dev)

    def emat(w):                 # encoder matrix with a LayerNorm-output operand
        if not enc_fp8:
            return mat(w)
        q, sc = quantize_rows_e4m3(w)
        enc8.extend([pack_matrix_fp8_k128(q).to(dev), sc.to(dev).contiguous()])
        return placeholder()

    ckv = torch.cat([torch.cat([sd[p + ".encoder_attn.k_proj.weight"].to(dev, torch.float32),
                                sd[p + ".encoder_attn.v_proj.weight"].to(dev, torch.float32)], 0) for p in kv_prefixes], 0)
    ckv8: List[torch.Tensor] = []
    if enc_fp8:
        q, sc = quantize_rows_e4m3(ckv)
        ckv8 = [pack_matrix_fp8_k128(q).to(dev), sc.to(dev).contiguous()]
    out += [placeholder() if enc_fp8 else mat(ckv),
            torch.cat([torch.cat([zeros(d), f32(sd[p + ".encoder_attn.v_proj.bias"])]) for p in kv_prefixes])]

    scales: List[torch.Tensor] = []

    def wmat(w, fp8):
        if not fp8:
            return dmat(w)
        q, sc = quantize_rows_e4m3(w)
        scales.append(sc.to(dev).contiguous())
        return pack_matrix_fp8(q).to(dev)

    def qkv(p, fp8=False):
        w = torch.cat([sd[p + ".q_proj.weight"], sd[p + ".k_proj.weight"], sd[p + ".v_proj.weight"]], 0)
        b = torch.cat([f32(sd[p + ".q_proj.bias"]), zeros(d), f32(sd[p + ".v_proj.bias"])])       # k_proj has no bias
        return [wmat(w, fp8), b]

    def lin(p, fp8=False):
        return [wmat(sd[p + ".weight"], fp8), f32(sd[p + ".bias"])]

    def ln(p):
        return [f32(sd[p + ".weight"]), f32(sd[p + ".bias"])]

    for i in range(cfg.encoder_layers):
        p = f"{enc}.layers.{i}"
        wq = torch.cat([sd[p + ".self_attn.q_proj.weight"], sd[p + ".self_attn.k_proj.weight"], sd[p + ".self_attn.v_proj.weight"]], 0)
        bq = torch.cat([f32(sd[p + ".self_attn.q_proj.bias"]), zeros(d), f32(sd[p + ".self_attn.v_proj.bias"])])
        elin = lambda q_: [mat(sd[q_ + ".weight"]), f32(sd[q_ + ".bias"])]          # encoder matrices stay bf16 whatever the decode contract
        out += ln(p + ".self_attn_layer_norm") + [emat(wq), bq] + elin(p + ".self_attn.out_proj")
        out += ln(p + ".final_layer_norm") + [emat(sd[p + ".fc1.weight"]), f32(sd[p + ".fc1.bias"])] + elin(p + ".fc2")
    for p in kv_prefixes:
        out += ln(p + ".self_attn_layer_norm") + qkv(p + ".self_attn", dec_fp8) + lin(p + ".self_attn.out_proj", dec_fp8)
        out += ln(p + ".encoder_attn_layer_norm") + lin(p + ".encoder_attn.q_proj", dec_fp8) + lin(p + ".encoder_attn.out_proj", dec_fp8)
        out += ln(p + ".final_layer_norm") + lin(p + ".fc1", dec_fp8) + lin(p + ".fc2", dec_fp8)
    # fp8: 6 scale vectors per decoder layer, in launch order (qkv, out, cq, cout, fc1, fc2); then the fp8-MFMA encoder entries
    return out + scales + enc8 + ckv8


def n_table_entries(cfg: MedusaConfig, dec_fp8: bool = False, enc_fp8: bool = False) -> int:
    return (19 + 12 * cfg.encoder_layers + 18 * cfg.n_kv_layers + (6 * cfg.n_kv_layers if dec_fp8 else 0)
            + (4 * cfg.encoder_layers + 2 if enc_fp8 else 0))


def build_blob(cfg: MedusaConfig, sd: Dict[str, torch.Tensor], device="cpu", dec_fp8: bool = False, enc_fp8: bool = False,
               act_fp16: bool = False) -> Tuple[torch.Tensor, np.ndarray]:
    """-> (uint8 blob on ``device``, uint64 offsets[n_table_entries])."""
    tensors = canonical_tensors(cfg, sd, device, dec_fp8, enc_fp8, act_fp16)
    assert len(tensors) == n_table_entries(cfg, dec_fp8, enc_fp8), (len(tensors), n_table_entries(cfg, dec_fp8, enc_fp8))
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
        return resolve_tied_embeddings(sd)
    binf = os.path.join(path, "pytorch_model.bin")
    if os.path.exists(binf):
        return resolve_tied_embeddings(torch.load(binf, map_location="cpu", weights_only=True))
    raise OSError(f"no model.safetensors / pytorch_model.bin under {path}")


def resolve_tied_embeddings(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """HF ties ``proj_out.weight`` to the decoder's ``embed_tokens.weight`` and a safetensors writer keeps ONE of the two names
    (transformers keeps embed_tokens; ``safetensors.torch.save_model`` keeps whichever sorts first).  Whichever survived serves both."""
    emb, proj = "whisper_model.model.decoder.embed_tokens.weight", "whisper_model.proj_out.weight"
    if emb not in sd and proj in sd:
        sd[emb] = sd[proj]
    return sd
