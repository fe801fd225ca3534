"""Synthetic inputs: seeded random-init checkpoints, 30 s clips and decoder prompts.

There are no real checkpoints or audio offline, so tests and ``bench.py`` use
the recipes here.  State-dict key names are exactly the
reference checkpoint's: ``whisper_model.model.encoder.*``,
``whisper_model.model.decoder.*``, ``whisper_model.proj_out.weight`` (tied),
``medusa_heads.{k}.0.linear.{weight,bias}``, ``medusa_block.*``.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import numpy as np
import torch

from .config import MedusaConfig, GenParams, ACCEPT_TYPICAL

SAMPLE_RATE = 16000
N_FFT = 400
HOP = 160


def synth_clip(i: int, n_samples: int = 480000) -> np.ndarray:
    """Clip *i* of SURVEY.md §8d: noise + four sinusoids, float32 in [-1, 1]."""
    rng = np.random.default_rng(1000 + i)
    t = np.arange(n_samples, dtype=np.float64) / SAMPLE_RATE
    wav = 0.05 * rng.standard_normal(n_samples)
    f = rng.uniform(100.0, 4000.0, size=4)
    ph = rng.uniform(0.0, 2 * np.pi, size=4)
    for j in range(4):
        wav += 0.1 * np.sin(2 * np.pi * f[j] * t + ph[j])
    return np.clip(wav, -1.0, 1.0).astype(np.float32)


def _sinusoids(length: int, channels: int) -> torch.Tensor:
    """Whisper's fixed encoder positional table (log-spaced sin/cos)."""
    inc = math.log(10000.0) / (channels // 2 - 1)
    inv = torch.exp(-inc * torch.arange(channels // 2, dtype=torch.float32))
    ang = torch.arange(length, dtype=torch.float32)[:, None] * inv[None, :]
    return torch.cat([ang.sin(), ang.cos()], dim=1)


def synth_state_dict(cfg: MedusaConfig, seed: int = 0, device: str = "cpu",
                     w_gain: float = 1.0, logit_std: float = 1.5, pos_std: float = 2.0,
                     head_w_std: float = 0.01, head_b_std: float = 0.3,
                     round_bf16: bool = True) -> Dict[str, torch.Tensor]:
    """Seeded random checkpoint with the reference's key layout.

    Linear weights ~ N(0, (w_gain/sqrt(fan_in))^2) so activations stay O(1) through
    32 layers; LayerNorm gains ~ 1 + 0.1 N, all biases non-zero so every fused
    epilogue term is exercised.  The tied token embedding has std ``logit_std/sqrt(d)``
    (so vocabulary logits have std ~ ``logit_std``: neither flat, where typical
    acceptance takes every candidate, nor one-hot), and decoder positions are wide
    (``pos_std``) so greedy decoding does not collapse to a repeated token; together
    they give varied tokens and a mixed accept histogram (probed).
    ``round_bf16`` rounds every tensor to bf16-representable fp32 values, i.e.
    the oracle and the engine see bit-identical parameters.
    """
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    d = cfg.d_model
    sd: Dict[str, torch.Tensor] = {}

    def normal(*shape, std=1.0):
        return torch.randn(*shape, generator=g, device=device, dtype=torch.float32) * std

    def linear(prefix, n_out, n_in, bias=True, gain=1.0):
        sd[prefix + ".weight"] = normal(n_out, n_in, std=gain * w_gain / math.sqrt(n_in))
        if bias:
            sd[prefix + ".bias"] = normal(n_out, std=0.05)

    def lnorm(prefix):
        sd[prefix + ".weight"] = 1.0 + normal(d, std=0.1)
        sd[prefix + ".bias"] = normal(d, std=0.05)

    def attn(prefix):
        linear(prefix + ".q_proj", d, d)
        linear(prefix + ".k_proj", d, d, bias=False)
        linear(prefix + ".v_proj", d, d)
        linear(prefix + ".out_proj", d, d, gain=0.5)

    enc = "whisper_model.model.encoder"
    sd[enc + ".conv1.weight"] = normal(d, cfg.num_mel_bins, 3, std=1.0 / math.sqrt(3 * cfg.num_mel_bins))
    sd[enc + ".conv1.bias"] = normal(d, std=0.05)
    sd[enc + ".conv2.weight"] = normal(d, d, 3, std=1.0 / math.sqrt(3 * d))
    sd[enc + ".conv2.bias"] = normal(d, std=0.05)
    sd[enc + ".embed_positions.weight"] = _sinusoids(cfg.max_source_positions, d).to(device)
    for i in range(cfg.encoder_layers):
        p = f"{enc}.layers.{i}"
        attn(p + ".self_attn")
        lnorm(p + ".self_attn_layer_norm")
        linear(p + ".fc1", cfg.encoder_ffn_dim, d)
        linear(p + ".fc2", d, cfg.encoder_ffn_dim, gain=0.5)
        lnorm(p + ".final_layer_norm")
    lnorm(enc + ".layer_norm")

    dec = "whisper_model.model.decoder"
    sd[dec + ".embed_tokens.weight"] = normal(cfg.vocab_size, d, std=logit_std / math.sqrt(d))
    sd[dec + ".embed_positions.weight"] = normal(cfg.max_target_positions, d, std=pos_std)

    def dec_layer(p):
        attn(p + ".self_attn")
        lnorm(p + ".self_attn_layer_norm")
        attn(p + ".encoder_attn")
        lnorm(p + ".encoder_attn_layer_norm")
        linear(p + ".fc1", cfg.decoder_ffn_dim, d)
        linear(p + ".fc2", d, cfg.decoder_ffn_dim, gain=0.5)
        lnorm(p + ".final_layer_norm")

    for i in range(cfg.decoder_layers):
        dec_layer(f"{dec}.layers.{i}")
    lnorm(dec + ".layer_norm")

    n_res_heads = cfg.medusa_num_heads + (0 if cfg.is_block else 1)   # model.py:235-246,256
    for k in range(n_res_heads):
        sd[f"medusa_heads.{k}.0.linear.weight"] = normal(d, d, std=head_w_std)
        sd[f"medusa_heads.{k}.0.linear.bias"] = normal(d, std=head_b_std)
    if cfg.is_block:
        dec_layer("medusa_block")

    if round_bf16:
        for k in sd:
            sd[k] = sd[k].to(torch.bfloat16).to(torch.float32)
    sd["whisper_model.proj_out.weight"] = sd[dec + ".embed_tokens.weight"]   # tied
    return sd


def default_prompt(cfg: MedusaConfig, language: Optional[str] = "en", task: str = "transcribe") -> List[int]:
    """Decoder prompt ids (G1; reference model.py:1519-1537 via HF ``_retrieve_init_tokens``)."""
    if cfg.is_multilingual:
        from .config import language_token
        key = language_token(language or "en")
        if key not in cfg.lang_to_id:
            raise ValueError(f"Unsupported language: {language}")
        return [cfg.decoder_start_token_id, cfg.lang_to_id[key], cfg.task_to_id[task],
                cfg.no_timestamps_token_id]
    return [cfg.decoder_start_token_id, cfg.no_timestamps_token_id]


def bench_gen_params(cfg: MedusaConfig, max_new_tokens: int = 128, accept_mode: int = ACCEPT_TYPICAL,
                     vanilla: bool = False) -> GenParams:
    """SURVEY.md §8d decode budget: fixed ``max_new_tokens``, EOS suppressed so every stream
    runs the full budget, exp-decay (140, 1.01) as in reference README.md:116-117."""
    prompt = default_prompt(cfg)
    sup = sorted(set((cfg.suppress_tokens or []) + [cfg.eos_token_id]))
    return GenParams(prompt=prompt, eos_token_id=cfg.eos_token_id, pad_token_id=cfg.pad_token_id,
                     suppress_tokens=sup, begin_suppress_tokens=list(cfg.begin_suppress_tokens or []),
                     max_length=min(len(prompt) + max_new_tokens, cfg.max_target_positions),
                     hard_max_length=cfg.max_length, exp_decay=(140, 1.01),
                     posterior_threshold=cfg.posterior_threshold, posterior_alpha=cfg.posterior_alpha,
                     accept_mode=accept_mode, temperature=1.0 if accept_mode == ACCEPT_TYPICAL else 0.0,
                     vanilla=vanilla)
