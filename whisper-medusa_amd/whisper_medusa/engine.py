"""ctypes binding of libwm.so (include/wm.h).  PyTorch is used only to own device memory and
the HIP stream; every call below crosses the C-ABI with plain pointers and sizes.

There is NO fallback: if the HIP library is missing or there is no GPU, construction raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import numpy as np
import torch

from .config import MedusaConfig, GenParams, HEADS_BLOCK

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libwm.so")      # the product library; tests/microbench scripts may point WM_LIB at a debug build
LIB_PATH_F16 = os.path.join(_HERE, "libwm_f16.so")  # the same sources built for the fp16 single-plane decode contract (wm_config.act_fp16)
WM_ABI_VERSION = 9


class WmConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "abi_version", "d_model", "enc_layers", "dec_layers", "n_heads", "ffn_dim", "vocab", "n_mels",
        "n_ctx", "n_tgt", "medusa_heads", "heads_type", "max_batch", "dec_weight_fp8")] + [("medusa_choices", C.c_int32 * 16), ("enc_fp8", C.c_int32),
                                                                                          ("act_fp16", C.c_int32), ("cross_kv_fp8", C.c_int32), ("sibling_rows", C.c_int32)]


class WmWeights(C.Structure):
    _fields_ = [("blob", C.c_void_p), ("blob_bytes", C.c_uint64), ("offsets", C.POINTER(C.c_uint64)),
                ("n_offsets", C.c_int32)]


class WmGenParams(C.Structure):
    _fields_ = [("prompt", C.POINTER(C.c_int32)), ("prompt_len", C.c_int32),
                ("eos_token_id", C.c_int32), ("pad_token_id", C.c_int32),
                ("suppress", C.POINTER(C.c_int32)), ("n_suppress", C.c_int32),
                ("begin_suppress", C.POINTER(C.c_int32)), ("n_begin_suppress", C.c_int32),
                ("max_length", C.c_int32), ("hard_max_length", C.c_int32),
                ("exp_decay_start", C.c_int32), ("exp_decay_factor", C.c_float),
                ("posterior_threshold", C.c_float), ("posterior_alpha", C.c_float),
                ("temperature", C.c_float), ("accept_mode", C.c_int32), ("vanilla", C.c_int32), ("begin_index", C.c_int32),
                ("force_accept", C.c_int32)]


class WmStats(C.Structure):
    _fields_ = [("iterations", C.c_int64), ("iterations_launched", C.c_int64), ("tokens_emitted", C.c_int64),
                ("accept_hist", C.c_int64 * 16),
                ("ms_logmel", C.c_float), ("ms_encode", C.c_float), ("ms_decode", C.c_float),
                ("graph_replays", C.c_int32), ("schedule_steps", C.c_int32), ("sibling_hits", C.c_int32)]


EXPORTS = ["wm_create", "wm_destroy", "wm_last_error", "wm_abi_version", "wm_build_act_fp16", "wm_resample_len", "wm_resample", "wm_logmel", "wm_encode", "wm_set_encoder_output",
           "wm_decode_begin", "wm_decode_run", "wm_get_tokens", "wm_get_stats", "wm_sync",
           "wm_get_encoder_output", "wm_forward_logits", "wm_get_cross_kv", "wm_profile_kernel"]

_lib = {}


def default_sibling_rows() -> int:
    """Sibling rows a context is created with when the caller does not choose: WM_SIBLINGS (default 5; the engine caps it at 15 - K).  They only
    act in single-stream decodes over a candidate chain (include/wm.h: wm_config.sibling_rows); emitted ids are the reference's with or without."""
    return int(os.environ.get("WM_SIBLINGS", "5"))


def default_act_fp16() -> bool:
    """The decode numerics contract a model gets when the caller does not choose (``act_fp16=None``): WM_ACT=f16|hilo.  Default f16 (round 6:
    token ids identical to the hi / lo contract and to the fp32-pinned tables in every compared run, one stream +7 %, 32 streams +16 %;
    DESIGN.md §2) — WM_ACT=hilo keeps rounds 1-5's bf16 hi / lo operand pairs."""
    return os.environ.get("WM_ACT", "f16").lower() in ("f16", "fp16")


def load_library(path: Optional[str] = None, act_fp16: bool = False) -> C.CDLL:
    """dlopen libwm.so (bf16 hi / lo decode contract) or libwm_f16.so (fp16 single-plane contract) and declare the prototypes.
    Raises if the library has not been built."""
    global _lib
    if path is None and act_fp16 in _lib:
        return _lib[act_fp16]
    p = path or os.environ.get("WM_LIB_F16" if act_fp16 else "WM_LIB") or (LIB_PATH_F16 if act_fp16 else LIB_PATH)
    if not os.path.exists(p):
        raise RuntimeError(f"{p} not found: build the HIP engine first (python whisper-medusa_amd/build.py). "
                           "There is no CPU fallback.")
    lib = C.CDLL(p)
    vp, i32, f32p, i32p = C.c_void_p, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_int32)
    lib.wm_create.argtypes = [C.POINTER(WmConfig), C.POINTER(WmWeights), i32, vp, C.POINTER(vp)]
    lib.wm_destroy.argtypes = [vp]; lib.wm_destroy.restype = None
    lib.wm_last_error.argtypes = [vp]; lib.wm_last_error.restype = C.c_char_p
    lib.wm_abi_version.argtypes = []
    lib.wm_build_act_fp16.argtypes = []
    lib.wm_resample_len.argtypes = [C.c_int64, i32, i32]; lib.wm_resample_len.restype = C.c_int64
    lib.wm_resample.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp]
    lib.wm_logmel.argtypes = [vp, vp, i32, i32, vp]
    lib.wm_encode.argtypes = [vp, vp, i32]
    lib.wm_set_encoder_output.argtypes = [vp, vp, i32]
    lib.wm_decode_begin.argtypes = [vp, C.POINTER(WmGenParams), i32]
    lib.wm_decode_run.argtypes = [vp, i32, i32p]
    lib.wm_get_tokens.argtypes = [vp, i32, i32p, i32, i32p]
    lib.wm_get_stats.argtypes = [vp, C.POINTER(WmStats)]
    lib.wm_sync.argtypes = [vp]
    lib.wm_get_encoder_output.argtypes = [vp, i32, f32p]
    lib.wm_forward_logits.argtypes = [vp, i32, i32p, i32, i32, i32, f32p]
    lib.wm_get_cross_kv.argtypes = [vp, i32, i32, i32, f32p, f32p]
    lib.wm_profile_kernel.argtypes = [vp, i32, i32, i32, f32p, C.POINTER(C.c_double)]
    for name in EXPORTS:
        if name not in ("wm_destroy", "wm_last_error", "wm_resample_len"):      # wm_resample_len returns int64 (set above)
            getattr(lib, name).restype = i32
    if lib.wm_abi_version() != WM_ABI_VERSION and not (os.environ.get("WM_LIB") and os.environ.get("WM_ABI_ANY")):
        raise RuntimeError("libwm.so ABI version mismatch")        # WM_ABI_ANY: A/B runs against an older debug build (tests/microbench)
    if bool(lib.wm_build_act_fp16()) != bool(act_fp16) and path is None:
        raise RuntimeError(f"{p} is built for the other decode contract (wm_build_act_fp16)")
    if path is None:
        _lib[act_fp16] = lib
    return lib


def _i32arr(v: Sequence[int]):
    a = (C.c_int32 * max(len(v), 1))(*[int(x) for x in v])
    return a


class Engine:
    """One context = one GPU.  ``blob`` is the packed parameter tensor (uint8, on that GPU)."""

    def __init__(self, cfg: MedusaConfig, blob: torch.Tensor, offsets: np.ndarray, max_batch: int = 1,
                 device: Optional[torch.device] = None, dec_weight_fp8: bool = False, enc_fp8: bool = False, act_fp16: bool = False,
                 cross_kv_fp8: bool = False, sibling_rows: Optional[int] = None):
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible: the Whisper-Medusa engine has no CPU path")
        self.act_fp16 = bool(act_fp16)
        self.lib = load_library(act_fp16=self.act_fp16)
        # wm_config.sibling_rows (include/wm.h): head 1's next-best tokens in the spare rows of a single-stream verify pass; WM_SIBLINGS=0 turns them off
        self.sibling_rows = default_sibling_rows() if sibling_rows is None else int(sibling_rows)
        self.cfg = cfg
        self.device = torch.device(device if device is not None else blob.device)
        if self.device.type != "cuda" or blob.device != self.device:
            raise RuntimeError("parameter blob must live on the engine's GPU")
        self.blob = blob                      # keep alive: the engine borrows it
        self.max_batch = int(max_batch)
        self._offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        c = WmConfig(WM_ABI_VERSION, cfg.d_model, cfg.encoder_layers, cfg.decoder_layers, cfg.n_heads,
                     cfg.decoder_ffn_dim, cfg.vocab_size, cfg.num_mel_bins, cfg.max_source_positions,
                     cfg.max_target_positions, cfg.medusa_num_heads,
                     1 if cfg.medusa_heads_type == HEADS_BLOCK else 0, self.max_batch, 1 if dec_weight_fp8 else 0,
                     (C.c_int32 * 16)(*[int(x) for x in cfg.medusa_choices]), 1 if enc_fp8 else 0, 1 if self.act_fp16 else 0, 1 if cross_kv_fp8 else 0, self.sibling_rows)
        w = WmWeights(C.c_void_p(blob.data_ptr()), blob.numel(),
                      self._offsets.ctypes.data_as(C.POINTER(C.c_uint64)), len(self._offsets))
        # every context owns a private non-blocking HIP stream (NULL -> wm_create makes one): several contexts built under
        # one torch.cuda.stream(s) block must not capture graphs on / launch into the same stream from different host threads
        h = C.c_void_p()
        rc = self.lib.wm_create(C.byref(c), C.byref(w), self.device.index or 0, C.c_void_p(None), C.byref(h))
        if rc != 0:
            raise RuntimeError(f"wm_create failed ({rc}): {self.lib.wm_last_error(None).decode()}")
        self.h = h
        self._B = None
        self._enc_stamp = None
        self._kv_stamp = object()

    def close(self):
        if getattr(self, "h", None):
            self.lib.wm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str):
        if rc != 0:
            msg = self.lib.wm_last_error(self.h).decode()
            if rc == -1:
                raise ValueError(f"{what}: {msg}")
            raise RuntimeError(f"{what} failed ({rc}): {msg}")

    def _inputs_ready(self):
        """The context runs on its own HIP stream: tensors PyTorch is still producing on its stream must be complete
        before the engine reads them (pageable H2D copies return before the DMA lands, kernels are asynchronous)."""
        torch.cuda.current_stream(self.device).synchronize()

    # ---- audio front door -----------------------------------------------------------------
    def resample(self, wav: torch.Tensor, sr_in: int, sr_out: int = 16000) -> torch.Tensor:
        """wav [B, channels, n] (or [B, n]) float32 on the GPU -> mono [B, ceil(n * sr_out / sr_in)] at ``sr_out``:
        channel mean + torchaudio-default windowed-sinc resampling (README.md:120-125 of the reference)."""
        wav = wav.to(self.device, torch.float32)
        if wav.dim() == 2:
            wav = wav[:, None, :]
        wav = wav.contiguous()
        B, ch, n = wav.shape
        n_out = int(self.lib.wm_resample_len(n, int(sr_in), int(sr_out)))
        if n_out < 1:
            raise ValueError("resample: empty input or bad sampling rates")
        out = torch.empty(B, n_out, dtype=torch.float32, device=wav.device)
        self._inputs_ready()
        self._check(self.lib.wm_resample(self.h, C.c_void_p(wav.data_ptr()), B, ch, n, int(sr_in), int(sr_out),
                                         C.c_void_p(out.data_ptr())), "wm_resample")
        return out

    # ---- F0 -------------------------------------------------------------------------------
    def logmel(self, wav: torch.Tensor) -> torch.Tensor:
        """wav [B, 160*2*n_ctx] float32 on the GPU -> features [B, n_mels, 2*n_ctx]."""
        wav = wav.to(self.device, torch.float32).contiguous()
        B, n = wav.shape
        feats = torch.empty(B, self.cfg.num_mel_bins, self.cfg.n_mel_frames, dtype=torch.float32, device=wav.device)
        self._inputs_ready()
        self._check(self.lib.wm_logmel(self.h, C.c_void_p(wav.data_ptr()), B, n, C.c_void_p(feats.data_ptr())), "wm_logmel")
        return feats

    # ---- F1/F2 ------------------------------------------------------------------------------
    def encode(self, feats: torch.Tensor) -> None:
        feats = feats.to(self.device, torch.float32).contiguous()
        B = feats.shape[0]
        if tuple(feats.shape[1:]) != (self.cfg.num_mel_bins, self.cfg.n_mel_frames):
            raise ValueError(f"Whisper expects the mel input features to be of length {self.cfg.n_mel_frames}, "
                             f"but found {feats.shape[-1]}")
        self._inputs_ready()
        self._check(self.lib.wm_encode(self.h, C.c_void_p(feats.data_ptr()), B), "wm_encode")
        self._B = B
        self._enc_stamp = object()            # identity of the resident encoder pass (api.forward: encoder_outputs / past_key_values handles)

    def set_encoder_output(self, hidden: torch.Tensor) -> None:
        """hidden [B, n_ctx, d_model] (the encoder's last hidden state, e.g. ``encoder_outputs[0]`` of a previous forward) replaces
        the encoder pass: stored bf16, cross K/V projected from it (reference forward(encoder_outputs=...), model.py:1232)."""
        hidden = hidden.to(self.device, torch.float32).contiguous()
        if hidden.dim() != 3 or tuple(hidden.shape[1:]) != (self.cfg.max_source_positions, self.cfg.d_model):
            raise ValueError(f"encoder_outputs[0] must be [B, {self.cfg.max_source_positions}, {self.cfg.d_model}], got {tuple(hidden.shape)}")
        self._inputs_ready()
        self._check(self.lib.wm_set_encoder_output(self.h, C.c_void_p(hidden.data_ptr()), hidden.shape[0]), "wm_set_encoder_output")
        self._B = hidden.shape[0]
        self._enc_stamp = object()

    # ---- F3..F14 ------------------------------------------------------------------------------
    def decode(self, gp: GenParams, B: int, max_iters: int = 1 << 30, on_iteration=None) -> List[List[int]]:
        prompt, sup, bsup = _i32arr(gp.prompt), _i32arr(gp.suppress_tokens), _i32arr(gp.begin_suppress_tokens)
        g = WmGenParams(prompt, len(gp.prompt), gp.eos_token_id, gp.pad_token_id, sup, len(gp.suppress_tokens),
                        bsup, len(gp.begin_suppress_tokens), gp.max_length, gp.hard_max_length,
                        int(gp.exp_decay[0]) if gp.exp_decay is not None else -1,      # the eval CLI parses the start as float
                        float(gp.exp_decay[1]) if gp.exp_decay is not None else 1.0,
                        gp.posterior_threshold, gp.posterior_alpha, gp.temperature if gp.temperature else 0.0,
                        gp.accept_mode, 1 if gp.vanilla else 0, int(gp.begin_index), int(getattr(gp, "force_accept", -1)))
        self._kv_stamp = object()             # the decode loop rewrites the self-attention cache: forward()'s cache handles go stale
        self._check(self.lib.wm_decode_begin(self.h, C.byref(g), B), "wm_decode_begin")
        left = C.c_int32(0)
        if on_iteration is None:
            self._check(self.lib.wm_decode_run(self.h, max_iters, C.byref(left)), "wm_decode_run")
            return [self.tokens(b) for b in range(B)]
        # streaming: one iteration per call, the tokens each stream gained are handed over as they appear
        seen = [len(gp.prompt)] * B
        it = 0
        while it < max_iters:
            self._check(self.lib.wm_decode_run(self.h, 1, C.byref(left)), "wm_decode_run")
            it += 1
            cur = [self.tokens(b) for b in range(B)]
            stop = on_iteration([cur[b][seen[b]:] for b in range(B)])
            seen = [len(c) for c in cur]
            if left.value == 0 or stop is True:                   # the callback may end the run (host-side stopping criteria)
                break
        return [self.tokens(b) for b in range(B)]

    def tokens(self, stream: int) -> List[int]:
        cap = self.cfg.max_target_positions + 16
        buf = (C.c_int32 * cap)()
        n = C.c_int32(0)
        self._check(self.lib.wm_get_tokens(self.h, stream, buf, cap, C.byref(n)), "wm_get_tokens")
        return list(buf[: min(n.value, cap)])

    def stats(self) -> dict:
        s = WmStats()
        self._check(self.lib.wm_get_stats(self.h, C.byref(s)), "wm_get_stats")
        return dict(iterations=s.iterations, iterations_launched=s.iterations_launched, tokens_emitted=s.tokens_emitted,
                    accept_hist=list(s.accept_hist)[: self.cfg.medusa_num_heads + 1],
                    ms_logmel=s.ms_logmel, ms_encode=s.ms_encode, ms_decode=s.ms_decode,
                    graph_replays=s.graph_replays, schedule_steps=s.schedule_steps, sibling_hits=s.sibling_hits)

    def sync(self):
        self._check(self.lib.wm_sync(self.h), "wm_sync")

    # ---- taps -------------------------------------------------------------------------------
    def encoder_output(self, B: int) -> torch.Tensor:
        out = np.empty((B, self.cfg.max_source_positions, self.cfg.d_model), dtype=np.float32)
        self._check(self.lib.wm_get_encoder_output(self.h, B, out.ctypes.data_as(C.POINTER(C.c_float))), "wm_get_encoder_output")
        return torch.from_numpy(out)

    def encoder_output_cached(self, B: int) -> torch.Tensor:
        """The resident encoder pass's hidden state, fetched from the device once per encoder pass (api.forward's tuple return hands it
        out on every call of a per-token loop: one D2H copy of [B, n_ctx, d] per pass, not per token)."""
        c = getattr(self, "_enc_host", None)
        if c is None or c[0] is not self._enc_stamp or c[1] != B:
            self._enc_host = (self._enc_stamp, B, self.encoder_output(B))
        return self._enc_host[2]

    def cross_kv(self, kv_layer: int, stream: int, head: int):
        S = self.cfg.max_source_positions
        k = np.empty((S, 64), dtype=np.float32); v = np.empty((S, 64), dtype=np.float32)
        self._check(self.lib.wm_get_cross_kv(self.h, kv_layer, stream, head, k.ctypes.data_as(C.POINTER(C.c_float)),
                                             v.ctypes.data_as(C.POINTER(C.c_float))), "wm_get_cross_kv")
        return torch.from_numpy(k), torch.from_numpy(v)

    def forward_logits(self, tokens: Sequence[Sequence[int]], pos0: int, disable_medusa: bool) -> torch.Tensor:
        B, T = len(tokens), len(tokens[0])
        flat = _i32arr([t for row in tokens for t in row])
        n_out = 1 if disable_medusa else self.cfg.medusa_num_heads + 1
        out = np.empty((n_out, B, T, self.cfg.vocab_size), dtype=np.float32)
        # the pass overwrites cache rows pos0 .. pos0 + T: every EngineKVCache handle issued so far goes stale (api.forward hands the
        # caller a fresh one carrying the new stamp) — a later pass at a smaller position can no longer be followed by an old handle
        self._kv_stamp = object()
        self._check(self.lib.wm_forward_logits(self.h, B, flat, T, pos0, 1 if disable_medusa else 0,
                                               out.ctypes.data_as(C.POINTER(C.c_float))), "wm_forward_logits")
        return torch.from_numpy(out)

    def profile_layer_gemms(self, rows: int, reps: int = 50, kernel: int = 0):
        """hipEvent-timed decode GEMMs of decoder layer 0 at `rows` token rows: kernel 0 = all six of a layer, 1..6 = one of them
        (LN1+QKV, out-proj, LN2+cross-q, cross-out, LN3+FC1, FC2), 7 = vocabulary projection.  Returns (ms per repetition, weight bytes)."""
        ms, nbytes = C.c_float(0), C.c_double(0)
        self._check(self.lib.wm_profile_kernel(self.h, kernel, rows, reps, C.byref(ms), C.byref(nbytes)), "wm_profile_kernel")
        return ms.value, nbytes.value
