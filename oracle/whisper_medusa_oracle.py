"""CPU ORACLE for the Whisper-Medusa hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this module; the product (``whisper-medusa_amd/``) never does.

It is a plain-PyTorch (CPU, fp32) restatement of the reference algorithm
(aiola-lab/whisper-medusa @ 2025-03-04) with the HuggingFace generation glue collapsed,
following SURVEY.md Appendix A.  Each function cites the reference ``file:line`` it
follows (paths relative to ``/root/reference/whisper_medusa``) or, where the arithmetic
lives in the un-vendored dependency ``transformers==4.49.0`` (requirements.txt:7), the
installed copy ``HF:models/whisper/modeling_whisper.py`` (transformers 5.15; the layer
math is unchanged between the two).

PINNING.  The reference holds no tests, golden vectors or fixtures for this path
 and cannot run end-to-end on the installed transformers.  The
oracle is pinned instead against outputs of the reference's *own code* run in the
build container by ``oracle/make_golden.py``:
  * ``medusa_utils.generate_medusa_buffers / generate_candidates / evaluate_posterior``
    imported live from ``/root/reference`` (both acceptance branches);
  * the reference's own ``WhisperMedusaModel.forward()`` (Medusa-Linear) under import
    stubs, driven cache-free, single passes and a full decode loop;
  * HF 5.15 ``WhisperEncoder`` / ``WhisperDecoderLayer`` / ``WhisperFeatureExtractor`` /
    logits processors with shared seeded weights (also re-checked live in tests/).
Medusa-Block ``forward`` of the reference crashes as-is under transformers 5.x
(model.py:1364-1380, :1414); with the two shims of ``make_golden.py:build_ref`` the
reference's OWN ``_forward_medusa_block`` runs, and the Block path is pinned the same way
(tests/golden/reference_block_runs.npz: 12 decode runs, logits of every head to 2e-6).
Only ``resample_sinc_hann`` (torchaudio, not installed) is unpinned — see its docstring.

NUMERICS.  ``sim="fp32"`` is the reference's default dtype.  ``sim="bf16"`` mirrors
the engine's storage/rounding contract: parameters are
bf16-representable; in the ENCODER every GEMM input is rounded to bf16 (MFMA prefill);
(``act="f16"``: the decoder GEMM operands are ONE fp16 plane instead of the hi/lo pair below — wm_config.act_fp16)
in the DECODER GEMM/attention operands are carried as a bf16 hi/lo pair (x = hi + lo,
~17 mantissa bits, ``_rd``) so they are effectively unrounded; the self/cross KV cache
and the encoder output are stored in bf16; all accumulation / LayerNorm / softmax /
residual arithmetic is fp32.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

HEAD_DIM = 64


def _bf16(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).to(torch.float32)


# --------------------------------------------------------------------------------------
# F0  log-mel front end  (HF:models/whisper/feature_extraction_whisper.py:95-133, called at
#     eval_whisper_medusa.py:46-50 / README.md:129)
# --------------------------------------------------------------------------------------
def _hz_to_mel_slaney(f):
    f = np.asarray(f, dtype=np.float64)
    mel = 3.0 * f / 200.0
    min_log_hz, min_log_mel, logstep = 1000.0, 15.0, 27.0 / np.log(6.4)
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) * logstep, mel)


def _mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    f = 200.0 * m / 3.0
    min_log_hz, min_log_mel, logstep = 1000.0, 15.0, np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f)


def mel_filter_bank(n_mels: int = 80, n_fft: int = 400, sr: int = 16000, fmax: float = 8000.0) -> np.ndarray:
    """Slaney-scale, Slaney-normalised triangular filters, [n_fft//2+1, n_mels] float64
    (HF ``mel_filter_bank(norm='slaney', mel_scale='slaney')``, feature_extraction_whisper.py:95-103)."""
    n_freq = 1 + n_fft // 2
    fft_freqs = np.linspace(0.0, sr // 2, n_freq)
    mel_pts = np.linspace(_hz_to_mel_slaney(0.0), _hz_to_mel_slaney(fmax), n_mels + 2)
    hz_pts = _mel_to_hz_slaney(mel_pts)
    fdiff = np.diff(hz_pts)
    slopes = hz_pts[None, :] - fft_freqs[:, None]
    down = -slopes[:, :-2] / fdiff[:-1]
    up = slopes[:, 2:] / fdiff[1:]
    fb = np.maximum(0.0, np.minimum(down, up))
    enorm = 2.0 / (hz_pts[2: n_mels + 2] - hz_pts[:n_mels])
    return fb * enorm[None, :]


def resample_sinc_hann(wave: np.ndarray, orig_freq: int, new_freq: int, lowpass_filter_width: int = 6,
                       rolloff: float = 0.99) -> np.ndarray:
    """``torchaudio.transforms.Resample(orig_freq, new_freq)`` with its defaults, applied to float32 ``[..., n]``.

    Call sites in the reference: README.md:120-125, eval_whisper_medusa.py:41-45.  The arithmetic lives in
    torchaudio==2.2.2 (requirements.txt:6; functional/functional.py ``_get_sinc_resample_kernel`` /
    ``_apply_sinc_resample_kernel``), which is neither under /root/reference nor installed here: this restates the
    published algorithm with the same torch primitives in the same order — **parity unpinned** for this function
    (tests check it against independent properties and scipy's polyphase resampler instead).
    Details that matter for bit-level agreement with torchaudio: the tap grid ``idx`` is float64, the phase offset
    ``arange(0, -new, -1) / new`` is a float32 quotient that is then promoted, the kernel is rounded to float32 last,
    and the convolution is an fp32 ``conv1d`` with stride ``orig`` over the clip padded by (width, width + orig)."""
    import math
    g = math.gcd(int(orig_freq), int(new_freq))
    x = torch.from_numpy(np.ascontiguousarray(wave, dtype=np.float32))
    if orig_freq == new_freq:
        return x.numpy()
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base_freq = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base_freq)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = torch.arange(0, -new, -1)[:, None, None] / new + idx
    t = t * base_freq
    t = t.clamp(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    scale = base_freq / orig
    kernels = torch.where(t == 0, torch.tensor(1.0).to(t), t.sin() / t)
    kernels = (kernels * (window * scale)).to(torch.float32)
    shape = x.shape
    x2 = x.reshape(-1, shape[-1])
    num_wavs, length = x2.shape
    x2 = torch.nn.functional.pad(x2, (width, width + orig))
    res = torch.nn.functional.conv1d(x2[:, None], kernels, stride=orig)
    res = res.transpose(1, 2).reshape(num_wavs, -1)
    target_length = int(math.ceil(new * length / orig))
    res = res[..., :target_length]
    return res.reshape(shape[:-1] + res.shape[-1:]).numpy()


def downmix_mono(wave: np.ndarray) -> np.ndarray:
    """``input_speech.mean(dim=0)`` for a multi-channel clip [channels, n] (reference README.md:121-122)."""
    return torch.from_numpy(np.ascontiguousarray(wave, dtype=np.float32)).mean(dim=0).numpy()


def log_mel(wav: np.ndarray, n_mels: int = 80, n_samples: int = 480000) -> np.ndarray:
    """wav [n] float -> [n_mels, n_samples/160] float32.  Pad/trim to ``n_samples``, reflect-pad
    STFT (n_fft 400, hop 160, periodic Hann), power, drop the last frame, mel, log10, clamp
    to max-8, (x+4)/4.  (feature_extraction_whisper.py:105-133)"""
    n_fft, hop = 400, 160
    w = np.zeros(n_samples, dtype=np.float64)
    n = min(len(wav), n_samples)
    w[:n] = np.asarray(wav[:n], dtype=np.float64)
    w = np.pad(w, (n_fft // 2, n_fft // 2), mode="reflect")
    n_frames = 1 + (len(w) - n_fft) // hop
    idx = np.arange(n_fft)[None, :] + hop * np.arange(n_frames)[:, None]
    window = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n_fft) / n_fft)       # periodic Hann
    spec = np.fft.rfft(w[idx] * window[None, :], axis=1)                      # [frames, 201]
    power = (spec.real ** 2 + spec.imag ** 2)[:-1]                            # drop last frame
    mel = power @ mel_filter_bank(n_mels, n_fft)                              # [frames-1, n_mels]
    logs = np.log10(np.maximum(mel, 1e-10))
    logs = np.maximum(logs, logs.max() - 8.0)
    return np.ascontiguousarray(((logs + 4.0) / 4.0).T, dtype=np.float32)


# --------------------------------------------------------------------------------------
# F7  logits processors  (model.py:1168-1207 build order; HF generation/logits_process.py
#     :1678-1775 exp-decay, :1851-1866 begin-suppress, :1898-1906 suppress)
# --------------------------------------------------------------------------------------
def process_logits(scores: torch.Tensor, cur_len: int, gp) -> torch.Tensor:
    """Apply the processor list to rows ``scores [R, V]``.  Every row sees the SAME
    ``cur_len`` = ``input_ids.shape[-1]`` before the update (model.py:653-665,689-694).
    Order = HF default processors (exp-decay) then the Whisper ones the reference prepends
    to the user list: begin-suppress, suppress (model.py:1177-1199)."""
    s = scores.clone()
    if gp.exp_decay is not None:
        start = gp.exp_decay[0] + len(gp.prompt)          # regulation_start, logits_process.py:1752
        if cur_len > start:
            e = gp.eos_token_id
            s[:, e] = s[:, e] + s[:, e].abs() * (pow(gp.exp_decay[1], cur_len - start) - 1.0)
    if gp.begin_suppress_tokens and cur_len == gp.begin_index:
        s[:, list(gp.begin_suppress_tokens)] = -float("inf")
    if gp.suppress_tokens:
        s[:, list(gp.suppress_tokens)] = -float("inf")
    return s


# --------------------------------------------------------------------------------------
# F11  accept / reject for a chain tree  (medusa_utils.py:526-588 with n_candidates == 1)
# --------------------------------------------------------------------------------------
def evaluate_posterior_chain(v: torch.Tensor, cand: torch.Tensor, gp) -> Tuple[int, dict]:
    """``v [K+1, V]`` processed verify logits, ``cand [K+1]`` candidate tokens.  Returns the
    accept length ``a`` (number of leading accepted candidates c_1..c_K)."""
    K = cand.numel() - 1
    dbg = {}
    if gp.accept_mode == 0 or gp.temperature == 0:                       # :547-560
        ok = (cand[1:] == torch.argmax(v[:-1], dim=-1))
    else:                                                                # :562-577
        p = torch.softmax(v[:-1] / gp.temperature, dim=-1)
        p_c = torch.gather(p, -1, cand[1:].unsqueeze(-1)).squeeze(-1)
        H = -torch.sum(p * torch.log(p + 1e-5), dim=-1)
        thr = torch.minimum(torch.full_like(H, gp.posterior_threshold), torch.exp(-H) * gp.posterior_alpha)
        ok = p_c > thr
        dbg = dict(p_c=p_c, H=H, thr=thr)
    a = int(torch.cumprod(ok.int(), dim=0).sum().item())
    return a, dbg


# --------------------------------------------------------------------------------------
# Candidate trees (medusa_choices with top-k > 1)
# --------------------------------------------------------------------------------------
def medusa_buffers(choices: List[int]) -> dict:
    """Restatement of generate_medusa_buffers (medusa_utils.py:305-421), same steps in numpy: tree_indices (:330-341),
    attention mask rows (:343-358), position ids (:360-363), retrieve indices (:365-379)."""
    c = np.asarray(choices, dtype=np.int64)
    cumprod, cumsum = np.cumprod(c), np.cumsum(c)
    n = int(cumprod.sum())
    mask = np.eye(n, dtype=np.float32)
    tree_indices: List[int] = []
    prev_sum, prev_prod = 0, 1
    for i in range(len(c)):
        tree_indices += list(np.tile(np.arange(prev_sum, cumsum[i]), prev_prod))
        prev_sum, prev_prod = int(cumsum[i]), int(cumprod[i])
    prev = -1
    for i in range(len(c)):
        cur = int(cumprod[:i].sum())
        if prev != -1:
            parent = np.repeat(np.arange(prev, cur), c[i])
            mask[cur: cur + len(parent)] += mask[parent]
        prev = cur
    depth: List[int] = []
    for i in range(len(c)):
        depth += [i] * int(cumprod[i])
    n_paths = int(np.prod(c))
    retrieve = np.zeros((n_paths, len(c)), dtype=np.int64)
    prev = 0
    for i in range(len(c)):
        cur = int(cumprod[: i + 1].sum())
        retrieve[:, i] = np.repeat(np.arange(prev, cur), n_paths // (cur - prev))
        prev = cur
    anc = [int(sum(1 << j for j in range(n) if mask[m, j] > 0)) for m in range(n)]
    return dict(tree_indices=[int(x) for x in tree_indices], depth=depth, mask=mask, anc=anc, retrieve=retrieve,
                n_nodes=n, n_paths=n_paths)


def tree_candidates(z: torch.Tensor, choices: List[int]) -> Tuple[torch.Tensor, torch.Tensor]:
    """generate_candidates (medusa_utils.py:424-458) for processed last-row logits ``z [K+1, V]``: flat list = base argmax +
    top-c_k of head k; returns (cartesian-product candidates [n_paths, K+1], flat list)."""
    flat = [torch.argmax(z[0]).unsqueeze(0)]
    for k in range(1, len(choices)):
        flat.append(torch.topk(z[k], int(choices[k])).indices)
    cands = torch.cartesian_prod(*flat)
    if cands.dim() == 1:
        cands = cands.unsqueeze(0)
    return cands, torch.cat(flat)


def evaluate_posterior_multi(v: torch.Tensor, cands: torch.Tensor, gp) -> Tuple[int, int, dict]:
    """evaluate_posterior (medusa_utils.py:526-588) with several candidate paths: ``v [n_paths, K+1, V]`` processed verify
    logits along each path, ``cands [n_paths, K+1]``.  Returns (best path, accept length, debug)."""
    if gp.accept_mode == 0 or gp.temperature == 0:                       # :547-560
        ok = (cands[:, 1:] == torch.argmax(v[:, :-1], dim=-1)).int()
        al = torch.cumprod(ok, dim=1).sum(dim=1)
        a = int(al.max())
        best = 0 if a == 0 else int(torch.argmax(al))
        return best, a, dict(accept_lengths=al)
    p = torch.softmax(v[:, :-1] / gp.temperature, dim=-1)                # :562-588
    p_c = torch.gather(p, -1, cands[:, 1:].unsqueeze(-1)).squeeze(-1)
    H = -torch.sum(p * torch.log(p + 1e-5), dim=-1)
    thr = torch.minimum(torch.full_like(H, gp.posterior_threshold), torch.exp(-H) * gp.posterior_alpha)
    ok = (p_c > thr).int()
    al = torch.cumprod(ok, dim=1).sum(dim=1)
    a = int(al.max())
    lik = None
    if a == 0:
        best = 0
    else:
        idx = torch.where(al == a)[0]
        lik = torch.sum(torch.log(p_c[idx, :a]), dim=-1)
        best = int(idx[torch.argmax(lik)])
    return best, a, dict(accept_lengths=al, p_c=p_c, thr=thr, H=H, lik=lik)


@dataclass
class DecodeResult:
    ids: List[int]                      # prompt + emitted tokens, post-EOS overwrite applied
    new_tokens: List[int]               # ids[P:] up to (excluding) the first EOS  -> the parity object
    accept_lengths: List[int] = field(default_factory=list)
    n_iters: int = 0
    trace: List[dict] = field(default_factory=list)
    verified: int = 0                   # decode_tree(engine_ids=...): ids confirmed identical to the engine's
    tie: Optional[dict] = None          # ... and the first differing iteration's decision margins
    ties: List[dict] = field(default_factory=list)   # decode_tree(engine_groups=...): every iteration whose outcome differed from the engine's
    sibling_hits: int = 0               # decode(siblings=S): accept length 0 AND the next root among head 1's top-2 .. top-(S+1) (wm_config.sibling_rows)


class Oracle:
    """Functional Whisper-Medusa over a plain state dict (reference key layout)."""

    def __init__(self, cfg, sd: Dict[str, torch.Tensor], sim: str = "fp32", dec_fp8: bool = False, enc_fp8: bool = False, act: Optional[str] = None,
                 xkv_fp8: bool = False):
        if act is None:         # the contract the engine defaults to in this process (whisper_medusa.engine.default_act_fp16: WM_ACT)
            import os as _os
            act = "f16" if _os.environ.get("WM_ACT", "f16").lower() in ("f16", "fp16") else "hilo"
        assert sim in ("fp32", "bf16") and act in ("hilo", "f16")
        self.cfg, self.sim = cfg, sim
        # decoder GEMM operand of the engine contract: "hilo" = bf16 hi + bf16 lo (two planes, ~17 bits), "f16" = ONE fp16 plane (11 bits; the
        # decoder matrices are then held in fp16 — exact from bf16 for |w| >= 2^-17 —, wm_config.act_fp16).  Attention operands (q, P) stay hi / lo.
        self.act = act
        # BASELINE.json configs[4] "fp8 MFMA" (not a reference feature): the encoder GEMMs whose operand is a LayerNorm output
        # (q/k/v, fc1) and the cross-K/V projection multiply e4m3 by e4m3: the LayerNorm output row x is quantised as
        # q_x = rne_e4m3(x / s_x), s_x = max|x| / 448 (1 for a zero row), the weight row likewise, and the product is
        # ((q_x . q_w) * s_x) * s_w in fp32.  The cross-K/V projection quantises the STORED (bf16) encoder output.
        self.enc_fp8 = enc_fp8
        # wm_config.cross_kv_fp8 (BASELINE configs[4]; not a reference feature): the decoder reads the cross-K/V from an fp8 e4m3 copy with one
        # scale per (kv layer, head) for K and one for V: scale = max|x| / 448 over the head's S x 64 stored (bf16) values, q = rne_e4m3(x / scale)
        self.xkv_fp8 = xkv_fp8
        self.w8: Dict[str, Tuple[torch.Tensor, torch.Tensor]] = {}
        self.sd = {k: v.detach().to(torch.float32).cpu() for k, v in sd.items()}
        self.H = cfg.d_model // HEAD_DIM
        if "whisper_model.proj_out.weight" not in self.sd:
            self.sd["whisper_model.proj_out.weight"] = self.sd["whisper_model.model.decoder.embed_tokens.weight"]
        # BASELINE.json configs[4] (fp8 weights; not a reference feature): the self-attention q/k/v/out, cross-attention q/out
        # and MLP matrices of every decoder layer (and of the Medusa block) are W[n][k] = scale[n] * q[n][k], q in fp8 e4m3
        # (round to nearest even), scale[n] = max_k |W[n][k]| / 448; a product is (x @ q^T) * scale.
        self.q8: Dict[str, Tuple[torch.Tensor, torch.Tensor]] = {}
        if dec_fp8:
            for k, w in self.sd.items():
                layer = k.startswith("whisper_model.model.decoder.layers.") or k.startswith("medusa_block.")
                if layer and k.endswith(".weight") and w.dim() == 2 and not (k.endswith("encoder_attn.k_proj.weight") or k.endswith("encoder_attn.v_proj.weight")):
                    amax = w.abs().amax(dim=1)
                    scale = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
                    q = (w / scale[:, None]).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(torch.float32)
                    self.q8[k[: -len(".weight")]] = (q, scale)

    # ---- small helpers --------------------------------------------------------------
    def _r(self, x):           # rounding point of the engine contract (encoder operands, KV / encoder-output storage)
        return _bf16(x) if self.sim == "bf16" else x

    def _rd(self, x):          # decoder operands: bf16 hi + bf16 lo
        if self.sim != "bf16":
            return x
        hi = _bf16(x)
        return hi + _bf16(x - hi)

    def _rg(self, x):          # decoder GEMM operands
        if self.sim == "bf16" and self.act == "f16":
            return x.to(torch.float16).to(torch.float32)
        return self._rd(x)

    def _lin(self, x, prefix, bias=True, dec=False):
        xr = self._rg(x) if dec else self._r(x)
        if prefix in self.q8:
            q, scale = self.q8[prefix]
            y = (xr @ q.t()) * scale
        else:
            y = xr @ self.sd[prefix + ".weight"].t()
        if bias and (prefix + ".bias") in self.sd:
            y = y + self.sd[prefix + ".bias"]
        return y

    def _lin_ln(self, h, ln_prefix, prefix, bias=True):
        """linear(LayerNorm(h)) of a decoder layer (HF:modeling_whisper.py:416-505, pre-LN).  fp32 / hi-lo contract: exactly that.  fp16
        single-plane contract (act="f16"): the engine's rounding point — it multiplies the operand fp16(gamma o h) and applies the row statistics
        to the product (csrc/wm_common.h "LayerNorm folded into the GEMM it feeds"):  rstd (W fp16(gamma o h) - mean W gamma) + (b + W beta),
        the same algebra with the rounding where the engine has it (hi / lo carries ~17 bits either way: no visible rounding point there)."""
        if not (self.sim == "bf16" and self.act == "f16"):
            return self._lin(self._ln(h, ln_prefix), prefix, bias=bias, dec=True)
        g, b = self.sd[ln_prefix + ".weight"], self.sd[ln_prefix + ".bias"]
        if prefix in self.q8:
            q, scale = self.q8[prefix]
            W = q * scale[:, None]
        else:
            W = self.sd[prefix + ".weight"]
        z = (h * g).to(torch.float16).to(torch.float32)
        mean = h.mean(dim=-1, keepdim=True)
        var = ((h * h).mean(dim=-1, keepdim=True) - mean * mean).clamp_min(0.0)
        rstd = torch.rsqrt(var + 1e-5)
        y = rstd * (z @ W.t() - mean * (W @ g)[None, :]) + (W @ b)[None, :]
        if bias and (prefix + ".bias") in self.sd:
            y = y + self.sd[prefix + ".bias"]
        return y

    @staticmethod
    def _q8_rows(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        amax = x.abs().amax(dim=1)
        scale = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
        q = (x / scale[:, None]).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(torch.float32)
        return q, scale

    @staticmethod
    def _q8_heads(x: torch.Tensor) -> torch.Tensor:
        """[H, S, 64] -> its e4m3 image with one scale per head (csrc/wm_encoder.hip k_xkv_quant), dequantised."""
        amax = x.abs().amax(dim=(1, 2))
        scale = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))[:, None, None]
        return (x / scale).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(torch.float32) * scale

    def _lin8(self, x, prefix, bias=True):
        """fp8-MFMA linear: e4m3 activations (row scale) x e4m3 weights (row scale), fp32 accumulation."""
        if prefix not in self.w8:
            self.w8[prefix] = self._q8_rows(self.sd[prefix + ".weight"])
        wq, ws = self.w8[prefix]
        xq, xs = self._q8_rows(x)
        y = ((xq @ wq.t()) * xs[:, None]) * ws[None, :]
        if bias and (prefix + ".bias") in self.sd:
            y = y + self.sd[prefix + ".bias"]
        return y

    def _ln(self, x, prefix):
        return F.layer_norm(x, (x.shape[-1],), self.sd[prefix + ".weight"], self.sd[prefix + ".bias"], 1e-5)

    def _heads(self, x):       # [T, d] -> [H, T, 64]
        return x.view(x.shape[0], self.H, HEAD_DIM).transpose(0, 1)

    def _attend(self, q, k, v, mask=None, round_p=False, dec=False):
        """q [H,T,64] (already scaled), k/v [H,S,64] -> [T, d].  Softmax in fp32
        (HF:modeling_whisper.py:214-238)."""
        w = q @ k.transpose(1, 2)
        if mask is not None:
            w = w + mask
        w = torch.softmax(w, dim=-1)
        if round_p:
            w = self._r(w)
        if dec:
            w = self._rd(w)
        o = w @ v
        return o.transpose(0, 1).reshape(q.shape[1], -1)

    # ---- F1 encoder (HF:modeling_whisper.py:592-646; layer :360-413) --------------------
    def encode(self, feats: torch.Tensor) -> torch.Tensor:
        """feats [n_mels, 2*S] fp32 -> encoder output [S, d]."""
        sd, p = self.sd, "whisper_model.model.encoder"
        x = self._r(feats.to(torch.float32))[None]
        x = F.gelu(F.conv1d(x, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=1))
        x = F.gelu(F.conv1d(self._r(x), sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], stride=2, padding=1))
        h = x[0].t() + sd[p + ".embed_positions.weight"]
        for i in range(self.cfg.encoder_layers):
            lp = f"{p}.layers.{i}"
            xn = self._ln(h, lp + ".self_attn_layer_norm")
            lin = self._lin8 if self.enc_fp8 else self._lin
            q = self._r(lin(xn, lp + ".self_attn.q_proj") * HEAD_DIM ** -0.5)
            k = self._r(lin(xn, lp + ".self_attn.k_proj", bias=False))
            v = self._r(lin(xn, lp + ".self_attn.v_proj"))
            a = self._attend(self._heads(q), self._heads(k), self._heads(v), round_p=True)
            h = h + self._lin(a, lp + ".self_attn.out_proj")
            xn = self._ln(h, lp + ".final_layer_norm")
            h = h + self._lin(F.gelu(lin(xn, lp + ".fc1")), lp + ".fc2")
        return self._r(self._ln(h, p + ".layer_norm"))

    # ---- F2 cross-KV projection (HF:modeling_whisper.py:322-335; Block: model.py:1382-1393) -----
    def _kv_layer_prefixes(self) -> List[str]:
        ps = [f"whisper_model.model.decoder.layers.{i}" for i in range(self.cfg.decoder_layers)]
        if self.cfg.is_block:
            ps.append("medusa_block")                     # cache slot L, model.py:248-256
        return ps

    def cross_kv(self, enc: torch.Tensor):
        out = []
        lin = self._lin8 if self.enc_fp8 else self._lin
        for lp in self._kv_layer_prefixes():
            k = self._r(lin(enc, lp + ".encoder_attn.k_proj", bias=False))
            v = self._r(lin(enc, lp + ".encoder_attn.v_proj"))
            kh, vh = self._heads(k), self._heads(v)
            if self.xkv_fp8:
                kh, vh = self._q8_heads(kh), self._q8_heads(vh)
            out.append((kh, vh))
        return out

    # ---- F3 decoder layer (HF:modeling_whisper.py:416-505) ------------------------------
    def _dec_layer(self, lp, h, slot, st, T):
        kv_len = st["kv_len"]
        ln1 = lp + ".self_attn_layer_norm"
        q = self._rd(self._lin_ln(h, ln1, lp + ".self_attn.q_proj") * HEAD_DIM ** -0.5)
        k = self._r(self._lin_ln(h, ln1, lp + ".self_attn.k_proj", bias=False))
        v = self._r(self._lin_ln(h, ln1, lp + ".self_attn.v_proj"))
        kc, vc = st["self_kv"][slot]
        kc = torch.cat([kc[:, :kv_len], self._heads(k)], dim=1)      # contiguous cache: rows kv_len.. overwritten
        vc = torch.cat([vc[:, :kv_len], self._heads(v)], dim=1)
        st["self_kv"][slot] = (kc, vc)
        mask = torch.full((T, kv_len + T), 0.0)
        if st.get("_anc") is None:
            mask[:, kv_len:] = torch.triu(torch.full((T, T), -float("inf")), diagonal=1)
        else:                                                        # candidate tree: node m sees itself and its ancestors
            for m_, bits in enumerate(st["_anc"]):                   # (the rows of medusa_attn_mask, medusa_utils.py:363-376)
                for n_ in range(T):
                    if not (bits >> n_) & 1:
                        mask[m_, kv_len + n_] = -float("inf")
        a = self._attend(self._heads(q), kc, vc, mask, dec=True)
        h = h + self._lin(a, lp + ".self_attn.out_proj", dec=True)
        q = self._rd(self._lin_ln(h, lp + ".encoder_attn_layer_norm", lp + ".encoder_attn.q_proj") * HEAD_DIM ** -0.5)
        kx, vx = st["cross_kv"][slot]
        a = self._attend(self._heads(q), kx, vx, dec=True)
        h = h + self._lin(a, lp + ".encoder_attn.out_proj", dec=True)
        h = h + self._lin(F.gelu(self._lin_ln(h, lp + ".final_layer_norm", lp + ".fc1")), lp + ".fc2", dec=True)
        return h

    def new_state(self, enc: torch.Tensor) -> dict:
        empty = torch.zeros(self.H, 0, HEAD_DIM)
        return dict(cross_kv=self.cross_kv(enc), kv_len=0,
                    self_kv=[(empty, empty) for _ in range(self.cfg.n_kv_layers)])

    def _res_head(self, k, x):
        """MedusaResBlock: x + SiLU(W x + b)  (model.py:180-210)."""
        return x + F.silu(self._lin(x, f"medusa_heads.{k}.0.linear", dec=True))

    def _vocab(self, y):
        return self._rg(y) @ self.sd["whisper_model.proj_out.weight"].t()      # tied, no bias (model.py:1277)

    def decoder_pass(self, st: dict, tokens: List[int], pos0: int, disable_medusa: bool,
                     last_only: bool = False, depth: Optional[List[int]] = None, anc: Optional[List[int]] = None) -> torch.Tensor:
        """One forward of WhisperMedusaModel (model.py:1223-1347 / 1349-1417) over ``tokens`` at
        positions ``pos0..``; appends T rows to every self-KV slot at ``st['kv_len']``.
        Returns stacked logits [n_heads_out, T, V] (n_heads_out = 1 if disable_medusa else K+1).
        Does NOT advance ``kv_len`` (the caller applies the F13 keep rule).
        Candidate-tree verify pass: ``depth`` gives every node's position offset (medusa_position_ids, medusa_utils.py:378-384)
        and ``anc`` its ancestor bitmask (medusa_attn_mask) — what tree_decoding (medusa_utils.py:494-521) would feed the decoder
        if the reference applied its own mask."""
        sd, cfg, p = self.sd, self.cfg, "whisper_model.model.decoder"
        T = len(tokens)
        ids = torch.tensor(tokens, dtype=torch.long)
        pos = torch.arange(pos0, pos0 + T) if depth is None else pos0 + torch.tensor(depth, dtype=torch.long)
        st["_anc"] = anc
        h = sd[p + ".embed_tokens.weight"][ids] + sd[p + ".embed_positions.weight"][pos]
        for i in range(cfg.decoder_layers):
            h = self._dec_layer(f"{p}.layers.{i}", h, i, st, T)
        hf = self._ln(h, p + ".layer_norm")                       # post-final-LN state (model.py:1262)
        rows = hf[-1:] if last_only else hf
        K = cfg.medusa_num_heads
        if not cfg.is_block:                                      # Medusa-Linear, model.py:1274-1284
            n = 1 if disable_medusa else K + 1
            return torch.stack([self._vocab(self._res_head(k, rows)) for k in range(n)])
        out = [self._vocab(rows)]                                 # Medusa-Block base logits, model.py:1287
        g = self._dec_layer("medusa_block", hf, cfg.decoder_layers, st, T)   # on post-LN state, model.py:1382-1393
        if not disable_medusa:                                    # model.py:1410-1417
            g_rows = g[-1:] if last_only else g
            out += [self._vocab(self._res_head(k, g_rows)) for k in range(K)]
        return torch.stack(out)

    @torch.no_grad()
    def detect_language(self, enc: torch.Tensor, lang_ids: List[int], start_token: Optional[int] = None) -> int:
        """HF WhisperGenerationMixin.detect_language (generation_whisper.py: one decoder step on <|startoftranscript|>, every
        non-language logit masked to -inf, argmax) — what the reference's generate() reaches through _retrieve_init_tokens
        when `language` is None (model.py:1519-1537).  With the Medusa model the first of the stacked heads is the base head.
        Returns the winning language TOKEN ID."""
        tok = self.cfg.decoder_start_token_id if start_token is None else start_token
        z = self.decoder_pass(self.new_state(enc), [tok], 0, disable_medusa=True)[0, 0]
        mask = torch.full_like(z, -float("inf"))
        mask[list(lang_ids)] = 0.0
        return int(torch.argmax(z + mask))

    # ---- the decode loop (model.py:634-810) --------------------------
    @torch.no_grad()
    def decode(self, enc: torch.Tensor, gp, trace: bool = False, max_iters: Optional[int] = None, siblings: int = 0) -> DecodeResult:
        """The reference loop (model.py:634-793) over a candidate chain.  `siblings` = S > 0 additionally COUNTS (it changes no token) the iterations
        in which the engine's sibling rows (include/wm.h: wm_config.sibling_rows) save the next base pass: the chain accepts nothing and the next
        root — argmax v_0, model.py:710-713 — is one of head 1's top-2 .. top-(S+1) processed tokens."""
        if gp.vanilla:
            return self._decode_vanilla(enc, gp, max_iters)
        cfg = self.cfg
        if [int(x) for x in cfg.medusa_choices] != [1] * (cfg.medusa_num_heads + 1):
            return self.decode_tree(enc, gp, max_iters=max_iters)
        K, P, eos = cfg.medusa_num_heads, len(gp.prompt), gp.eos_token_id
        st = self.new_state(enc)
        ids = list(gp.prompt)
        res = DecodeResult(ids=[], new_tokens=[])
        finished = False
        while not finished:
            L, kv = len(ids), st["kv_len"]
            # (a) base pass over ids[kv:L]                                         model.py:639-648
            z = self.decoder_pass(st, ids[kv:L], kv, disable_medusa=False, last_only=True)[:, 0]   # [K+1, V]
            st["kv_len"] = L
            z = process_logits(z, L, gp)                                          # model.py:653-665
            cand = torch.argmax(z, dim=-1)                                        # medusa_utils.py:446-458 (top-1 chain)
            # (d) verify pass over the K+1 candidates at positions L..L+K         medusa_utils.py:494-521
            v = self.decoder_pass(st, cand.tolist(), L, disable_medusa=True)[0]   # [K+1, V]
            v = process_logits(v, L, gp)                                          # model.py:689-694 (same L for all rows)
            a, dbg = evaluate_posterior_chain(v, cand, gp)                        # model.py:697-703
            if a == 0:                                                            # model.py:710-713, medusa_utils.py:636-641
                emit = [int(cand[0]), int(torch.argmax(v[0]))]
                if siblings > 0 and int(torch.argmax(v[0])) in torch.topk(z[1], siblings + 1).indices[1:].tolist():
                    res.sibling_hits += 1
                st["kv_len"] = L + 1                                              # model.py:388-392 keep [:a+1]
            else:
                emit = [int(t) for t in cand[: a + 1]]
                st["kv_len"] = L + a                                              # keep [:a]
            ids += emit
            res.accept_lengths.append(a)
            res.n_iters += 1
            if trace:
                res.trace.append(dict(L=L, cand=cand.clone(), z_top=z.max(dim=-1).values.clone(),
                                      v=v.clone(), a=a, emit=list(emit), **dbg))
            L = len(ids)
            # (j) stop rules                                                      model.py:774-793
            finished = (eos in emit) or (L >= gp.max_length) or (L + K >= gp.hard_max_length)
            if max_iters is not None and res.n_iters >= max_iters:
                break
        return self._finish(res, ids, P, eos)

    # ---- the decode loop over a candidate tree ----------------------------------------------------------
    @torch.no_grad()
    def decode_tree(self, enc: torch.Tensor, gp, choices: Optional[List[int]] = None, engine_ids: Optional[List[int]] = None,
                    tol_logit: float = 5e-4, tol_rel_p: float = 2e-3, max_iters: Optional[int] = None,
                    engine_groups: Optional[List[List[int]]] = None) -> DecodeResult:
        """model.py:634-793 with ``medusa_choices = [1, c_1, .., c_K]``: candidates = cartesian product of the base argmax and
        every head's top-c_k (medusa_utils.py:424-458), ONE verify pass over the tree's nodes at positions L + depth with the
        ancestor mask (tree_decoding :494-521 — the reference builds that mask and never passes it on; a node's logits are
        only meaningful with it, so the mask is part of this restatement), accept/reject over all paths (:526-588), keep the
        K/V rows of the chosen path (model.py:378-402).
        With ``engine_ids`` the run doubles as a parity check that survives numerical ties: res.trace receives, for the first
        iteration whose strict outcome differs from the engine's tokens, the smallest decision margin of that iteration
        (``tie`` entry) — a divergence is a numerical tie iff that margin is below the tolerances — and the walk stops there
        (``res.verified`` = number of ids confirmed identical).
        With ``engine_groups`` (the tokens the engine emitted in each of its iterations, from its streaming callback) the walk does NOT
        stop: every iteration is run from the ENGINE's prefix (teacher forcing) and compared with the engine's group; a mismatch is
        recorded in ``res.ties`` with that iteration's margins and the walk continues from the engine's tokens (their K/V rows are
        recomputed by the next base pass), so every iteration after a tie is still checked."""
        cfg = self.cfg
        choices = [int(x) for x in (choices if choices is not None else cfg.medusa_choices)]
        K, P, eos = cfg.medusa_num_heads, len(gp.prompt), gp.eos_token_id
        assert len(choices) == K + 1 and not gp.vanilla
        tb = medusa_buffers(choices)
        retrieve = torch.from_numpy(tb["retrieve"])
        st = self.new_state(enc)
        ids = list(gp.prompt)
        res = DecodeResult(ids=[], new_tokens=[])
        res.verified, res.tie, res.ties = len(ids), None, []
        it_no = 0
        while True:
            if engine_groups is not None and it_no >= len(engine_groups):
                break
            L, kv = len(ids), st["kv_len"]
            z = self.decoder_pass(st, ids[kv:L], kv, disable_medusa=False, last_only=True)[:, 0]
            st["kv_len"] = L
            z = process_logits(z, L, gp)
            cands, flat = tree_candidates(z, choices)
            tree_tok = flat[torch.tensor(tb["tree_indices"])]                      # medusa_utils.py:457
            vt = self.decoder_pass(st, tree_tok.tolist(), L, disable_medusa=True, depth=tb["depth"], anc=tb["anc"])[0]
            st["_anc"] = None
            vt = process_logits(vt, L, gp)                                         # every node sees the same cur_len
            v = vt[retrieve]                                                       # [n_paths, K+1, V]   (:518-521)
            best, a, dbg = evaluate_posterior_multi(v, cands, gp)
            if a == 0:
                emit = [int(cands[best, 0]), int(torch.argmax(v[best, 0]))]
                keep = [0]
            else:
                emit = [int(t) for t in cands[best, : a + 1]]
                keep = [int(n) for n in retrieve[best, :a]]
            if engine_groups is not None:
                grp = [int(t) for t in engine_groups[it_no]]
                it_no += 1
                if grp != emit:
                    res.ties.append(dict(L=L, oracle_emit=emit, engine=grp,
                                         margin=self._tree_margins(z, vt, v, cands, choices, retrieve, best, a, dbg, gp, tol_logit, tol_rel_p)))
                    ids += grp                       # follow the engine; the provisional rows are dropped, the next base pass recomputes them
                    for slot, (kc, vc) in enumerate(st["self_kv"]):
                        st["self_kv"][slot] = (kc[:, :L], vc[:, :L])
                    st["kv_len"] = L
                    res.n_iters += 1
                    L = len(ids)
                    if (eos in grp) or (L >= gp.max_length) or (L + K >= gp.hard_max_length):
                        break
                    continue
            if engine_ids is not None:
                tail = engine_ids[L: L + len(emit)]
                if tail != emit[: len(tail)] or not tail:
                    res.tie = dict(L=L, oracle_emit=emit, engine=engine_ids[L: L + K + 2],
                                   margin=self._tree_margins(z, vt, v, cands, choices, retrieve, best, a, dbg, gp, tol_logit, tol_rel_p))
                    break
            # keep the chosen path's K/V rows (select_indices, model.py:378-392)
            rows = torch.tensor([L + n for n in keep], dtype=torch.long)
            for slot, (kc, vc) in enumerate(st["self_kv"]):
                st["self_kv"][slot] = (torch.cat([kc[:, :L], kc[:, rows]], dim=1), torch.cat([vc[:, :L], vc[:, rows]], dim=1))
            st["kv_len"] = L + len(keep)
            ids += emit
            res.verified = len(ids)
            res.accept_lengths.append(a)
            res.n_iters += 1
            L = len(ids)
            if (eos in emit) or (L >= gp.max_length) or (L + K >= gp.hard_max_length):
                break
            if max_iters is not None and res.n_iters >= max_iters:
                break
        return self._finish(res, ids, P, eos)

    @staticmethod
    def _tree_margins(z, vt, v, cands, choices, retrieve, best, a, dbg, gp, tol_logit, tol_rel_p) -> dict:
        """Smallest margins of one tree iteration's decisions, in units of their tolerance (< 1: a numerical tie is possible)."""
        m = {}
        top = torch.topk(z[0], 2).values
        m["base argmax"] = float(top[0] - top[1]) / tol_logit
        for k in range(1, len(choices)):
            t = torch.topk(z[k], int(choices[k]) + 1).values
            m[f"head {k} top-{choices[k]}"] = float((t[:-1] - t[1:]).min()) / tol_logit
        if gp.accept_mode == 0 or gp.temperature == 0:
            t2 = torch.topk(vt, 2, dim=-1).values
            m["verify argmax"] = float((t2[:, 0] - t2[:, 1]).min()) / tol_logit
        else:
            rel = ((dbg["p_c"] - dbg["thr"]).abs() / dbg["thr"])
            m["p_c vs thr"] = float(rel.min()) / tol_rel_p
            if dbg.get("lik") is not None and dbg["lik"].numel() > 1:
                lk = torch.unique(dbg["lik"])
                if lk.numel() > 1:
                    lk = torch.sort(lk, descending=True).values
                    m["path likelihood"] = float(lk[0] - lk[1]) / 1e-4
            t2 = torch.topk(vt[0], 2).values
            m["root argmax"] = float(t2[0] - t2[1]) / tol_logit
        m["min"] = min(m.values())
        return m

    # ---- tie-aware parity ----------------------------------------------------------------------
    @torch.no_grad()
    def decode_following(self, enc: torch.Tensor, gp, engine_ids: List[int], tol_logit: float = 5e-4, tol_rel_p: float = 2e-3,
                         max_iters: Optional[int] = None) -> Tuple[bool, List[dict], List[int]]:
        """Parity check that survives numerical ties.  Two correct implementations of this loop agree on every decision
        EXCEPT where the decision's own margin is below their numerical difference (fp32 summation order: ~1e-4 on logits of
        scale ~6): an argmax between two logits 1e-4 apart, or `p_c > thr` with both sides equal to 1e-3 relative.  The loop is
        chaotic after such a flip, so plain list equality reports a 'divergence' that is not an error.

        This walks the oracle along the ENGINE's token sequence: at every decision whose margin is below the tolerance
        (`tol_logit` absolute on processed logits, `tol_rel_p` relative on p_c vs thr) both outcomes are admissible and the one
        that reproduces the engine's next tokens is taken; every other decision must match exactly.  Returns (ok, ties, ids):
        `ties` lists the decisions that were resolved by following the engine (empty for a strictly identical run)."""
        import itertools
        cfg = self.cfg
        K, P, eos = cfg.medusa_num_heads, len(gp.prompt), gp.eos_token_id
        assert not gp.vanilla
        st = self.new_state(enc)
        ids = list(gp.prompt)
        ties: List[dict] = []
        n_it = 0
        self.last_follow_accepts: List[int] = []
        while True:
            L, kv = len(ids), st["kv_len"]
            z = self.decoder_pass(st, ids[kv:L], kv, disable_medusa=False, last_only=True)[:, 0]
            st["kv_len"] = L
            z = process_logits(z, L, gp)
            top = torch.topk(z, 2, dim=-1)
            opts = []
            for k in range(K + 1):
                m = float(top.values[k, 0] - top.values[k, 1])
                opts.append([int(top.indices[k, 0])] + ([int(top.indices[k, 1])] if m < tol_logit else []))
            n_combo = 1
            for o in opts:
                n_combo *= len(o)
            combos = itertools.product(*opts) if n_combo <= 64 else [tuple(o[0] for o in opts)]
            self_kv_snapshot = list(st["self_kv"])
            matched = None
            for cand_t in combos:
                st["self_kv"] = list(self_kv_snapshot); st["kv_len"] = L
                cand = torch.tensor(cand_t, dtype=torch.long)
                v = self.decoder_pass(st, list(cand_t), L, disable_medusa=True)[0]
                v = process_logits(v, L, gp)
                # admissible accept lengths
                if gp.accept_mode == 0 or gp.temperature == 0:
                    vt = torch.topk(v[:-1], 2, dim=-1)
                    ok = (cand[1:] == vt.indices[:, 0])
                    tie = ((vt.values[:, 0] - vt.values[:, 1]) < tol_logit) & ((cand[1:] == vt.indices[:, 0]) | (cand[1:] == vt.indices[:, 1]))
                else:
                    p = torch.softmax(v[:-1] / gp.temperature, dim=-1)
                    p_c = torch.gather(p, -1, cand[1:].unsqueeze(-1)).squeeze(-1)
                    H = -torch.sum(p * torch.log(p + 1e-5), dim=-1)
                    thr = torch.minimum(torch.full_like(H, gp.posterior_threshold), torch.exp(-H) * gp.posterior_alpha)
                    ok = p_c > thr
                    tie = (p_c - thr).abs() <= tol_rel_p * thr
                a_set, a = [], 0
                for i in range(K):
                    if bool(tie[i]):
                        a_set.append(a)                      # the tied candidate may be rejected here ...
                        a += 1                               # ... or accepted
                        continue
                    if not bool(ok[i]):
                        break
                    a += 1
                a_set.append(a)
                for a in sorted(set(a_set), reverse=True):
                    if a == 0:
                        v0 = torch.topk(v[0], 2)
                        seconds = [int(v0.indices[0])] + ([int(v0.indices[1])] if float(v0.values[0] - v0.values[1]) < tol_logit else [])
                        emits = [[cand_t[0], s2] for s2 in seconds]
                    else:
                        emits = [list(cand_t[: a + 1])]
                    for emit in emits:
                        tail = engine_ids[L: L + len(emit)]
                        if tail == emit[: len(tail)] and len(tail) > 0:
                            matched = (cand_t, a, emit)
                            break
                    if matched:
                        break
                if matched:
                    break
            if matched is None:
                st["self_kv"] = list(self_kv_snapshot)
                return False, ties + [dict(L=L, reason="no admissible continuation reproduces the engine", engine=engine_ids[L: L + K + 1],
                                           oracle_cand=[o[0] for o in opts])], ids
            cand_t, a, emit = matched
            strict_cand = tuple(o[0] for o in opts)
            # was anything other than the strict outcome taken?
            st_strict = None
            if cand_t != strict_cand:
                ties.append(dict(L=L, kind="candidate argmax", strict=list(strict_cand), taken=list(cand_t)))
            else:
                # strict accept length for this candidate set
                a_strict = 0
                for i in range(K):
                    if not bool(ok[i]):
                        break
                    a_strict += 1
                if a != a_strict:
                    ties.append(dict(L=L, kind="accept", strict=a_strict, taken=a))
                elif a == 0 and emit[1] != int(torch.argmax(v[0])):
                    ties.append(dict(L=L, kind="verify argmax", strict=int(torch.argmax(v[0])), taken=emit[1]))
            # state after the iteration (the verify pass of the matched candidate set is the one in st)
            st["kv_len"] = L + 1 if a == 0 else L + a
            ids += emit
            self.last_follow_accepts.append(a)
            n_it += 1
            L = len(ids)
            finished = (eos in emit) or (L >= gp.max_length) or (L + K >= gp.hard_max_length)
            if finished or (max_iters is not None and n_it >= max_iters):
                break
        res = self._finish(DecodeResult(ids=[], new_tokens=[]), ids, P, eos)
        return res.ids == engine_ids[: len(res.ids)] and (max_iters is not None or len(res.ids) == len(engine_ids)), ties, res.ids

    def _decode_vanilla(self, enc, gp, max_iters=None) -> DecodeResult:
        """Anchor: plain greedy decoding on head 0 / base logits, one token per pass."""
        P, eos = len(gp.prompt), gp.eos_token_id
        st = self.new_state(enc)
        ids = list(gp.prompt)
        res = DecodeResult(ids=[], new_tokens=[])
        while True:
            L, kv = len(ids), st["kv_len"]
            z = self.decoder_pass(st, ids[kv:L], kv, disable_medusa=True, last_only=True)[0]
            st["kv_len"] = L
            tok = int(torch.argmax(process_logits(z, L, gp)[0]))
            ids.append(tok)
            res.n_iters += 1
            if tok == eos or len(ids) >= gp.max_length or (max_iters is not None and res.n_iters >= max_iters):
                break
        return self._finish(res, ids, P, eos)

    @staticmethod
    def _finish(res, ids, P, eos):
        # post-EOS overwrite (model.py:798-810)
        if eos in ids:
            j = ids.index(eos)
            ids = ids[: j + 1] + [eos] * (len(ids) - j - 1)
        res.ids = ids
        gen = ids[P:]
        res.new_tokens = gen[: gen.index(eos)] if eos in gen else gen
        return res

    @torch.no_grad()
    def transcribe(self, wav: np.ndarray, gp, **kw) -> DecodeResult:
        """log-mel -> encoder -> Medusa decode for one clip (reference README.md:120-139)."""
        cfg = self.cfg
        feats = torch.from_numpy(log_mel(wav, cfg.num_mel_bins, cfg.n_mel_frames * 160))
        return self.decode(self.encode(feats), gp, **kw)
