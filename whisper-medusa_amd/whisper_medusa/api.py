"""Drop-in ``WhisperMedusaModel`` backed by the HIP engine.

Mirrors the reference's public API for the inference path (whisper_medusa/models/model.py):
``from_pretrained`` (:265-291), ``generate`` (:1419-1779, same keyword names, same error behaviour for
the unsupported features), ``forward`` (:1223-1347, logits ``[K+1, B, T, V]``), ``to``/``eval``.
The canonical caller is eval_whisper_medusa.py:28-69 / README.md:101-142:

    model = WhisperMedusaModel.from_pretrained(path); model = model.to("cuda")
    ids = model.generate(input_features, language="en")        # LongTensor [1, T]
    text = processor.decode(ids[0], skip_special_tokens=True)

Differences, all supersets: batch > 1 is accepted (semantics = B independent batch-1 runs, SURVEY.md §0
fact 2); ``extract_features(wav)`` computes the log-mel on the GPU (the reference leaves it to
``WhisperProcessor``); ``vanilla=True`` runs plain greedy decoding for the anchor measurement.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from .config import MedusaConfig, GenParams, ACCEPT_TYPICAL, ACCEPT_GREEDY
from .engine import Engine
from . import weights as _weights
from . import synth as _synth


class EngineKVCache:
    """``past_key_values`` of ``forward(use_cache=True)``: the self-attention K/V rows themselves stay in the engine's contiguous cache
    (DESIGN.md §3); this handle records which engine / encoder pass / batch they belong to and how many positions are filled, which
    is all a following ``forward(past_key_values=...)`` needs to append behind them (HF's Cache.get_seq_length())."""

    def __init__(self, engine_id: int, enc_stamp, kv_stamp, batch: int, seq_length: int):
        self.engine_id, self.enc_stamp, self.kv_stamp, self.batch, self._len = engine_id, enc_stamp, kv_stamp, batch, int(seq_length)

    def get_seq_length(self, layer_idx: int = 0) -> int:
        return self._len

    def __len__(self):
        return self._len


class EngineEncoderOutput:
    """``encoder_outputs`` as ``forward`` hands them back (HF BaseModelOutput shape: ``[0]`` / ``.last_hidden_state``).  The hidden
    state is fetched from the engine only when somebody looks at it; passed back into ``forward(encoder_outputs=...)`` while the
    engine still holds that encoder pass, it is recognised by its stamp and nothing is recomputed."""

    def __init__(self, engine, batch: int, stamp):
        self._engine, self._batch, self._wm_stamp, self._t = engine, batch, stamp, None

    @property
    def last_hidden_state(self) -> torch.Tensor:
        if self._t is None:
            if getattr(self._engine, "_enc_stamp", None) is not self._wm_stamp:
                raise RuntimeError("this encoder pass is no longer resident in the engine")
            self._t = self._engine.encoder_output_cached(self._batch)
        return self._t

    def __getitem__(self, i):
        if i != 0:
            raise IndexError(i)
        return self.last_hidden_state


@dataclass
class MedusaForwardOutput:
    """Seq2SeqLMOutput fields the engine can fill (model.py:1336-1347)."""
    logits: torch.Tensor                 # [K+1 (or 1), B, T, V]  (model.py:1301)
    past_key_values: Optional[EngineKVCache] = None
    encoder_outputs: Optional[EngineEncoderOutput] = None
    loss: Optional[torch.Tensor] = None

    @property
    def encoder_last_hidden_state(self):
        return None if self.encoder_outputs is None else self.encoder_outputs.last_hidden_state


class GenerateEncoderDecoderOutput(dict):
    """What ``generate(return_dict_in_generate=True)`` returns in the reference (model.py:812-823: HF's ModelOutput of the same
    name): fields by attribute, by key and by position.  The engine keeps scores / attentions / hidden states on the device and
    never materialises them, so — as in the reference when ``output_scores`` etc. are not requested — they are ``None``; the KV
    cache is engine state, not a Python object (``past_key_values`` is ``None``)."""
    _fields = ("sequences", "scores", "logits", "encoder_attentions", "encoder_hidden_states", "decoder_attentions",
               "cross_attentions", "decoder_hidden_states", "past_key_values")

    def __init__(self, sequences, **extra):
        super().__init__()                       # like ModelOutput: only the fields that are not None are keys
        self["sequences"] = sequences
        self.update({k: v for k, v in extra.items() if v is not None})

    def __getattr__(self, k):
        if k in self:
            return dict.__getitem__(self, k)
        if k in self._fields:
            return None
        raise AttributeError(k)

    def to_tuple(self):
        return tuple(dict.__getitem__(self, k) for k in self.keys())

    def __getitem__(self, k):
        if isinstance(k, int):
            return self.to_tuple()[k]
        return super().__getitem__(k)


# HF logits processors / stopping criteria that are STATIC functions of (token id, current length) are lowered into wm_gen_params
# (reference: generate() hands `logits_processor` / `stopping_criteria` to HF's _get_logits_processor / _get_stopping_criteria,
# model.py:1106-1124; the engine fuses exactly these into its select kernels).  Matched by class name: no transformers import.
_LOWERABLE_PROCESSORS = ("SuppressTokensLogitsProcessor", "SuppressTokensAtBeginLogitsProcessor", "ExponentialDecayLengthPenalty")
_LOWERABLE_CRITERIA = ("MaxLengthCriteria", "EosTokenCriteria")


def _as_int_list(x) -> List[int]:
    if x is None:
        return []
    if hasattr(x, "flatten"):
        return [int(v) for v in x.flatten().tolist()]
    if isinstance(x, (list, tuple, set)):
        return [int(v) for v in x]
    return [int(x)]


def lower_processors(gp: GenParams, logits_processor, stopping_criteria) -> GenParams:
    """Fold caller-supplied HF processors / criteria into the generation parameters.  A custom processor replaces the default of
    its type (HF's merge rule); a logits processor that is not a static mask / penalty is kept for the host (``gp._host_processors``:
    generate() then runs the loop with the logits copied out after every pass — the slow path, the reference's own); a stopping
    criterion that is not a static length / EOS rule is kept for the host (``gp._host_criteria``) and evaluated after every iteration."""
    P = len(gp.prompt)
    for proc in (logits_processor or []):
        name = type(proc).__name__
        if name == "SuppressTokensLogitsProcessor":
            gp.suppress_tokens = _as_int_list(proc.suppress_tokens)
        elif name == "SuppressTokensAtBeginLogitsProcessor":
            # only the token list is the caller's: the reference overwrites begin_index on every processor that has set_begin_index
            # with the number of init tokens (model.py:1537, :1640-1644) — _gen_params has already put that number in gp
            gp.begin_suppress_tokens = _as_int_list(proc.begin_suppress_tokens)
        elif name == "ExponentialDecayLengthPenalty":
            if _as_int_list(proc.eos_token_id) != [gp.eos_token_id]:
                raise NotImplementedError("ExponentialDecayLengthPenalty on a token other than eos_token_id is not supported by the HIP engine")
            start = int(proc.regulation_start) - P                   # regulation_start is absolute in HF, relative to the prompt in wm_gen_params
            if start < 0:
                # HF would already be penalising at the first generated token with exponent P - regulation_start; the engine's
                # field cannot say that (negative = off), and silently dropping the penalty would change the tokens
                raise NotImplementedError(f"ExponentialDecayLengthPenalty.regulation_start ({int(proc.regulation_start)}) lies inside the "
                                          f"decoder prompt ({P} tokens): build it with input_ids_seq_length = the prompt length")
            gp.exp_decay = (start, float(proc.regulation_factor))
        else:
            # any other processor is a Python callable on the logits (the reference hands the list to HF, model.py:1106-1116): it runs on
            # the host between the engine's passes (WhisperMedusaModel._decode_host_processors: the reference's own loop structure,
            # one stream at a time, logits copied out after every pass).  Order as in HF: the defaults above first, custom ones after.
            host = getattr(gp, "_host_processors", None) or []
            host.append(proc)
            gp._host_processors = host
    for crit in (stopping_criteria or []):
        name = type(crit).__name__
        if name == "MaxLengthCriteria":
            gp.max_length = min(gp.max_length, int(crit.max_length))
        elif name == "EosTokenCriteria":
            if _as_int_list(crit.eos_token_id) != [gp.eos_token_id]:
                raise NotImplementedError("EosTokenCriteria with other ids than eos_token_id is not supported by the HIP engine")
        else:
            # anything else is evaluated on the host after every iteration (generate(): the per-iteration path the streamer uses),
            # exactly where the reference calls it (model.py:786: once per Medusa iteration, on the ids emitted so far)
            host = getattr(gp, "_host_criteria", None) or []
            host.append(crit)
            gp._host_criteria = host
    return gp


def _resolve_checkpoint(name_or_path, revision=None, cache_dir=None, local_files_only=False) -> str:
    """A checkpoint directory as it is; anything else is taken as a hub repo id and resolved with huggingface_hub.snapshot_download
    (cached snapshot first; config, weights and tokenizer files only)."""
    import os
    path = os.fspath(name_or_path)
    if os.path.isdir(path):
        return path
    try:
        from huggingface_hub import snapshot_download
    except ImportError as e:                                    # pragma: no cover - huggingface_hub ships with transformers
        raise OSError(f"{path!r} is not a checkpoint directory and huggingface_hub is not installed to resolve it as a hub name") from e
    patterns = ["*.json", "*.safetensors", "pytorch_model.bin", "*.txt", "*.model"]
    try:
        try:
            return snapshot_download(path, revision=revision, cache_dir=cache_dir, allow_patterns=patterns, local_files_only=True)
        except Exception:
            if local_files_only:
                raise
            return snapshot_download(path, revision=revision, cache_dir=cache_dir, allow_patterns=patterns)
    except Exception as e:
        raise OSError(f"{path!r} is neither a checkpoint directory nor a hub repository that could be resolved "
                      f"({type(e).__name__}: {e})") from e


class WhisperMedusaModel:
    def __init__(self, config: MedusaConfig, state_dict: Dict[str, torch.Tensor], device: Union[str, torch.device, None] = None,
                 max_batch: int = 1, dec_weight_fp8: bool = False, enc_fp8: bool = False, act_fp16: Optional[bool] = None,
                 cross_kv_fp8: bool = False):
        self.config = config
        self.generation_config = config          # posterior_threshold / alpha / token ids live on the config here
        self._sd = state_dict
        self._max_batch = max_batch
        self._fp8 = bool(dec_weight_fp8)         # decoder-layer matrices stored as fp8 e4m3 + per-row scale (BASELINE configs[4])
        self._enc_fp8 = bool(enc_fp8)            # encoder QKV / FC1 / cross-K/V projection on the fp8 MFMA (BASELINE configs[4])
        # decode numerics contract (include/wm.h wm_config.act_fp16): False = bf16 hi / lo operand pairs (libwm.so), True = one fp16
        # plane (libwm_f16.so, decoder matrices held as fp16); None = the default (engine.default_act_fp16: WM_ACT)
        from .engine import default_act_fp16
        self._act_fp16 = default_act_fp16() if act_fp16 is None else bool(act_fp16)
        self._xkv_fp8 = bool(cross_kv_fp8)       # the decode loop streams an e4m3 copy of the encoder cross-K/V (BASELINE configs[4]; wm_config.cross_kv_fp8)
        self._micro_batches = 1                  # contexts a batch is decoded with; None = automatic (set_micro_batches(None))
        self._pool = None
        self._engine: Optional[Engine] = None
        self._blob = None
        self.device = torch.device("cpu")
        if device is not None:
            self.to(device)

    # ---- construction ---------------------------------------------------------------------
    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, *args, device=None, max_batch: int = 1, dec_weight_fp8: bool = False,
                        enc_fp8: bool = False, act_fp16: Optional[bool] = None, cross_kv_fp8: bool = False, **kwargs):
        """Load ``config.json`` + ``model.safetensors`` from a checkpoint directory, or from a Hugging Face hub name
        (``aiola/whisper-medusa-linear-libri``, README.md:101-104 of the reference) resolved through ``huggingface_hub`` — its
        local cache first, a download when the machine has network access (reference model.py:265-291)."""
        pretrained_model_name_or_path = _resolve_checkpoint(pretrained_model_name_or_path, kwargs.get("revision"),
                                                            kwargs.get("cache_dir"), kwargs.get("local_files_only", False))
        config = MedusaConfig.from_pretrained(pretrained_model_name_or_path)
        sd = _weights.load_state_dict_from_dir(pretrained_model_name_or_path)
        return cls(config, sd, device=device, max_batch=max_batch, dec_weight_fp8=dec_weight_fp8, enc_fp8=enc_fp8, act_fp16=act_fp16, cross_kv_fp8=cross_kv_fp8)

    def save_pretrained(self, save_directory: str, safe_serialization: bool = True) -> None:
        """``config.json`` + ``model.safetensors`` with the reference's parameter names (what its Trainer writes,
        trainer.py:45-51; the tied ``proj_out.weight`` is dropped like HF does) — re-loadable by ``from_pretrained`` here
        and by the reference.  Needs the state dict (models built with ``from_blob`` only hold the packed blob)."""
        import os
        if not self._sd:
            raise RuntimeError("save_pretrained needs the parameter state dict; this model was built from a packed blob")
        os.makedirs(save_directory, exist_ok=True)
        self.config.save_pretrained(save_directory)
        sd = {k: v.detach().cpu().contiguous().clone() for k, v in self._sd.items() if k != "whisper_model.proj_out.weight"}
        if safe_serialization:
            from safetensors.torch import save_file
            save_file(sd, os.path.join(save_directory, "model.safetensors"), metadata={"format": "pt"})
        else:
            torch.save(sd, os.path.join(save_directory, "pytorch_model.bin"))

    # ---- packed export (checkpoint tooling; the fp8 export format) -----------------------------------------
    PACKED_FORMAT = 1

    def save_packed(self, save_directory: str) -> None:
        """Write the engine-ready parameter blob: ``wm_packed.bin`` (the bytes ``wm_create`` consumes: bf16 MFMA-fragment
        matrices, fp32 vectors and — for models built with ``dec_weight_fp8`` / ``enc_fp8`` — the e4m3 matrices with their
        per-row scales, quantised once, on the CPU) and ``wm_packed.json`` (format + ABI layout version, the fp8 flags, the
        offset table, sha256 of the blob) next to ``config.json``.  ``from_packed`` loads it without touching the fp32 checkpoint."""
        import hashlib, json, os
        from .engine import WM_ABI_VERSION
        if self._blob is None:
            if not self._sd:
                raise RuntimeError("nothing to export: the model holds neither a state dict nor a packed blob")
            self._blob, self._offsets = _weights.build_blob(self.config, self._sd, device="cpu", dec_fp8=self._fp8, enc_fp8=self._enc_fp8, act_fp16=self._act_fp16)
        os.makedirs(save_directory, exist_ok=True)
        self.config.save_pretrained(save_directory)
        raw = self._blob.detach().cpu().contiguous().numpy().tobytes()
        with open(os.path.join(save_directory, "wm_packed.bin"), "wb") as f:
            f.write(raw)
        meta = dict(packed_format=self.PACKED_FORMAT, abi_layout=WM_ABI_VERSION, dec_weight_fp8=self._fp8, enc_fp8=self._enc_fp8, act_fp16=self._act_fp16, cross_kv_fp8=self._xkv_fp8,
                    n_bytes=len(raw), sha256=hashlib.sha256(raw).hexdigest(), offsets=[int(o) for o in self._offsets])
        with open(os.path.join(save_directory, "wm_packed.json"), "w") as f:
            json.dump(meta, f)

    @classmethod
    def from_packed(cls, directory: str, device=None, max_batch: int = 1):
        """Load a ``save_packed`` export.  Refuses a blob written for another parameter-table layout or with a wrong checksum."""
        import hashlib, json, os
        from .engine import WM_ABI_VERSION
        with open(os.path.join(directory, "wm_packed.json")) as f:
            meta = json.load(f)
        if meta.get("packed_format") != cls.PACKED_FORMAT or meta.get("abi_layout") != WM_ABI_VERSION:
            raise ValueError(f"packed export has format {meta.get('packed_format')} / table layout {meta.get('abi_layout')}; this build "
                             f"reads format {cls.PACKED_FORMAT} / layout {WM_ABI_VERSION}: re-export from the checkpoint")
        raw = np.fromfile(os.path.join(directory, "wm_packed.bin"), dtype=np.uint8)
        if raw.size != meta["n_bytes"] or hashlib.sha256(raw.tobytes()).hexdigest() != meta["sha256"]:
            raise ValueError("packed export is truncated or corrupt (size / sha256 mismatch)")
        config = MedusaConfig.from_pretrained(directory)
        if len(meta["offsets"]) != _weights.n_table_entries(config, meta["dec_weight_fp8"], meta["enc_fp8"]):
            raise ValueError("packed export's offset table does not match its config")
        self = cls(config, {}, device=None, max_batch=max_batch, dec_weight_fp8=meta["dec_weight_fp8"], enc_fp8=meta["enc_fp8"],
                   act_fp16=bool(meta.get("act_fp16", False)), cross_kv_fp8=bool(meta.get("cross_kv_fp8", False)))
        self._blob, self._offsets = torch.from_numpy(raw), np.asarray(meta["offsets"], dtype=np.uint64)
        return self.to(device) if device is not None else self

    @classmethod
    def from_blob(cls, config: MedusaConfig, blob: torch.Tensor, offsets: np.ndarray, max_batch: int = 1, dec_weight_fp8: bool = False,
                  enc_fp8: bool = False, act_fp16: bool = False, cross_kv_fp8: bool = False):
        """Build directly from a packed parameter blob already resident on a GPU (the path the
        8-GPU data-parallel launcher uses after the RCCL broadcast, ``dist.py``)."""
        self = cls(config, {}, device=None, max_batch=max_batch, dec_weight_fp8=dec_weight_fp8, enc_fp8=enc_fp8, act_fp16=act_fp16, cross_kv_fp8=cross_kv_fp8)
        self._blob, self._offsets = blob, offsets
        self.device = blob.device
        self._engine = Engine(config, blob, offsets, max_batch=max_batch, device=blob.device, dec_weight_fp8=dec_weight_fp8, enc_fp8=enc_fp8, act_fp16=act_fp16, cross_kv_fp8=cross_kv_fp8)
        return self

    def to(self, device):
        device = torch.device(device)
        if device.type != "cuda":
            if self._engine is None:
                self.device = device
                return self
            raise RuntimeError("the Whisper-Medusa engine runs on a HIP device only")
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        if self._engine is not None and self.device == device:
            return self
        self._drop_pool()
        if self._engine is not None:
            self._engine.close()                                   # frees its KV caches / scratch before the new context allocates
            self._engine = None
        with torch.cuda.device(device):
            if self._sd:
                self._blob, self._offsets = _weights.build_blob(self.config, self._sd, device=device, dec_fp8=self._fp8, enc_fp8=self._enc_fp8, act_fp16=self._act_fp16)
            elif self._blob is not None:
                self._blob = self._blob.to(device)                 # built with from_blob: move the packed blob itself
            else:
                raise RuntimeError("model has neither a state dict nor a packed blob to place on the device")
            self._engine = Engine(self.config, self._blob, self._offsets, max_batch=self._max_batch, device=device, dec_weight_fp8=self._fp8, enc_fp8=self._enc_fp8, act_fp16=self._act_fp16, cross_kv_fp8=self._xkv_fp8)
        self._drop_pool()
        self.device = device
        return self

    def cuda(self, device=None):
        return self.to("cuda" if device is None else device)

    def eval(self):
        return self

    def set_max_batch(self, max_batch: int):
        if max_batch != self._max_batch:
            self._max_batch = max_batch
            if self._engine is not None:
                self._engine.close()
                self._engine = Engine(self.config, self._blob, self._offsets, max_batch=max_batch, device=self.device, dec_weight_fp8=self._fp8, enc_fp8=self._enc_fp8, act_fp16=self._act_fp16, cross_kv_fp8=self._xkv_fp8)
            self._drop_pool()
        return self

    def set_micro_batches(self, n: Optional[int]):
        """Decode batches as ``n`` concurrent micro-batches (own engine context + HIP stream each, shared weights):
        see ``pool.py``.  n = 1 (default) is the single-context path — the model's own engine then holds the state of the
        last call (encoder output, statistics); ``None`` picks per call (``_micro_batches_for``).  Tokens do not depend on n."""
        if n is not None and n < 1:
            raise ValueError("micro_batches must be >= 1")
        if n != self._micro_batches:
            self._micro_batches = n
            self._drop_pool()
        return self

    def _micro_batches_for(self, B: int) -> int:
        """Contexts a batch of B clips is decoded with.  Automatic policy, measured on MI355X with large-v2 (tests/microbench/
        r02_call25.sh): two or three streams decode faster as single-stream contexts running side by side (2: 2135 vs 1455
        tokens/s, 3: 2335 vs 2187 — the single-stream kernels are the fused 16-row ones and two chains of short launches fill
        each other's gaps), from four streams on one batched context wins (4: 2822 vs 2517 / 2477; 8: 4177 vs 3834 / 2326)."""
        if self._micro_batches is not None:
            return self._micro_batches
        return B if B in (2, 3) else 1

    def _drop_pool(self):
        if self._pool is not None:
            self._pool.close()
            self._pool = None

    def _get_pool(self, n: int):
        """Pool of n contexts.  Under the automatic policy (two or three clips -> as many single-stream contexts) ONE pool serves both
        sizes: it is built with as many contexts as the first batch needs — two clips do not pay for a third context's KV caches and
        scratch — and GROWS to three when a batch of three arrives (the existing contexts, their caches and captured graphs stay)."""
        auto = self._micro_batches is None
        have = getattr(self, "_pool_n", None)
        if self._pool is not None and not auto and have != n:
            self._drop_pool()
        if self._pool is None:
            from .pool import ContextPool
            _ = self.engine                                         # raises when not on a HIP device
            # automatic policy: contexts of ONE stream each (per_ctx = ceil(max_batch / contexts) must not depend on the first batch's size)
            self._pool = ContextPool(self.config, self._blob, self._offsets, n, n if auto else max(self._max_batch, n), self._fp8, self._enc_fp8, self._act_fp16, self._xkv_fp8)
            self._pool_n = n
        elif auto and have < n:
            self._pool.grow(n)
            self._pool_n = n
        return self._pool

    @property
    def engine(self) -> Engine:
        if self._engine is None:
            raise RuntimeError("model is not on a HIP device: call model.to('cuda') first (there is no CPU path)")
        return self._engine

    def get_medusa_choice(self):
        return self.config.medusa_choices

    # ---- F0 ---------------------------------------------------------------------------------
    def extract_features(self, wav: Union[np.ndarray, torch.Tensor, Sequence[np.ndarray]], sampling_rate: int = 16000) -> torch.Tensor:
        """waveform(s) -> log-mel ``input_features`` [B, 80, 3000] on the GPU; pads / trims to 30 s like
        ``WhisperFeatureExtractor`` (eval_whisper_medusa.py:46-50).  Clips in a list are mono [n] or multi-channel
        [channels, n] (a bare 2-D array is a batch of mono clips); anything that is not 16 kHz mono first goes through the audio front door on the GPU
        (channel mean + torchaudio-default resampling, README.md:120-125 of the reference)."""
        n = 160 * self.config.n_mel_frames
        if isinstance(wav, (list, tuple)):
            clips = [np.asarray(w, dtype=np.float32) for w in wav]
        else:
            a = wav.detach().cpu().numpy() if isinstance(wav, torch.Tensor) else np.asarray(wav)
            a = a.astype(np.float32)
            clips = [a] if a.ndim == 1 else [r for r in a]          # 2-D array = batch of mono clips; multi-channel clips go in a list
        if len(clips) > self._max_batch:
            self.set_max_batch(len(clips))                          # wm_logmel checks B <= max_batch
        buf = torch.zeros(len(clips), n, dtype=torch.float32, device=self.device)
        for i, c in enumerate(clips):
            if c.ndim == 2 or sampling_rate != 16000:
                t = torch.from_numpy(np.atleast_2d(c)[None]).to(self.device)         # [1, channels, n_in]
                r = self.engine.resample(t, sampling_rate, 16000)[0]
            else:
                r = torch.from_numpy(c).to(self.device)
            m = min(r.numel(), n)
            buf[i, :m] = r[:m]
        return self.engine.logmel(buf)

    def features_from_file(self, path: str) -> torch.Tensor:
        """PCM WAV file -> ``input_features`` [1, 80, 3000] (decode, downmix, resample, log-mel)."""
        from .audio import read_wav
        x, sr = read_wav(path)
        return self.extract_features([x], sampling_rate=sr)

    # ---- generate ---------------------------------------------------------------------------
    def _gen_params(self, language, task, exponential_decay_length_penalty, max_new_tokens, max_length,
                    temperature, vanilla, posterior_threshold, posterior_alpha, suppress_tokens,
                    begin_suppress_tokens, prompt_ids) -> GenParams:
        cfg = self.config
        prompt = _synth.default_prompt(cfg, language or "en", task or "transcribe")        # G1, model.py:1519-1537
        begin_suppress_index = None
        if prompt_ids is not None:
            # short-form conditioning (HF _prepare_decoder_input_ids): decoder_input_ids = cat([prompt_ids, init_tokens]);
            # `prompt_ids` comes from processor.get_prompt_ids() and already starts with <|startofprev|>
            pid = [int(t) for t in (prompt_ids.flatten().tolist() if hasattr(prompt_ids, "flatten") else list(prompt_ids))]
            if len(pid) + len(prompt) + cfg.medusa_num_heads + 2 > cfg.max_target_positions:
                raise ValueError(f"prompt_ids of length {len(pid)} leave no room to generate "                      # model.py:1526-1529
                                 f"(max_target_positions {cfg.max_target_positions})")
            begin_suppress_index = len(prompt)      # reference model.py:1537: begin_index = init_tokens.shape[1] (without prompt_ids)
            prompt = pid + prompt
        P = len(prompt)
        if max_new_tokens is not None:
            mlen = P + int(max_new_tokens)                                                # G2, model.py:1635-1639
        elif max_length is not None:
            mlen = int(max_length)
        else:
            mlen = cfg.max_length
        mlen = min(mlen, cfg.max_target_positions)
        if temperature is not None and float(temperature) > 0.0:
            # reference: do_sample path falls through with `result` unbound (model.py:1128-1156)
            raise NotImplementedError("sampling (temperature > 0) is not supported with medusa")
        # G4: generate() forces generation_config.temperature = 1.0 when not sampling (model.py:1877-1881),
        # so the typical-acceptance branch of evaluate_posterior runs.  temperature=0.0 selects exact-match.
        mode = ACCEPT_GREEDY if (temperature is not None and float(temperature) == 0.0) else ACCEPT_TYPICAL
        sup = list(cfg.suppress_tokens or []) if suppress_tokens is None else list(suppress_tokens)
        bsup = list(cfg.begin_suppress_tokens or []) if begin_suppress_tokens is None else list(begin_suppress_tokens)
        return GenParams(prompt=prompt, eos_token_id=cfg.eos_token_id, pad_token_id=cfg.pad_token_id,
                         suppress_tokens=sup, begin_suppress_tokens=bsup, max_length=mlen,
                         hard_max_length=cfg.max_length,
                         exp_decay=tuple(exponential_decay_length_penalty) if exponential_decay_length_penalty else None,
                         posterior_threshold=cfg.posterior_threshold if posterior_threshold is None else posterior_threshold,
                         posterior_alpha=cfg.posterior_alpha if posterior_alpha is None else posterior_alpha,
                         accept_mode=mode, temperature=1.0 if mode == ACCEPT_TYPICAL else 0.0, vanilla=bool(vanilla),
                         begin_suppress_index=begin_suppress_index)

    @torch.no_grad()
    def generate(self, input_features: Optional[torch.Tensor] = None, generation_config=None, logits_processor=None,
                 stopping_criteria=None, prefix_allowed_tokens_fn=None, synced_gpus: bool = False,
                 return_timestamps: Optional[bool] = None, task: Optional[str] = None, language: Optional[str] = None,
                 is_multilingual: Optional[bool] = None, prompt_ids: Optional[torch.Tensor] = None,
                 prompt_condition_type: Optional[str] = None, condition_on_prev_tokens: Optional[bool] = None,
                 temperature: Optional[Union[float, Tuple[float, ...]]] = None,
                 compression_ratio_threshold: Optional[float] = None, logprob_threshold: Optional[float] = None,
                 no_speech_threshold: Optional[float] = None, num_segment_frames: Optional[int] = None,
                 attention_mask: Optional[torch.Tensor] = None, time_precision: float = 0.02,
                 time_precision_features: float = 0.01, return_token_timestamps: Optional[bool] = None,
                 return_segments: bool = False, return_dict_in_generate: Optional[bool] = None,
                 force_unique_generate_call: Optional[bool] = None, **kwargs):
        """Same signature as the reference (model.py:1419-1449).  Returns ``LongTensor [B, T]`` holding the
        prompt + generated ids, right-padded with ``pad_token_id`` (model.py:1747-1762)."""
        if generation_config is not None:
            # HF semantics (model.py:936-943 -> GenerationMixin._prepare_generation_config): a copy of the passed config, updated by every
            # explicit argument of this call — an explicit argument wins, the config fills what the call leaves open.  Only the fields this
            # path reads are consulted; sampling has no Medusa path in the reference either (model.py:1128-1156).
            gc = generation_config

            def pick(name, cur):
                return cur if cur is not None else getattr(gc, name, None)
            return_timestamps = pick("return_timestamps", return_timestamps)
            task, language = pick("task", task), pick("language", language)
            no_speech_threshold = pick("no_speech_threshold", no_speech_threshold)
            return_token_timestamps = pick("return_token_timestamps", return_token_timestamps)
            return_dict_in_generate = pick("return_dict_in_generate", return_dict_in_generate)
            for name in ("max_new_tokens", "max_length", "num_beams", "suppress_tokens", "begin_suppress_tokens",
                         "exponential_decay_length_penalty", "posterior_threshold", "posterior_alpha"):
                if kwargs.get(name) is None and getattr(gc, name, None) is not None:
                    kwargs[name] = getattr(gc, name)
            if kwargs.get("do_sample") is None and getattr(gc, "do_sample", False):
                kwargs["do_sample"] = True
            # fields HF's _prepare_generation_config would apply and this path does not read: say so instead of ignoring them silently
            # (a value equal to the model's own generation config, or to HF's default, is not a request)
            own = getattr(self, "generation_config", None)
            ignored = []
            for name, dflt in (("temperature", 1.0), ("eos_token_id", None), ("pad_token_id", None), ("forced_decoder_ids", None),
                               ("top_k", 50), ("top_p", 1.0), ("repetition_penalty", 1.0), ("no_repeat_ngram_size", 0),
                               ("length_penalty", 1.0), ("bad_words_ids", None), ("min_length", 0), ("min_new_tokens", None)):
                v = getattr(gc, name, None)
                if v is None or v == dflt or (own is not None and v == getattr(own, name, None)):
                    continue
                ignored.append(name)
            if ignored:
                import warnings
                warnings.warn("generate(generation_config=...): the Medusa path does not honour " + ", ".join(ignored) +
                              " from a passed config (see INTEGRATION.md, 'generation_config fields'); pass processors / criteria explicitly",
                              UserWarning, stacklevel=2)
        if kwargs.get("do_sample"):
            raise NotImplementedError("sampling (do_sample=True) is not supported with medusa")      # model.py:1128-1156: no Medusa branch
        if return_timestamps:
            raise NotImplementedError("return_timestamps is not supported with medusa for now")   # model.py:1171-1175
        if no_speech_threshold is not None:
            raise NotImplementedError("no_speech_detection is not supported with medusa for now")  # model.py:1201-1205
        if kwargs.get("num_beams", 1) not in (None, 1):
            raise Exception("Beam search is not supported with medusa for now")                     # model.py:1153-1156
        if prefix_allowed_tokens_fn:
            raise NotImplementedError("prefix_allowed_tokens_fn is not supported by the HIP engine")
        if return_token_timestamps:
            raise NotImplementedError("token timestamps are not supported with medusa")
        if input_features is None:
            raise ValueError("input_features is required")
        if input_features.dim() != 3:
            raise ValueError("input_features must be [B, n_mels, frames]")
        if input_features.shape[-1] > self.config.n_mel_frames:
            if not kwargs.get("chunk_longform"):
                raise NotImplementedError("Longform generation is not supported yet")              # model.py:1213-1214
            return self._generate_longform(input_features, dict(kwargs, language=language, task=task, temperature=temperature,
                                                                prompt_ids=prompt_ids, logits_processor=logits_processor,
                                                                stopping_criteria=stopping_criteria))
        if language is None and self.config.is_multilingual and kwargs.get("detect_language", True) and input_features.shape[0] >= 1 \
                and not kwargs.get("_language_resolved"):
            return self._generate_detecting_language(input_features, dict(kwargs, task=task, temperature=temperature, prompt_ids=prompt_ids,
                                                                          return_dict_in_generate=return_dict_in_generate,
                                                                          return_segments=return_segments, logits_processor=logits_processor,
                                                                          stopping_criteria=stopping_criteria))
        B = input_features.shape[0]
        if B > self._max_batch:
            self.set_max_batch(B)
        gp = self._gen_params(language, task, kwargs.get("exponential_decay_length_penalty"),
                              kwargs.get("max_new_tokens"), kwargs.get("max_length"),
                              temperature if not isinstance(temperature, (tuple, list)) else temperature[0],
                              kwargs.get("vanilla", False), kwargs.get("posterior_threshold"),
                              kwargs.get("posterior_alpha"), kwargs.get("suppress_tokens"),
                              kwargs.get("begin_suppress_tokens"), prompt_ids)
        if logits_processor or stopping_criteria:
            gp = lower_processors(gp, logits_processor, stopping_criteria)
        self._last_prompt = list(gp.prompt)
        feats = input_features.to(self.device, torch.float32).contiguous()
        n_ctx = self._micro_batches_for(B) if (kwargs.get("streamer") is None and not getattr(gp, "_host_criteria", None)
                                               and not getattr(gp, "_host_processors", None)) else 1
        # language detection has just encoded exactly this batch on the model's own engine: decoding there saves the pool's second
        # encoder pass over the same clips (the automatic policy's gain at 2-3 clips is smaller than an encoder pass)
        if self._micro_batches is None and kwargs.get("_encoded_batch") == B and getattr(self._engine, "_B", None) == B:
            n_ctx = 1
        if n_ctx > 1 and B >= 2:
            pool = self._get_pool(n_ctx)
            seqs = pool.run(feats, gp)                                      # F1..F14 per micro-batch, concurrently
            self.last_stats = pool.last_stats
            return self._outputs(seqs, gp, return_dict_in_generate, return_segments)
        eng = self.engine
        streamer = kwargs.get("streamer")
        host_crit = getattr(gp, "_host_criteria", None)
        if streamer is not None and B != 1:
            raise ValueError("streamer only supports batch size 1")        # HF streamers are batch-1
        if getattr(gp, "_host_processors", None):
            seqs = self._decode_host_processors(feats, gp, streamer, host_crit)
            return self._outputs(seqs, gp, return_dict_in_generate, return_segments)
        # language detection just encoded exactly these clips on this engine (one language group, same order): its encoder output and
        # cross-K/V are still resident — decode from them instead of running the encoder a second time
        if not (kwargs.get("_encoded_batch") == B and getattr(eng, "_B", None) == B):
            eng.encode(feats)                                               # F1 + F2
        if streamer is not None or host_crit:
            # one iteration per engine call: the streamer gets the tokens of every iteration (model.py:1034-1035 prompt, :758-759
            # tokens, :795-796 end); stopping criteria that are not static length / EOS rules are asked after every iteration with
            # the ids emitted so far (model.py:786) — a stream they stop keeps its ids up to that iteration
            cur = [list(gp.prompt) for _ in range(B)]
            stop_len: List[Optional[int]] = [None] * B
            if streamer is not None:
                streamer.put(torch.tensor([gp.prompt], dtype=torch.long))

            K = self.config.medusa_num_heads

            def over(b):
                # stopped by a criterion, or finished by the engine's own rules (EOS emitted / length limits, model.py:774-793).  NOT "the
                # stream emitted nothing": with several streams a step may be a stream's base pass (merged-step schedule), which emits nothing
                if stop_len[b] is not None:
                    return True
                gen = cur[b][len(gp.prompt):]
                return gp.eos_token_id in gen or len(cur[b]) >= gp.max_length or len(cur[b]) + K >= gp.hard_max_length

            def on_it(new):
                for b in range(B):
                    if stop_len[b] is not None or not new[b]:
                        continue
                    cur[b].extend(new[b])
                    if streamer is not None:
                        streamer.put(torch.tensor(new[b], dtype=torch.long))
                    ids_b = torch.tensor([cur[b]], dtype=torch.long)
                    if host_crit and any(bool(torch.as_tensor(c(ids_b, None)).all()) for c in host_crit):
                        stop_len[b] = len(cur[b])
                return bool(host_crit) and all(over(b) for b in range(B))     # True ends the run (every stream stopped or finished)

            seqs = eng.decode(gp, B, on_iteration=on_it)
            seqs = [s_[: stop_len[b]] if stop_len[b] is not None else s_ for b, s_ in enumerate(seqs)]
            if streamer is not None:
                streamer.end()
        else:
            seqs = eng.decode(gp, B)                                        # F3..F14
        self.last_stats = eng.stats()
        return self._outputs(seqs, gp, return_dict_in_generate, return_segments)

    # ---- arbitrary logits processors: the reference's loop with the passes on the engine --------------------------------------------
    @staticmethod
    def _static_processors(scores: torch.Tensor, cur_len: int, gp: GenParams) -> torch.Tensor:
        """The processors the engine fuses into its select kernels, in torch, for the host path: HF's ExponentialDecayLengthPenalty,
        SuppressTokensAtBeginLogitsProcessor, SuppressTokensLogitsProcessor on rows ``scores [R, V]``; every row sees the same
        ``cur_len = input_ids.shape[-1]`` (model.py:653-665, :689-694)."""
        s_ = scores.clone()
        if gp.exp_decay is not None:
            start = int(gp.exp_decay[0]) + len(gp.prompt)
            if cur_len > start:
                e = gp.eos_token_id
                s_[:, e] = s_[:, e] + s_[:, e].abs() * (pow(float(gp.exp_decay[1]), cur_len - start) - 1.0)
        if gp.begin_suppress_tokens and cur_len == gp.begin_index:
            s_[:, list(gp.begin_suppress_tokens)] = -float("inf")
        if gp.suppress_tokens:
            s_[:, list(gp.suppress_tokens)] = -float("inf")
        return s_

    @staticmethod
    def _accept_length(v: torch.Tensor, cand: torch.Tensor, gp: GenParams) -> int:
        """evaluate_posterior for one candidate chain (medusa_utils.py:526-588): exact match at temperature 0 (:547-560), typical
        acceptance otherwise (:562-577: p_c > min(threshold, alpha exp(-H)), H = -sum p log(p + 1e-5)); the number of leading accepts."""
        if gp.accept_mode == ACCEPT_GREEDY or not gp.temperature:
            ok = cand[1:] == torch.argmax(v[:-1], dim=-1)
        else:
            p = torch.softmax(v[:-1] / gp.temperature, dim=-1)
            p_c = torch.gather(p, -1, cand[1:].unsqueeze(-1)).squeeze(-1)
            H = -torch.sum(p * torch.log(p + 1e-5), dim=-1)
            ok = p_c > torch.minimum(torch.full_like(H, gp.posterior_threshold), torch.exp(-H) * gp.posterior_alpha)
        return int(torch.cumprod(ok.int(), dim=0).sum().item())

    def _decode_host_processors(self, feats: torch.Tensor, gp: GenParams, streamer=None, host_crit=None) -> List[List[int]]:
        """generate() with logits processors that are arbitrary Python callables (the reference hands ANY list to HF's machinery,
        model.py:1106-1116, and calls it on the base / Medusa logits and on the verify logits, :653-665, :689-694).  The engine's fused
        loop never lets logits leave the device, so this path runs the reference's loop structure (model.py:634-793) on the host, one
        stream at a time (the reference asserts batch 1), with every decoder pass on the engine (`wm_forward_logits`: K/V rows appended
        to the contiguous cache at the pass's position; rejected rows are overwritten by the next pass) and everything that touches
        logits — the static processors, the caller's processors, arg-max candidates, the posterior — in torch.  Same tokens as the fused
        loop wherever the caller's processors are pure functions of (ids, logits); orders of magnitude slower: it exists for API
        completeness, not for throughput.  Candidate chains only."""
        cfg, eng = self.config, self.engine
        if cfg.is_tree:
            raise NotImplementedError("custom logits processors with a candidate tree (medusa_choices with top-k > 1) are not supported")
        if gp.vanilla:
            raise NotImplementedError("custom logits processors with vanilla=True are not supported")
        procs = list(gp._host_processors)
        K, P, eos = cfg.medusa_num_heads, len(gp.prompt), gp.eos_token_id

        def run(ids_t: torch.Tensor, scores: torch.Tensor) -> torch.Tensor:
            out = self._static_processors(scores, ids_t.shape[-1], gp)
            for pr in procs:
                out = pr(ids_t, out)
            return out

        def engine_pass(tokens: List[int], pos0: int, disable_medusa: bool) -> torch.Tensor:       # [n_out, T, V]
            if len(tokens) <= 16:
                return eng.forward_logits([tokens], pos0, disable_medusa)[:, 0]
            return torch.cat([eng.forward_logits([tokens[c: c + 16]], pos0 + c, disable_medusa)[:, 0] for c in range(0, len(tokens), 16)], dim=1)

        seqs: List[List[int]] = []
        n_iter = n_tok = 0
        hist = [0] * (K + 1)
        for b in range(feats.shape[0]):
            eng.encode(feats[b: b + 1].contiguous())
            ids, kv = list(gp.prompt), 0
            if streamer is not None:
                streamer.put(torch.tensor([gp.prompt], dtype=torch.long))
            while True:
                L = len(ids)
                ids_t = torch.tensor([ids], dtype=torch.long)
                z = engine_pass(ids[kv:L], kv, False)[:, -1]                     # base + Medusa logits of the last position  [K+1, V]
                # two calls like the reference (model.py:653-655 base rows, :656-665 Medusa rows): a processor that indexes rows by batch
                # (HF RepetitionPenalty / NoRepeatNGram gather on row 0 of input_ids) then touches the base row in one call and the first
                # Medusa head's row in the other, as there.  (Only the LAST position is processed: on the first iteration the reference hands
                # over all P prompt positions x heads and consumes the last, medusa_utils.py:446-449.)
                zp = torch.cat([run(ids_t, z[:1]), run(ids_t, z[1:])], dim=0) if z.shape[0] > 1 else run(ids_t, z)
                cand = torch.argmax(zp, dim=-1)                                   # generate_candidates, top-1 chain (medusa_utils.py:446-458)
                v = run(ids_t, engine_pass(cand.tolist(), L, True)[0])            # verify pass (medusa_utils.py:494-521), same input_ids
                a = self._accept_length(v, cand, gp)
                if a == 0:
                    emit = [int(cand[0]), int(torch.argmax(v[0]))]               # model.py:710-713, medusa_utils.py:636-641
                    kv = L + 1
                else:
                    emit = [int(t) for t in cand[: a + 1]]
                    kv = L + a
                ids += emit
                hist[a] += 1; n_iter += 1; n_tok += len(emit)
                if streamer is not None:
                    streamer.put(torch.tensor(emit, dtype=torch.long))
                stop = bool(host_crit) and any(bool(torch.as_tensor(c(torch.tensor([ids], dtype=torch.long), None)).all()) for c in host_crit)
                if stop or eos in emit or len(ids) >= gp.max_length or len(ids) + K >= gp.hard_max_length:     # model.py:774-793
                    break
            if eos in ids[P:]:                                                    # post-EOS overwrite, model.py:798-810
                j = ids.index(eos, P)
                ids = ids[: j + 1] + [eos] * (len(ids) - j - 1)
            seqs.append(ids)
            if streamer is not None:
                streamer.end()
        self.last_stats = dict(iterations=n_iter, iterations_launched=n_iter, tokens_emitted=n_tok, accept_hist=hist, host_processors=len(procs))
        return seqs

    def _outputs(self, seqs, gp, return_dict_in_generate, return_segments):
        """Default: the padded LongTensor.  ``return_dict_in_generate`` / ``return_segments``: the reference's dict form
        ``{"sequences": ..., ["segments": ...]}`` (model.py:1747-1779; one segment per clip, short-form only)."""
        return self._wrap_outputs(self._pad(seqs, gp), [len(gp.prompt)] * len(seqs), gp.pad_token_id, gp.eos_token_id,
                                  return_dict_in_generate, return_segments)

    def _wrap_outputs(self, t, prompt_lens, pad, eos, return_dict_in_generate, return_segments):
        """``return_dict_in_generate``: a GenerateEncoderDecoderOutput (model.py:812-823, :1715-1742); ``return_segments`` alone:
        the dict {"sequences", "segments"} of model.py:1764-1779 (one segment per clip, short-form only)."""
        if not return_dict_in_generate and not return_segments:
            return t
        segs = None
        if return_segments and not return_dict_in_generate:
            segs = [[{"start": torch.tensor(0.0), "end": torch.tensor(30.0 * self.config.max_source_positions / 1500.0),
                      "tokens": t[i, P:][t[i, P:] != pad] if pad != eos else t[i, P:],
                      "result": t[i]}] for i, P in enumerate(prompt_lens)]
        if return_dict_in_generate:
            # the reference returns the plain ModelOutput here (model.py:1715-1745 returns before the segments dict is built):
            # no extra key, so to_tuple() / integer indexing keep HF's positions
            return GenerateEncoderDecoderOutput(t)
        return {"sequences": t, "segments": segs}

    def _pad(self, seqs: List[List[int]], gp: GenParams) -> torch.Tensor:
        """G3: strip trailing pad/eos beyond the first EOS, right-pad to a tensor (model.py:1929-1973,1747-1762)."""
        out = []
        for s in seqs:
            if gp.eos_token_id in s[len(gp.prompt):]:
                j = s.index(gp.eos_token_id, len(gp.prompt))
                s = s[: j + 1]
            out.append(s)
        T = max(len(s) for s in out)
        t = torch.full((len(out), T), gp.pad_token_id, dtype=torch.long)
        for i, s in enumerate(out):
            t[i, : len(s)] = torch.tensor(s, dtype=torch.long)
        return t.to(self.device)

    # ---- beyond the reference's generate(): language detection, long-form chunking, data-parallel sharding -------------
    @torch.no_grad()
    def detect_language(self, input_features: torch.Tensor) -> List[str]:
        """Language token of every clip: argmax over the language ids of the base logits after <|startoftranscript|>
        (HF WhisperGenerationMixin.detect_language, which the reference reaches through _retrieve_init_tokens when
        `language` is None, model.py:1519-1537; with the Medusa model the first of the stacked heads is the base head)."""
        cfg = self.config
        if not cfg.is_multilingual:
            raise ValueError("language detection needs a multilingual checkpoint")
        feats = input_features.to(self.device, torch.float32).contiguous()
        B = feats.shape[0]
        if B > self._max_batch:
            self.set_max_batch(B)
        self.engine.encode(feats)
        z = self.engine.forward_logits([[cfg.decoder_start_token_id]] * B, 0, True)[0, :, 0]      # [B, V]
        toks = [k for k in sorted(cfg.lang_to_id, key=lambda k: cfg.lang_to_id[k]) if 0 <= cfg.lang_to_id[k] < cfg.vocab_size]
        if not toks:
            raise ValueError("language detection: no language token of lang_to_id lies inside the vocabulary")
        ids = torch.tensor([cfg.lang_to_id[t] for t in toks])
        best = z[:, ids].argmax(-1)
        return [toks[int(i)] for i in best]

    def _generate_detecting_language(self, input_features, kw):
        langs = self.detect_language(input_features)
        rdg, rseg = kw.pop("return_dict_in_generate", None), kw.pop("return_segments", False)
        groups: Dict[str, List[int]] = {}
        for i, l in enumerate(langs):
            groups.setdefault(l, []).append(i)
        rows: List[Optional[torch.Tensor]] = [None] * len(langs)
        plens = [0] * len(langs)
        for l, idx in groups.items():
            # one group = every clip in its original order: the engine still holds the encoder pass detect_language() ran
            reuse = {"_encoded_batch": len(langs)} if len(groups) == 1 else {}
            out = self.generate(input_features[idx], language=l, _language_resolved=True, **reuse, **kw)
            for j, i in enumerate(idx):
                rows[i] = out[j]
                plens[i] = len(self._last_prompt)
        T = max(r.numel() for r in rows)
        t = torch.full((len(rows), T), self.config.pad_token_id, dtype=torch.long, device=self.device)
        for i, r in enumerate(rows):
            t[i, : r.numel()] = r
        self.detected_languages = langs
        return self._wrap_outputs(t, plens, self.config.pad_token_id, self.config.eos_token_id, rdg, rseg)

    def _generate_longform(self, input_features, kw):
        """`chunk_longform=True`: clips longer than 30 s (the reference raises, model.py:1213-1214) are cut into 30 s windows,
        the windows of all clips are decoded as ONE batch of independent streams (that is what the engine's batch is), and
        each clip's windows are concatenated: prompt once, then the generated ids of every window without its EOS / padding.
        No timestamps, no conditioning on the previous window (both unsupported with Medusa in the reference as well)."""
        cfg = self.config
        kw.pop("chunk_longform", None)
        F = cfg.n_mel_frames
        B, _, T = input_features.shape
        n = -(-T // F)
        x = torch.zeros(B, cfg.num_mel_bins, n * F, dtype=torch.float32, device=self.device)
        x[..., :T] = input_features.to(self.device, torch.float32)
        # zero-padded log-mel frames are not silence (silence maps to a constant negative level): pad in the feature domain
        # with the clip's own minimum, which is what the log-mel of digital silence clamps to
        if n * F > T:
            x[..., T:] = input_features.to(self.device, torch.float32).amin(dim=(1, 2), keepdim=True)
        win = x.view(B, cfg.num_mel_bins, n, F).permute(0, 2, 1, 3).reshape(B * n, cfg.num_mel_bins, F).contiguous()
        # the language is detected ONCE per clip, on its first window, and every window of the clip is decoded in that language
        # (clips of different languages go through generate() as separate groups, each with its own prompt)
        if kw.get("language") is None and cfg.is_multilingual and kw.get("detect_language", True):
            first = self.detect_language(win[::n])
            kw["detect_language"] = False
            langs = first
        else:
            langs = [kw.get("language")] * B
        kw.pop("language", None)
        rows, prompts = [None] * (B * n), [None] * B
        for l in sorted(set(langs), key=lambda v: (v is None, v or "")):
            clips = [b for b in range(B) if langs[b] == l]
            widx = [b * n + j for b in clips for j in range(n)]
            o = self.generate(win[widx], language=l, **kw)
            for q, wi in enumerate(widx):
                rows[wi] = o[q]
            for b in clips:
                prompts[b] = list(self._last_prompt)
        eos, pad = cfg.eos_token_id, cfg.pad_token_id
        seqs = []
        for b in range(B):
            ids = list(prompts[b])
            P = len(ids)
            for j in range(n):
                row = rows[b * n + j][P:].tolist()
                for t in row:
                    if t == eos or t == pad:
                        break
                    ids.append(t)
            seqs.append(ids + [eos])
        Tm = max(len(s_) for s_ in seqs)
        t = torch.full((B, Tm), pad, dtype=torch.long, device=self.device)
        for i, s_ in enumerate(seqs):
            t[i, : len(s_)] = torch.tensor(s_, dtype=torch.long)
        return t

    @torch.no_grad()
    def generate_sharded(self, input_features: torch.Tensor, **kw) -> torch.Tensor:
        """Data-parallel generate() over the ranks of an initialised torch.distributed group (one process per GPU, SURVEY.md
        §8e): stream s is decoded on rank s mod world (dist.shard_streams), every rank receives all sequences
        (dist.gather_token_lists: ragged int lists, no device collective on the data path).  Every rank passes the SAME
        `input_features` (or at least its own shard's rows at the right indices)."""
        import torch.distributed as td
        from . import dist as _dist
        B = input_features.shape[0]
        if not (td.is_available() and td.is_initialized()) or td.get_world_size() == 1:
            return self.generate(input_features, **kw)
        rank, world = td.get_rank(), td.get_world_size()
        mine = _dist.shard_streams(B, rank, world)
        local = []
        if mine:
            out = self.generate(input_features[mine], **kw)
            local = [(s_, out[j].tolist()) for j, s_ in enumerate(mine)]
        seqs = _dist.gather_token_lists(local)
        pad = self.config.pad_token_id
        T = max(len(s_) for s_ in seqs)
        t = torch.full((B, T), pad, dtype=torch.long, device=self.device)
        for i, s_ in enumerate(seqs):
            t[i, : len(s_)] = torch.tensor(s_, dtype=torch.long)
        return t

    def generate_from_wav(self, wav, **kw) -> torch.Tensor:
        """log-mel on the GPU, then ``generate`` — the whole hot path of SURVEY.md §8a in one call."""
        return self.generate(self.extract_features(wav), **kw)

    # ---- forward ----------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, input_features: Optional[torch.Tensor] = None, attention_mask=None,
                decoder_input_ids: Optional[torch.Tensor] = None, decoder_attention_mask=None, head_mask=None,
                decoder_head_mask=None, cross_attn_head_mask=None, encoder_outputs=None, past_key_values=None,
                decoder_inputs_embeds=None, decoder_position_ids=None, labels=None, use_cache: Optional[bool] = None,
                output_attentions: Optional[bool] = None, output_hidden_states: Optional[bool] = None,
                return_dict: Optional[bool] = None, disable_medusa: Optional[bool] = False, **kwargs):
        """One decoder pass over ``decoder_input_ids`` [B, T] — the reference's ``forward`` (model.py:1223-1347), same keywords:

        * ``input_features``: the encoder runs first; ``encoder_outputs`` (a tuple / ModelOutput whose first element is the last
          hidden state [B, n_ctx, d], or that tensor) replaces the encoder pass (model.py:1232 -> HF WhisperModel.forward) — the
          object a previous ``forward`` returned is recognised and costs nothing; neither: the last encoded batch is reused;
        * ``past_key_values`` / ``use_cache``: the KV cache lives in the engine (contiguous rows per layer / stream / head); the
          returned ``past_key_values`` is an ``EngineKVCache`` handle saying how many positions are cached, and a following call that
          passes it back appends its tokens behind them (``get_seq_length()`` like HF's cache classes).  ``decoder_position_ids``
          override the start position as in the reference;
        * ``return_dict=False`` gives the reference's tuple ``(logits, past_key_values, encoder_last_hidden_state)`` (``use_cache=None``
          counts as True, HF's ``config.use_cache`` default); the hidden state is copied from the device once per encoder pass;
        * masks, ``decoder_inputs_embeds``, attentions / hidden states and ``labels`` (training) are not part of the engine: they raise.

        ``logits`` are ``[K+1 (1 with disable_medusa), B, T, V]``.  The engine evaluates 16 positions per pass: longer inputs go
        through in 16-token chunks, each appending its K/V rows behind the previous chunk's."""
        if decoder_input_ids is None:
            raise ValueError("decoder_input_ids is required")
        if labels is not None or kwargs.get("labels") is not None:
            raise NotImplementedError("training (labels / loss) is out of scope for the inference engine")
        if decoder_attention_mask is not None:
            # a mask that masks nothing is what HF's own generate() passes along; anything else would need per-position masking in the
            # engine's causal attention (not built: the reference's loop never produces one)
            if not bool(torch.as_tensor(decoder_attention_mask).bool().all()):
                raise NotImplementedError("forward(decoder_attention_mask=...) with masked positions is not supported by the HIP engine")
            decoder_attention_mask = None
        for name, v in (("decoder_attention_mask", decoder_attention_mask), ("head_mask", head_mask), ("decoder_head_mask", decoder_head_mask),
                        ("cross_attn_head_mask", cross_attn_head_mask), ("decoder_inputs_embeds", decoder_inputs_embeds)):
            if v is not None:
                raise NotImplementedError(f"forward({name}=...) is not supported by the HIP engine")
        if output_attentions or output_hidden_states:
            raise NotImplementedError("attention weights / hidden states are not materialised by the HIP engine")
        B = int(decoder_input_ids.shape[0])
        if B > self._max_batch and (input_features is not None or encoder_outputs is not None):
            self.set_max_batch(B)                                           # a new encoder pass for more streams than the context holds
        eng = self.engine
        # ---- encoder side
        enc_ref = None
        if encoder_outputs is not None:
            stamp = getattr(encoder_outputs, "_wm_stamp", None)
            if stamp is not None and stamp is getattr(eng, "_enc_stamp", None):
                enc_ref = encoder_outputs                                   # our own handle, still resident: nothing to do
            else:
                first = encoder_outputs if isinstance(encoder_outputs, torch.Tensor) else (
                    encoder_outputs.last_hidden_state if hasattr(encoder_outputs, "last_hidden_state") else encoder_outputs[0])
                eng.set_encoder_output(first)
        elif input_features is not None:
            eng.encode(input_features.to(self.device, torch.float32).contiguous())
        elif getattr(eng, "_enc_stamp", None) is None:
            raise ValueError("forward() needs input_features or encoder_outputs (nothing has been encoded on this model yet)")
        # ---- start position
        pos0 = 0
        if past_key_values is not None:
            if not isinstance(past_key_values, EngineKVCache):
                raise NotImplementedError("past_key_values must be the EngineKVCache a previous forward(use_cache=True) returned: "
                                          "the KV cache is engine state (contiguous HBM rows), not a tuple of tensors")
            if past_key_values.engine_id != id(eng) or past_key_values.enc_stamp is not eng._enc_stamp or past_key_values.batch != B \
                    or past_key_values.kv_stamp is not eng._kv_stamp:
                raise ValueError("past_key_values belongs to another engine, encoder pass or batch size, or a generate() call has "
                                 "reused the cache since")
            pos0 = past_key_values.get_seq_length()
        if decoder_position_ids is not None:
            pos0 = int(torch.as_tensor(decoder_position_ids).flatten()[0])
        toks = decoder_input_ids.tolist()
        T = len(toks[0])
        if pos0 + T > self.config.max_target_positions:
            raise ValueError(f"decoder_input_ids of length {T} at position {pos0} exceed max_target_positions "
                             f"{self.config.max_target_positions}")
        dm = bool(disable_medusa)
        if T <= 16:
            logits = eng.forward_logits(toks, pos0, dm)
        else:
            logits = torch.cat([eng.forward_logits([row[c: c + 16] for row in toks], pos0 + c, dm) for c in range(0, T, 16)], dim=2)
        # use_cache=None means config.use_cache = True, as in HF (the reference's tuple is (logits, past_key_values, encoder state))
        want_cache = (use_cache is None) or bool(use_cache) or past_key_values is not None
        # the handle carries the stamp the pass just set (engine.forward_logits): older handles — positions now overwritten — are refused
        pkv = EngineKVCache(id(eng), eng._enc_stamp, eng._kv_stamp, B, pos0 + T) if want_cache else None
        if enc_ref is None:
            enc_ref = EngineEncoderOutput(eng, B, eng._enc_stamp)
        if return_dict is False:
            return (logits, pkv, enc_ref.last_hidden_state) if pkv is not None else (logits, enc_ref.last_hidden_state)
        return MedusaForwardOutput(logits=logits, past_key_values=pkv, encoder_outputs=enc_ref)

    __call__ = forward


def get_model(path, device=None):
    """reference model.py:2079-2097."""
    return WhisperMedusaModel.from_pretrained(path, device=device)
