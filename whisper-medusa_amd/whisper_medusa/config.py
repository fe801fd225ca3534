"""Model + generation configuration for the MI355X Whisper-Medusa engine.

`MedusaConfig` reads the same ``config.json`` schema the reference checkpoint
carries (reference: whisper_medusa/utils/config_and_args.py:17-62 — the Medusa
fields — on top of HF ``WhisperConfig`` fields), but is a plain dataclass: no
hub lookup, no HF model instantiation.

`GenParams` holds what the reference pulls from ``generation_config`` for one
``generate()`` call (reference: models/model.py:1168-1207 processors,
:1877-1884 temperature forcing, :774-793 stop rules; medusa_utils.py:14-18
posterior defaults).
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, field, asdict
from typing import List, Optional, Tuple

HEAD_DIM = 64  # every Whisper size uses 64-wide attention heads

# Whisper's multilingual language tokens (<|en|> = 50259 ... in this order: 99 languages for large-v2) and the names HF's
# `language=` argument accepts besides the codes (HF tokenization_whisper.py LANGUAGES / TO_LANGUAGE_CODE).  Used when a
# checkpoint ships no generation_config.json with its own `lang_to_id`.
WHISPER_LANGUAGES = {
    "en": "english", "zh": "chinese", "de": "german", "es": "spanish", "ru": "russian", "ko": "korean", "fr": "french",
    "ja": "japanese", "pt": "portuguese", "tr": "turkish", "pl": "polish", "ca": "catalan", "nl": "dutch", "ar": "arabic",
    "sv": "swedish", "it": "italian", "id": "indonesian", "hi": "hindi", "fi": "finnish", "vi": "vietnamese", "he": "hebrew",
    "uk": "ukrainian", "el": "greek", "ms": "malay", "cs": "czech", "ro": "romanian", "da": "danish", "hu": "hungarian",
    "ta": "tamil", "no": "norwegian", "th": "thai", "ur": "urdu", "hr": "croatian", "bg": "bulgarian", "lt": "lithuanian",
    "la": "latin", "mi": "maori", "ml": "malayalam", "cy": "welsh", "sk": "slovak", "te": "telugu", "fa": "persian",
    "lv": "latvian", "bn": "bengali", "sr": "serbian", "az": "azerbaijani", "sl": "slovenian", "kn": "kannada", "et": "estonian",
    "mk": "macedonian", "br": "breton", "eu": "basque", "is": "icelandic", "hy": "armenian", "ne": "nepali", "mn": "mongolian",
    "bs": "bosnian", "kk": "kazakh", "sq": "albanian", "sw": "swahili", "gl": "galician", "mr": "marathi", "pa": "punjabi",
    "si": "sinhala", "km": "khmer", "sn": "shona", "yo": "yoruba", "so": "somali", "af": "afrikaans", "oc": "occitan",
    "ka": "georgian", "be": "belarusian", "tg": "tajik", "sd": "sindhi", "gu": "gujarati", "am": "amharic", "yi": "yiddish",
    "lo": "lao", "uz": "uzbek", "fo": "faroese", "ht": "haitian creole", "ps": "pashto", "tk": "turkmen", "nn": "nynorsk",
    "mt": "maltese", "sa": "sanskrit", "lb": "luxembourgish", "my": "myanmar", "bo": "tibetan", "tl": "tagalog", "mg": "malagasy",
    "as": "assamese", "tt": "tatar", "haw": "hawaiian", "ln": "lingala", "ha": "hausa", "ba": "bashkir", "jw": "javanese",
    "su": "sundanese",
}
LANGUAGE_ALIASES = {"burmese": "my", "valencian": "ca", "flemish": "nl", "haitian": "ht", "letzeburgesch": "lb", "pushto": "ps",
                    "panjabi": "pa", "moldavian": "ro", "moldovan": "ro", "sinhalese": "si", "castilian": "es", "mandarin": "zh"}


def default_lang_to_id(first_id: int = 50259) -> dict:
    return {f"<|{code}|>": first_id + i for i, code in enumerate(WHISPER_LANGUAGES)}


def language_token(language: str) -> str:
    """'en' | 'english' | '<|en|>' -> '<|en|>' (what HF's _retrieve_init_tokens accepts); ValueError otherwise."""
    l = language.strip().lower()
    if l.startswith("<|") and l.endswith("|>"):
        return l
    if l in WHISPER_LANGUAGES:
        return f"<|{l}|>"
    names = {v: k for k, v in WHISPER_LANGUAGES.items()}
    names.update(LANGUAGE_ALIASES)
    if l in names:
        return f"<|{names[l]}|>"
    raise ValueError(f"Unsupported language: {language}")


HEADS_LINEAR = "base_head"      # reference: model.py:221-230
HEADS_BLOCK = "medusa_block"


MAX_TREE_NODES = 64      # the verify pass of a stream is up to four 16-row MFMA query tiles
MAX_TREE_PATHS = 32
MAX_TREE_TOPK = 4


def tree_buffers(medusa_choices) -> dict:
    """Static index buffers of the candidate tree for ``medusa_choices = [1, c_1, .., c_K]`` (head k contributes its top-c_k
    tokens, the tree is their cartesian product) — the same objects as the reference's ``generate_medusa_buffers``
    (medusa_utils.py:305-421; pinned against it in tests/test_oracle_golden.py), derived directly:
      flat candidate list  [base argmax | top-c_1 of head 1 | top-c_2 of head 2 | ...]
      node (depth i, index j inside the depth, j in [0, prod_{l<=i} c_l)):  token = flat[cumsum_{i-1} + j % c_i],
      parent = node (i-1, j // c_i); nodes are numbered depth by depth.
    Returns tree_indices, depth (= medusa_position_ids), parent, anc_mask (bit n of anc_mask[m]: node n is m or an ancestor of m
    — the rows of medusa_attn_mask), retrieve_indices [n_paths][K+1] and the per-head top-k."""
    c = [int(x) for x in medusa_choices]
    if not c or c[0] != 1 or any(x < 1 for x in c):
        raise ValueError("medusa_choices must be [1, c_1, ..., c_K] with c_k >= 1")
    K = len(c) - 1
    cumprod, cumsum = [], []
    p = a = 0
    p = 1
    for x in c:
        p *= x; a += x
        cumprod.append(p); cumsum.append(a)
    tree_indices, depth, parent = [], [], []
    start = [0]
    for i in range(K + 1):
        start.append(start[-1] + cumprod[i])
    for i in range(K + 1):
        off = cumsum[i - 1] if i else 0
        for j in range(cumprod[i]):
            tree_indices.append(off + j % c[i])
            depth.append(i)
            parent.append(-1 if i == 0 else start[i - 1] + j // c[i])
    n = len(tree_indices)
    anc = []
    for m in range(n):
        bits, q = 0, m
        while q >= 0:
            bits |= 1 << q
            q = parent[q]
        anc.append(bits)
    n_paths = cumprod[-1]
    retrieve = []
    for path in range(n_paths):
        row = []
        for i in range(K + 1):
            row.append(start[i] + path // (n_paths // cumprod[i]))
        retrieve.append(row)
    return dict(tree_indices=tree_indices, depth=depth, parent=parent, anc_mask=anc, retrieve_indices=retrieve,
                n_nodes=n, n_paths=n_paths, topk=c[1:], n_flat=cumsum[-1])


@dataclass
class MedusaConfig:
    # --- Whisper dims (HF WhisperConfig names) ---
    d_model: int = 1280
    encoder_layers: int = 32
    decoder_layers: int = 32
    encoder_attention_heads: int = 20
    decoder_attention_heads: int = 20
    encoder_ffn_dim: int = 5120
    decoder_ffn_dim: int = 5120
    vocab_size: int = 51865
    num_mel_bins: int = 80
    max_source_positions: int = 1500
    max_target_positions: int = 448
    # --- Medusa fields (reference config_and_args.py:35-47) ---
    medusa_num_heads: int = 10
    medusa_num_layers: int = 1
    medusa_hidden_size: int = 1280
    medusa_heads_type: str = HEADS_LINEAR
    medusa_choices: List[int] = field(default_factory=lambda: [1] * 11)
    whisper_model_name: str = "openai/whisper-large-v2"
    output_whisper_original: bool = False
    # --- token ids / generation defaults (HF generation_config.json) ---
    eos_token_id: int = 50257
    pad_token_id: int = 50257
    decoder_start_token_id: int = 50258
    is_multilingual: bool = True
    lang_to_id: dict = field(default_factory=default_lang_to_id)
    prev_sot_token_id: int = 50361            # <|startofprev|> (prompt_ids conditioning)
    task_to_id: dict = field(default_factory=lambda: {"transcribe": 50359, "translate": 50358})
    no_timestamps_token_id: int = 50363
    suppress_tokens: Optional[List[int]] = None
    begin_suppress_tokens: Optional[List[int]] = field(default_factory=lambda: [220, 50257])
    max_length: int = 448
    posterior_threshold: float = 0.09   # medusa_utils.py:17
    posterior_alpha: float = 0.3        # medusa_utils.py:18

    # ------------------------------------------------------------------
    def __post_init__(self):
        if self.medusa_heads_type not in (HEADS_LINEAR, HEADS_BLOCK):
            # same error class/wording as reference model.py:225-229
            raise ValueError(
                f"medusa_heads_type {self.medusa_heads_type} is not supported, "
                f"select from {[HEADS_LINEAR, HEADS_BLOCK]}"
            )
        if self.d_model != self.encoder_attention_heads * HEAD_DIM or \
           self.d_model != self.decoder_attention_heads * HEAD_DIM:
            raise ValueError("engine supports head_dim == 64 only (all Whisper sizes)")
        if self.medusa_num_layers != 1:
            raise ValueError("engine supports medusa_num_layers == 1 (the only value the reference ships)")
        if self.medusa_hidden_size != self.d_model:
            raise ValueError("medusa_hidden_size must equal d_model (residual head, model.py:210)")
        K = self.medusa_num_heads
        ch = [int(x) for x in self.medusa_choices]
        if len(ch) != K + 1 or ch[0] != 1:
            raise ValueError("medusa_choices must have medusa_num_heads + 1 entries and start with 1 (the base head's argmax)")
        if K + 1 > 16:
            raise ValueError("engine supports at most 15 Medusa heads (verify pass is one 16-row MFMA tile)")
        if ch != [1] * (K + 1):
            # a real candidate tree (top-k > 1): beyond the reference, which builds the tree mask and never applies it
            # (medusa_utils.py:494-516).  The verify pass is up to four 16-row query tiles per stream.
            tb = tree_buffers(ch)
            if tb["n_nodes"] > MAX_TREE_NODES or tb["n_paths"] > MAX_TREE_PATHS or max(tb["topk"]) > MAX_TREE_TOPK:
                raise ValueError(f"candidate tree {ch}: {tb['n_nodes']} nodes / {tb['n_paths']} paths / top-{max(tb['topk'])}; the engine "
                                 f"supports <= {MAX_TREE_NODES} nodes, <= {MAX_TREE_PATHS} paths, top-k <= {MAX_TREE_TOPK}")

    @property
    def is_tree(self) -> bool:
        return [int(x) for x in self.medusa_choices] != [1] * (self.medusa_num_heads + 1)

    @property
    def is_block(self) -> bool:
        return self.medusa_heads_type == HEADS_BLOCK

    @property
    def n_heads(self) -> int:
        return self.decoder_attention_heads

    @property
    def n_kv_layers(self) -> int:
        """decoder layers that own a KV cache slot (Block adds slot L, model.py:248-256)."""
        return self.decoder_layers + (1 if self.is_block else 0)

    @property
    def n_mel_frames(self) -> int:
        return 2 * self.max_source_positions

    def to_dict(self) -> dict:
        return asdict(self)

    @classmethod
    def from_dict(cls, d: dict) -> "MedusaConfig":
        known = {f for f in cls.__dataclass_fields__}
        kw = {k: v for k, v in d.items() if k in known and v is not None}
        return cls(**kw)

    @classmethod
    def from_pretrained(cls, path: str) -> "MedusaConfig":
        """Read ``config.json`` (+ ``generation_config.json`` if present) from a checkpoint dir."""
        with open(os.path.join(path, "config.json")) as f:
            d = json.load(f)
        gpath = os.path.join(path, "generation_config.json")
        if os.path.exists(gpath):
            with open(gpath) as f:
                g = json.load(f)
            for k in ("eos_token_id", "pad_token_id", "decoder_start_token_id", "is_multilingual",
                      "lang_to_id", "task_to_id", "no_timestamps_token_id", "suppress_tokens",
                      "begin_suppress_tokens", "max_length", "posterior_threshold", "posterior_alpha"):
                if k in g and g[k] is not None:
                    d[k] = g[k]
        if "medusa_choices" not in d and "medusa_num_heads" in d:
            d["medusa_choices"] = [1] * (d["medusa_num_heads"] + 1)
        return cls.from_dict(d)

    def save_pretrained(self, path: str) -> None:
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump(self.to_dict(), f, indent=1)

    # named shapes --------------------------------------------------------
    @classmethod
    def large_v2(cls, heads_type: str = HEADS_LINEAR, K: int = 10, medusa_choices=None) -> "MedusaConfig":
        return cls(medusa_heads_type=heads_type, medusa_num_heads=K, medusa_choices=list(medusa_choices or [1] * (K + 1)))

    @classmethod
    def tiny_en(cls, heads_type: str = HEADS_LINEAR, K: int = 4) -> "MedusaConfig":
        return cls(d_model=384, encoder_layers=4, decoder_layers=4, encoder_attention_heads=6,
                   decoder_attention_heads=6, encoder_ffn_dim=1536, decoder_ffn_dim=1536,
                   vocab_size=51864, medusa_num_heads=K, medusa_hidden_size=384,
                   medusa_choices=[1] * (K + 1), medusa_heads_type=heads_type,
                   whisper_model_name="openai/whisper-tiny.en",
                   eos_token_id=50256, pad_token_id=50256, decoder_start_token_id=50257,
                   is_multilingual=False, lang_to_id={}, task_to_id={},
                   no_timestamps_token_id=50362, begin_suppress_tokens=[220, 50256])

    @classmethod
    def micro(cls, heads_type: str = HEADS_LINEAR, K: int = 4, d_model: int = 128, layers: int = 2,
              vocab: int = 1031, n_ctx: int = 96, n_tgt: int = 64, medusa_choices=None) -> "MedusaConfig":
        """A few-hundred-K-parameter shape for fast parity tests (not a real Whisper size)."""
        return cls(d_model=d_model, encoder_layers=layers, decoder_layers=layers,
                   encoder_attention_heads=d_model // 64, decoder_attention_heads=d_model // 64,
                   encoder_ffn_dim=4 * d_model, decoder_ffn_dim=4 * d_model, vocab_size=vocab,
                   max_source_positions=n_ctx, max_target_positions=n_tgt,
                   medusa_num_heads=K, medusa_hidden_size=d_model, medusa_choices=list(medusa_choices or [1] * (K + 1)),
                   medusa_heads_type=heads_type, whisper_model_name="micro",
                   eos_token_id=vocab - 3, pad_token_id=vocab - 3, decoder_start_token_id=vocab - 2,
                   is_multilingual=False, lang_to_id={}, task_to_id={},
                   no_timestamps_token_id=vocab - 1, begin_suppress_tokens=[7, vocab - 3],
                   max_length=n_tgt)


ACCEPT_TYPICAL = 1   # temperature != 0 branch, medusa_utils.py:562-588 (what generate() really runs)
ACCEPT_GREEDY = 0    # temperature == 0 branch, medusa_utils.py:547-560


@dataclass
class GenParams:
    """Everything one decode run needs besides the weights (mirrors the C-ABI ``wm_gen_params``)."""
    prompt: List[int]
    eos_token_id: int
    pad_token_id: int
    suppress_tokens: List[int] = field(default_factory=list)
    begin_suppress_tokens: List[int] = field(default_factory=list)
    max_length: int = 448            # MaxLengthCriteria: stop when L >= max_length
    hard_max_length: int = 448       # model.py:789-793: stop when L + K >= self.generation_config.max_length
    exp_decay: Optional[Tuple[int, float]] = None   # (start, factor); start is relative to the prompt length
    posterior_threshold: float = 0.09
    posterior_alpha: float = 0.3
    accept_mode: int = ACCEPT_TYPICAL   # G4: generate() forces temperature=1.0 -> typical acceptance
    temperature: float = 1.0
    vanilla: bool = False            # anchor: plain greedy decoding with head 0 / base logits
    force_accept: int = -1           # benchmark knob (wm.h): >= 0 forces the accept length of every iteration; -1 = off
    begin_suppress_index: Optional[int] = None
    # ^ sequence length at which SuppressTokensAtBeginLogitsProcessor fires.  None = len(prompt) (no prompt_ids: the prompt IS the
    #   init tokens).  With `prompt_ids` the reference passes begin_index = init_tokens.shape[1] (model.py:1537, 1551, 1640-1644),
    #   i.e. the number of init tokens WITHOUT the prepended previous-text ids, so the begin suppression never fires there.

    @property
    def begin_index(self) -> int:
        return len(self.prompt) if self.begin_suppress_index is None else int(self.begin_suppress_index)
