"""Shared test helpers: seeded checkpoints, clips and the GenParams used by the golden runs."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "whisper-medusa_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

from whisper_medusa.config import MedusaConfig, GenParams, ACCEPT_TYPICAL, ACCEPT_GREEDY  # noqa: E402
from whisper_medusa import synth  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def default_act_f16():
    """The decode numerics contract this process defaults to (engine AND oracle): WM_ACT, default f16 (whisper_medusa/engine.py)."""
    return os.environ.get("WM_ACT", "f16").lower() in ("f16", "fp16")

# (tag, config factory, checkpoint seed, max_new) — must match oracle/make_golden.py:main()
GOLDEN_MODELS = {
    "micro": (lambda: MedusaConfig.micro(K=4), 11, 40),
    "micro10": (lambda: MedusaConfig.micro(K=10, d_model=128, layers=2), 12, 40),
    "tiny": (lambda: MedusaConfig.tiny_en(K=4), 0, 24),
}
# Medusa-Block runs minted from the reference's own Block forward (oracle/make_golden.py:build_ref shims)
GOLDEN_BLOCK_MODELS = {
    "microblock": (lambda: MedusaConfig.micro(K=4, heads_type="medusa_block"), 13, 40),
    "micro10block": (lambda: MedusaConfig.micro(K=10, heads_type="medusa_block", d_model=128, layers=2), 14, 40),
    "tinyblock": (lambda: MedusaConfig.tiny_en("medusa_block", K=4), 1, 24),
}


def golden_gen_params(cfg, mode, max_new, suppress_eos=True, exp_decay=(6, 1.3)):
    """Same recipe as oracle/make_golden.py:gen_params_for."""
    prompt = synth.default_prompt(cfg)
    sup = [cfg.eos_token_id] if suppress_eos else []
    return GenParams(prompt=prompt, eos_token_id=cfg.eos_token_id, pad_token_id=cfg.pad_token_id,
                     suppress_tokens=sorted(set(sup + [3, 5])), begin_suppress_tokens=list(cfg.begin_suppress_tokens),
                     max_length=min(len(prompt) + max_new, cfg.max_target_positions),
                     hard_max_length=cfg.max_length, exp_decay=exp_decay,
                     accept_mode=mode, temperature=1.0 if mode == ACCEPT_TYPICAL else 0.0)


def clip_for(cfg, i=0):
    return synth.synth_clip(i, n_samples=cfg.n_mel_frames * 160)


# Parity evidence of one pytest session: every oracle comparison is counted, every followed numerical tie recorded, and the
# agreement tables of the cross-mode tests kept; tests/conftest.py prints ONE summary line at the end of the run (visible in the
# driver's -q log) and writes tests/parity_report.json (+ gpurun_out/ when that exists).
PARITY = {"runs": 0, "strict_equal": 0, "ties": [], "tables": {}}


def record_table(name, **numbers):
    PARITY["tables"][name] = numbers


def check_tokens(orc, enc_b, gp, got, label="", max_ties=2, tol_logit=5e-4):
    """Token parity of one engine run against the oracle: strictly identical ids, or — when the first difference sits at a
    decision whose margin in the ORACLE is below the numerical difference of two correct implementations (an argmax between
    logits < tol_logit apart, p_c within 2e-3 relative of the threshold) — identical along the engine's admissible branch
    (oracle.decode_following).  Returns (accept lengths, ties); ties are printed: they are part of the evidence."""
    ref = orc.decode(enc_b, gp)
    PARITY["runs"] += 1
    if got == ref.ids:
        PARITY["strict_equal"] += 1
        return ref.accept_lengths, []
    ok, ties, ids = orc.decode_following(enc_b, gp, got, tol_logit=tol_logit)
    first = next((i for i, (a, b) in enumerate(zip(got, ref.ids)) if a != b), min(len(got), len(ref.ids)))
    print(f"parity[{label}]: strict comparison differs at index {first}; numerical ties followed: {ties}")
    assert ok and 0 < len(ties) <= max_ties and all("kind" in t for t in ties), (label, first, got, ref.ids, ties)
    for t in ties:
        PARITY["ties"].append({"run": str(label), **{k: (float(v) if isinstance(v, (int, float)) else str(v)) for k, v in t.items()}})
    return list(orc.last_follow_accepts), ties


# ---- fp32-pinned tables against ids minted offline (oracle/make_fp32_golden.py -> tests/golden/fp32_pinned_runs.npz) ----------------
FP32_GOLD = os.path.join(GOLD, "fp32_pinned_runs.npz")
FP32_FINGERPRINT_KEY = "whisper_model.model.decoder.layers.0.fc1.weight"


def fp32_golden(tag, sd):
    """The offline table of shape `tag`; refuses a checkpoint other than the one the table was minted on (CPU generator, same seed)."""
    import hashlib
    g = np.load(FP32_GOLD)
    sha = hashlib.sha256(sd[FP32_FINGERPRINT_KEY].detach().float().cpu().contiguous().numpy().tobytes()).hexdigest()
    assert bytes(g[f"{tag}_sha"]).decode() == sha, "fp32_pinned_runs.npz was minted on another checkpoint: python oracle/make_fp32_golden.py"
    return g


def fp32_agreement_rows(g, tag, mode, got):
    """Engine ids `got[clip]` against the pinned fp32 oracle's: per clip (clip, ids agreeing before the first divergence, ids generated by the
    oracle, top-2 logit margin of the oracle iteration that emitted the first differing id, its smallest |p_c - threshold|)."""
    m = "typical" if mode == ACCEPT_TYPICAL else "greedy"
    ids, lens, emit = g[f"{tag}_{m}_ids"], g[f"{tag}_{m}_len"], g[f"{tag}_{m}_emit"]
    margin, pcs, P = g[f"{tag}_{m}_margin"], g[f"{tag}_{m}_pc_thr"], int(g[f"{tag}_{m}_prompt_len"])
    rows = []
    for i in range(len(got)):
        ref = [int(t) for t in ids[i][: int(lens[i])]]
        f = next((j for j, (a, b) in enumerate(zip(got[i], ref)) if a != b), min(len(got[i]), len(ref)))
        mg, pc = float("nan"), float("nan")
        if f < len(ref):
            pos = P
            for it in range(emit.shape[1]):
                if emit[i][it] == 0:
                    break
                if pos + int(emit[i][it]) > f:
                    mg, pc = float(margin[i][it]), float(pcs[i][it])
                    break
                pos += int(emit[i][it])
        rows.append((i, f - P, len(ref) - P, mg, pc))
    return rows
