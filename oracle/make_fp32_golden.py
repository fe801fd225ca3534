"""Mint the fp32-pinned token tables offline.  TEST INFRASTRUCTURE ONLY (never imported by the product).

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_fp32_golden.py [tiny] [large]

The oracle's sim="fp32" mode is the one pinned to the reference's own code (tests/test_oracle_golden.py reproduces the runs
`oracle/make_golden.py` minted from /root/reference token for token).  Here that mode runs audio -> tokens — its OWN log-mel, its OWN
fp32 encoder, its own decode loop (model.py:634-793, medusa_utils.py:526-671) — on seeded checkpoints and seeded clips, and the ids go to
`tests/golden/fp32_pinned_runs.npz`.  The GPU tests regenerate the SAME checkpoint (CPU generator, `whisper_medusa.synth`) and the same
clips, run the engine and compare: the 700 s of host-CPU oracle time the live tables cost on the GPU box are spent here,
once.  The live tables stay in the suite behind `-m "gpu and slow"` (`WM_SLOW=1`).

Stored per (shape, mode, clip): the ids, the number of ids every iteration emitted, the top-2 margin of the verify logits' row 0 of every
iteration and (typical acceptance) the smallest |p_c - threshold| of the iteration — what the tests print at a first divergence.
Weights are not stored.  The checkpoint's sha256 (one tensor) is, so a change of the generator is noticed instead of mis-compared.
"""
import hashlib
import os
import sys
import time

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "whisper-medusa_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLD = os.path.join(ROOT, "tests", "golden", "fp32_pinned_runs.npz")

from whisper_medusa.config import MedusaConfig, ACCEPT_TYPICAL, ACCEPT_GREEDY  # noqa: E402
from whisper_medusa import synth  # noqa: E402
from oracle.whisper_medusa_oracle import Oracle, log_mel  # noqa: E402

# (tag) -> config factory, checkpoint seed, synth kwargs, first clip index, clips, new tokens, GenParams recipe
SHAPES = {
    "tiny": dict(cfg=lambda: MedusaConfig.tiny_en(K=4), seed=0, synth={}, clip0=200, clips=16, new=32, recipe="golden"),
    "large": dict(cfg=lambda: MedusaConfig.large_v2("base_head", K=10), seed=0, synth=dict(logit_std=4.5), clip0=300, clips=8, new=48,
                  recipe="bench"),
}
FINGERPRINT_KEY = "whisper_model.model.decoder.layers.0.fc1.weight"


def gen_params(cfg, recipe, mode, new):
    if recipe == "golden":
        from helpers import golden_gen_params
        return golden_gen_params(cfg, mode, new)
    return synth.bench_gen_params(cfg, max_new_tokens=new, accept_mode=mode)


def fingerprint(sd):
    return hashlib.sha256(sd[FINGERPRINT_KEY].contiguous().numpy().tobytes()).hexdigest()


def mint(tag, out):
    s = SHAPES[tag]
    cfg = s["cfg"]()
    t0 = time.time()
    sd = synth.synth_state_dict(cfg, seed=s["seed"], device="cpu", **s["synth"])
    orc = Oracle(cfg, sd, sim="fp32")
    n = cfg.n_mel_frames * 160
    out[f"{tag}_sha"] = np.frombuffer(fingerprint(sd).encode(), dtype=np.uint8)
    out[f"{tag}_meta"] = np.asarray([s["seed"], s["clip0"], s["clips"], s["new"]], dtype=np.int64)
    encs = []
    for i in range(s["clips"]):
        wav = synth.synth_clip(s["clip0"] + i, n)
        encs.append(orc.encode(torch.from_numpy(log_mel(wav, cfg.num_mel_bins, n))))
        print(f"[{tag}] clip {i}: fp32 log-mel + encoder done ({time.time() - t0:.0f} s)", flush=True)
    for mode, mname in ((ACCEPT_GREEDY, "greedy"), (ACCEPT_TYPICAL, "typical")):
        gp = gen_params(cfg, s["recipe"], mode, s["new"])
        ids, emits, margins, pcs = [], [], [], []
        for i in range(s["clips"]):
            r = orc.decode(encs[i], gp, trace=True)
            ids.append(r.ids)
            emits.append([len(t["emit"]) for t in r.trace])
            margins.append([float(torch.topk(t["v"][0], 2).values.diff().abs()) for t in r.trace])
            pcs.append([float((t["p_c"] - t["thr"]).abs().min()) if "p_c" in t else float("nan") for t in r.trace])
            print(f"[{tag}] {mname} clip {i}: {len(r.ids) - len(gp.prompt)} ids in {r.n_iters} iterations ({time.time() - t0:.0f} s)", flush=True)
        L = max(len(x) for x in ids)
        I = max(len(x) for x in emits)
        out[f"{tag}_{mname}_ids"] = np.asarray([x + [-1] * (L - len(x)) for x in ids], dtype=np.int32)
        out[f"{tag}_{mname}_len"] = np.asarray([len(x) for x in ids], dtype=np.int32)
        out[f"{tag}_{mname}_emit"] = np.asarray([x + [0] * (I - len(x)) for x in emits], dtype=np.int32)
        out[f"{tag}_{mname}_margin"] = np.asarray([x + [np.nan] * (I - len(x)) for x in margins], dtype=np.float32)
        out[f"{tag}_{mname}_pc_thr"] = np.asarray([x + [np.nan] * (I - len(x)) for x in pcs], dtype=np.float32)
        out[f"{tag}_{mname}_prompt_len"] = np.asarray(len(gp.prompt), dtype=np.int32)


def main():
    tags = [a for a in sys.argv[1:] if a in SHAPES] or list(SHAPES)
    out = dict(np.load(GOLD)) if os.path.exists(GOLD) else {}
    for tag in tags:
        mint(tag, out)
        np.savez_compressed(GOLD, **out)
        print(f"wrote {GOLD}: {sorted(k for k in out if k.startswith(tag))}", flush=True)


if __name__ == "__main__":
    main()
