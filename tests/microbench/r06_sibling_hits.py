"""Round 6 (counted before building): how often would a depth-1 SIBLING row save the base pass?
With a chain of K + 1 candidates in a 16-row verify tile, rows K+1.. could carry head 1's top-2 .. top-(S+1) tokens as leaves under the root; when the
chain accepts nothing (a = 0) and argmax v_0 — the next root — is one of them, that row's hidden state is the next iteration's base state and the base
pass can be skipped.  The oracle (CPU, the checker) walks the reference loop on the synthetic checkpoint and counts, over the a = 0 iterations, how often
argmax v_0 is among head 1's top-2 .. top-(S+1).

    python tests/microbench/r06_sibling_hits.py [--model tiny|micro10|large] [--clips 4] [--max-new 64]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "whisper-medusa_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
from whisper_medusa import MedusaConfig, ACCEPT_TYPICAL, synth  # noqa: E402
from oracle.whisper_medusa_oracle import Oracle, process_logits, evaluate_posterior_chain, log_mel  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="tiny")
    ap.add_argument("--clips", type=int, default=4)
    ap.add_argument("--max-new", type=int, default=64)
    ap.add_argument("--logit-std", type=float, default=4.5)
    ap.add_argument("--siblings", type=int, default=5)
    args = ap.parse_args()
    cfg = {"tiny": lambda: MedusaConfig.tiny_en(K=10), "micro10": lambda: MedusaConfig.micro(K=10, d_model=128, layers=2),
           "large": lambda: MedusaConfig.large_v2("base_head", K=10)}[args.model]()
    sd = synth.synth_state_dict(cfg, seed=0, logit_std=args.logit_std)
    orc = Oracle(cfg, {k: v.float() for k, v in sd.items()}, sim="fp32")
    gp = synth.bench_gen_params(cfg, max_new_tokens=args.max_new, accept_mode=ACCEPT_TYPICAL)
    K = cfg.medusa_num_heads
    n_it = n_a0 = n_hit = 0
    hist = [0] * (K + 1)
    for c in range(args.clips):
        n_samp = cfg.n_mel_frames * 160
        enc = orc.encode(torch.from_numpy(log_mel(synth.synth_clip(500 + c, n_samp), cfg.num_mel_bins, n_samp)))
        st = orc.new_state(enc)
        ids = list(gp.prompt)
        while True:
            L, kv = len(ids), st["kv_len"]
            z = orc.decoder_pass(st, ids[kv:L], kv, disable_medusa=False, last_only=True)[:, 0]
            st["kv_len"] = L
            z = process_logits(z, L, gp)
            cand = torch.argmax(z, dim=-1)
            sib = torch.topk(z[1], args.siblings + 1).indices[1:].tolist()
            v = process_logits(orc.decoder_pass(st, cand.tolist(), L, disable_medusa=True)[0], L, gp)
            a, _ = evaluate_posterior_chain(v, cand, gp)
            n_it += 1; hist[a] += 1
            if a == 0:
                nxt = int(torch.argmax(v[0]))
                n_a0 += 1; n_hit += int(nxt in sib)
                ids += [int(cand[0]), nxt]; st["kv_len"] = L + 1
            else:
                ids += [int(t) for t in cand[: a + 1]]; st["kv_len"] = L + a
            if gp.eos_token_id in ids[L:] or len(ids) >= gp.max_length or len(ids) + K >= gp.hard_max_length:
                break
    print(f"model={args.model} clips={args.clips} iterations={n_it} accept_hist={hist} a=0: {n_a0} ({n_a0 / max(n_it, 1):.2f}) "
          f"sibling hits (argmax v0 in head 1's top-2..{args.siblings + 1}): {n_hit} ({n_hit / max(n_a0, 1):.3f} of the a=0 iterations)")


if __name__ == "__main__":
    main()
