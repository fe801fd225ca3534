"""round 6 debugging aid: is the self-K/V cache a two-stream decode run leaves behind bit-identical to the single-stream runs'?  Probed through
wm_forward_logits at position P (one stream at a time: the probe itself is batch-free).  argv: heads [K]; toggles come from the shell environment."""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "whisper-medusa_amd"), ROOT, os.path.join(ROOT, "tests")]
from helpers import MedusaConfig, synth, clip_for          # noqa: E402
from whisper_medusa import WhisperMedusaModel              # noqa: E402

dev = torch.device("cuda", 0)
heads = sys.argv[1] if len(sys.argv) > 1 else "medusa_block"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 1
cfg = MedusaConfig.micro(K=K, heads_type=heads, n_tgt=96)
sd = synth.synth_state_dict(cfg, seed=41)
model = WhisperMedusaModel(cfg, sd, device=dev, max_batch=2)
eng = model.engine
feats = model.extract_features([clip_for(cfg, 3), clip_for(cfg, 4)[: cfg.n_mel_frames * 80]])
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("WM_"))
for plen2 in (23, 17, 15, 7, 3):
    pid2 = torch.tensor([cfg.vocab_size - 5] + [10 + (7 * i) % 900 for i in range(plen2 - 1)])
    gp2 = model._gen_params(None, None, (6, 1.3), 20, None, None, False, None, None, None, None, pid2)
    P2 = len(gp2.prompt)
    eng.encode(feats)
    enc = eng.encoder_output(2)
    eng.set_encoder_output(enc)
    eng.decode(gp2, 2, max_iters=iters)
    t2 = [eng.tokens(b) for b in range(2)]
    probe2 = eng.forward_logits([[5], [5]], P2, False)
    out = []
    for b in range(2):
        eng.set_encoder_output(enc[b: b + 1])
        eng.decode(gp2, 1, max_iters=iters)
        t1 = eng.tokens(0)
        probe1 = eng.forward_logits([[5]], P2, False)[:, 0]
        d = (probe2[:, b] - probe1).abs()
        out.append(f"b{b} tok {'==' if t1 == t2[b] else '!='} max|d| {float(d.max()):.2e} n!=0 {int((d > 0).sum())}")
    print(f"[{tag}] {heads} K={K} P={P2} rows first pass {2 * (P2 if P2 <= 16 else P2 - 16)}: " + "; ".join(out), flush=True)
