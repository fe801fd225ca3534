"""Round 6: is the encoder pass bit-stable when two contexts run it at the same time (the micro-batch pool's situation)?
Two engines (6 clips each, one shared blob) encode concurrently from two host threads, N times; the sha256 of each context's encoder output must
not move.  Also the decode that follows (merged-step schedule, 6 streams): token lists must not move.
    WM_LIB_F16=... python tests/microbench/r06_enc_concurrent.py [--reps 40]"""
import argparse
import hashlib
import os
import sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "whisper-medusa_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from whisper_medusa import MedusaConfig, ACCEPT_TYPICAL, synth, weights  # noqa: E402
from whisper_medusa.engine import Engine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=40)
ap.add_argument("--per-ctx", type=int, default=6)
args = ap.parse_args()
dev = torch.device("cuda", 0)
cfg = MedusaConfig.large_v2("base_head", K=10)
sd = synth.synth_state_dict(cfg, seed=0, device="cpu", logit_std=4.5)
blob, offs = weights.build_blob(cfg, {k: v.to(dev) for k, v in sd.items()}, device=dev, act_fp16=True)
del sd
engs = [Engine(cfg, blob, offs, max_batch=args.per_ctx, device=dev, act_fp16=True) for _ in range(2)]
n = cfg.n_mel_frames * 160
wav = np.stack([synth.synth_clip(20 + i, n) for i in range(2 * args.per_ctx)])
wav[3, n // 2:] = 0.0
feats = engs[0].logmel(torch.from_numpy(wav[: args.per_ctx]).to(dev)), engs[1].logmel(torch.from_numpy(wav[args.per_ctx:]).to(dev))
torch.cuda.synchronize()
gp = synth.bench_gen_params(cfg, max_new_tokens=40, accept_mode=ACCEPT_TYPICAL)


def one(i):
    with torch.cuda.device(dev):
        engs[i].encode(feats[i])
        h = hashlib.sha256(engs[i].encoder_output(args.per_ctx).float().cpu().numpy().tobytes()).hexdigest()[:12]
        seqs = engs[i].decode(gp, args.per_ctx)
        return h, hashlib.sha256(repr(seqs).encode()).hexdigest()[:12]


ex = ThreadPoolExecutor(max_workers=2)
seen = [set(), set()]
tok = [set(), set()]
for r in range(args.reps):
    res = [f.result() for f in [ex.submit(one, 0), ex.submit(one, 1)]]
    for i in range(2):
        seen[i].add(res[i][0]); tok[i].add(res[i][1])
print(f"lib={os.path.basename(os.environ.get('WM_LIB_F16', 'libwm_f16.so'))} reps={args.reps} per_ctx={args.per_ctx}: encoder shas ctx0 {sorted(seen[0])} ctx1 {sorted(seen[1])}; "
      f"token-list shas ctx0 {sorted(tok[0])} ctx1 {sorted(tok[1])}", flush=True)
