"""Throughput of N engine contexts (own HIP stream each, shared weights) driven from N host threads:
python tests/microbench/two_ctx.py <n_ctx> <streams_per_ctx> [steps]"""
import os, sys, threading, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "whisper-medusa_amd"))
from whisper_medusa import synth, weights
from whisper_medusa.api import WhisperMedusaModel
from whisper_medusa.config import MedusaConfig, ACCEPT_TYPICAL

nctx, B, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = torch.device("cuda", 0)
cfg = MedusaConfig.large_v2("base_head", K=10)
sd = synth.synth_state_dict(cfg, seed=0, device=str(dev), logit_std=4.5)
blob, offs = weights.build_blob(cfg, sd, device=dev)
del sd
models = [WhisperMedusaModel.from_blob(cfg, blob, offs, max_batch=B) for _ in range(nctx)]
gp = synth.bench_gen_params(cfg, max_new_tokens=128, accept_mode=ACCEPT_TYPICAL)
n_samp = cfg.n_mel_frames * 160
wavs = [torch.from_numpy(np.stack([synth.synth_clip(i * B + j, n_samp) for j in range(B)])).to(dev) for i in range(nctx)]
toks = [0] * nctx

def work(i, n):
    eng = models[i].engine
    for _ in range(n):
        eng.encode(eng.logmel(wavs[i]))
        seqs = eng.decode(gp, B)
        toks[i] += sum(len(s) - len(gp.prompt) for s in seqs)

for i in range(nctx):
    work(i, 1)                      # warm-up + graph capture, one context at a time
torch.cuda.synchronize()
toks = [0] * nctx
t0 = time.perf_counter()
th = [threading.Thread(target=work, args=(i, steps)) for i in range(nctx)]
[t.start() for t in th]; [t.join() for t in th]
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"contexts {nctx} x {B} streams: {sum(toks)} tokens in {dt * 1e3:.1f} ms -> {sum(toks) / dt:.1f} tok/s (log-mel + encode + decode)")
