"""Round 6: what the sibling rows cost.  One stream, large-v2 + Medusa-Linear K = 10 (bench checkpoint), accept length FORCED (no hits under force_accept):
ms per iteration at a = 1 (verify pass only) and a = 0 (verify + base), for the library / WM_SIBLINGS setting of this process.
    WM_SIBLINGS=5|0 python tests/microbench/r06_sib_cost.py"""
import copy
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "whisper-medusa_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from whisper_medusa import MedusaConfig, WhisperMedusaModel, ACCEPT_TYPICAL, synth, weights  # noqa: E402

dev = torch.device("cuda", 0)
cfg = MedusaConfig.large_v2("base_head", K=10)
sd = synth.synth_state_dict(cfg, seed=0, device=str(dev), logit_std=4.5)
blob, offs = weights.build_blob(cfg, sd, device=dev, act_fp16=True)
del sd
model = WhisperMedusaModel.from_blob(cfg, blob, offs, max_batch=1, act_fp16=True)
eng = model.engine
wav = torch.from_numpy(np.stack([synth.synth_clip(500, cfg.n_mel_frames * 160)])).to(dev)
eng.encode(eng.logmel(wav))
gp = synth.bench_gen_params(cfg, max_new_tokens=128, accept_mode=ACCEPT_TYPICAL)
out = []
for a in (1, 0, 3):
    g = copy.copy(gp); g.force_accept = a
    eng.decode(g, 1); eng.decode(g, 1)
    st = eng.stats()
    out.append(f"a={a}: {st['ms_decode'] / max(st['iterations'], 1):.4f} ms/iter ({st['iterations']} it)")
eng.decode(gp, 1); eng.decode(gp, 1)
st = eng.stats()
print(f"WM_SIBLINGS={os.environ.get('WM_SIBLINGS', '5')} sibling_rows={eng.sibling_rows}: " + " | ".join(out) +
      f" | free run {st['ms_decode'] / max(st['iterations'], 1):.4f} ms/iter, hits {st['sibling_hits']} of {st['accept_hist'][0]} a=0 over {st['iterations']} it", flush=True)
