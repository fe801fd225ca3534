"""Localise the micro10block stream-1 divergence: engine vs oracle, iteration by iteration, then the verify-pass logits of the
first differing iteration recomputed through wm_forward_logits on the engine's own KV cache."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "whisper-medusa_amd"), ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from helpers import MedusaConfig, synth, golden_gen_params, clip_for, ACCEPT_TYPICAL
from oracle.whisper_medusa_oracle import Oracle, log_mel, process_logits
from whisper_medusa import WhisperMedusaModel

dev = torch.device("cuda", 0)
cfg = MedusaConfig.micro(K=10, heads_type="medusa_block", d_model=256, layers=3)
sd = synth.synth_state_dict(cfg, seed=14)
orc = Oracle(cfg, sd, sim="bf16")
model = WhisperMedusaModel(cfg, sd, device=dev, max_batch=2)
eng = model.engine
wavs = [clip_for(cfg, i) for i in range(2)]
wavs[-1] = wavs[-1][: len(wavs[-1]) // 3]
n = cfg.n_mel_frames * 160
feats = np.stack([log_mel(w, cfg.num_mel_bins, n) for w in wavs])
f1 = torch.from_numpy(feats[1:2]).to(dev)
gp = golden_gen_params(cfg, ACCEPT_TYPICAL, 40, suppress_eos=True)
eng.encode(f1)
enc = eng.encoder_output(1)[0]
ref = orc.decode(enc, gp, trace=True)
for env in ({}, {"WM_NO_CARRY": "1"}, {"WM_NO_GRAPH": "1"}):
    for k, v in env.items(): os.environ[k] = v
    eng.encode(f1)
    got = eng.decode(gp, 1)[0]
    for k in env: del os.environ[k]
    fd = next((i for i, (a, b) in enumerate(zip(got, ref.ids)) if a != b), min(len(got), len(ref.ids)))
    print(env, "first divergence at", fd, "of", len(ref.ids), "engine", got[fd - 2: fd + 4], "oracle", ref.ids[fd - 2: fd + 4], "accepts", eng.stats()["accept_hist"])
P = len(gp.prompt)
pos, it_bad = P, None
eng.encode(f1)
got = eng.decode(gp, 1)[0]
fd = next((i for i, (a, b) in enumerate(zip(got, ref.ids)) if a != b), None)
if fd is None:
    print("no divergence"); sys.exit(0)
for i, t in enumerate(ref.trace):
    if pos + len(t["emit"]) > fd:
        it_bad = i; break
    pos += len(t["emit"])
t = ref.trace[it_bad]
L = t["L"]
print("diverging oracle iteration", it_bad, "L", L, "a", t["a"], "emit", t["emit"], "cand", t["cand"].tolist())
# engine state after it_bad iterations, then the verify pass on the oracle's candidates
eng.encode(f1)
eng.decode(gp, 1, max_iters=it_bad)
toks = eng.tokens(0)
print("engine tokens after", it_bad, "iterations:", toks[P:], " oracle:", ref.ids[P:L])
z = eng.forward_logits([t["cand"].tolist()], L, True)[0, 0]          # [K+1, V]
zp = process_logits(z.clone(), L, gp)
v = t["v"]
fin = torch.isfinite(v) & torch.isfinite(zp)
d = (zp - v).abs()[fin]
print("verify logits vs oracle: max|d|", float(d.max()), "mean|d|", float(d.mean()), "oracle scale", float(v[fin].abs().max()))
for r in range(z.shape[0]):
    fr = torch.isfinite(v[r]) & torch.isfinite(zp[r])
    print("  row", r, "max|d|", float((zp[r] - v[r]).abs()[fr].max()), "argmax eng/orc", int(zp[r].argmax()), int(v[r].argmax()))
# base pass of that iteration as well (K+1 heads on the last token)
st = orc.new_state(enc)
ids = ref.ids[:L]
zb = eng.forward_logits([ids[L - 1: L]], L - 1, False)[:, 0, 0]       # base pass over the last token at position L-1
print("(base-pass logits need the engine cache of the run; compare candidates instead) engine cand via argmax:", process_logits(zb, L, gp).argmax(-1).tolist())
