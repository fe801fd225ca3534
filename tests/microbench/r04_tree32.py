"""Round 4: cost of a candidate TREE at 32 streams against the chain, large-v2 + Medusa-Linear K = 10.
Decode-only ms per iteration with the accept length of every iteration forced (wm.h force_accept: the tokens are then meaningless, the
cost is not), and with the random-init heads' own acceptance.  python tests/microbench/r04_tree32.py"""
import copy
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "whisper-medusa_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from whisper_medusa import MedusaConfig, WhisperMedusaModel, synth, weights, ACCEPT_TYPICAL  # noqa: E402

dev = torch.device("cuda", 0)
B, NEW = 32, 64
n = None
for name, choices in (("chain [1]*11 (11 rows / stream)", None), ("tree [1,2,2,1x8] (39 nodes, 4 paths)", [1, 2, 2] + [1] * 8),
                      ("tree [1,3,2,1x8] (46 nodes, 6 paths)", [1, 3, 2] + [1] * 8)):
    cfg = MedusaConfig.large_v2("base_head", K=10) if choices is None else MedusaConfig.large_v2("base_head", K=10, medusa_choices=choices)
    sd = synth.synth_state_dict(cfg, seed=0, device=str(dev), logit_std=4.5)
    blob, offs = weights.build_blob(cfg, sd, device=dev)
    del sd
    model = WhisperMedusaModel.from_blob(cfg, blob, offs, max_batch=B)
    eng = model.engine
    n = cfg.n_mel_frames * 160
    wav = torch.from_numpy(np.stack([synth.synth_clip(500 + j, n) for j in range(B)])).to(dev)
    eng.encode(eng.logmel(wav))
    gp = synth.bench_gen_params(cfg, max_new_tokens=NEW, accept_mode=ACCEPT_TYPICAL)
    eng.decode(gp, B)
    eng.decode(gp, B)
    st = eng.stats()
    row = [f"own acceptance: {st['ms_decode'] / max(st['iterations'], 1):.3f} ms/it, {st['tokens_emitted'] / max(st['iterations'], 1) / B:.2f} tok/it, "
           f"{st['tokens_emitted'] / (st['ms_decode'] * 1e-3):.0f} tok/s"]
    for a in (0, 1, 3, 10):
        g = copy.copy(gp); g.force_accept = a
        eng.decode(g, B)
        st = eng.stats()
        row.append(f"a={a}: {st['ms_decode'] / max(st['iterations'], 1):.3f} ms/it")
    print(f"{name}: " + " | ".join(row), flush=True)
    eng.close()
    del model, blob
    torch.cuda.empty_cache()
