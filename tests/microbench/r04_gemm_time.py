"""Round 4: decode GEMMs of one decoder layer at 352 / 32 rows, microseconds per launch group (wm_profile_kernel), for the library WM_LIB points at.
    WM_LIB=.../libwm_occ5.so python tests/microbench/r04_gemm_time.py [tag]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "whisper-medusa_amd"), ROOT):
    sys.path.insert(0, p)
import torch  # noqa: E402
from whisper_medusa import MedusaConfig, WhisperMedusaModel, synth, weights  # noqa: E402

NAMES = {0: "layer(6)", 1: "LN1+QKV", 2: "out-proj", 3: "LN2+cross-q", 4: "cross-out", 5: "LN3+FC1", 6: "FC2", 7: "vocab"}
tag = sys.argv[1] if len(sys.argv) > 1 else "base"
dev = torch.device("cuda", 0)
cfg = MedusaConfig.large_v2("base_head", K=10)
sd = synth.synth_state_dict(cfg, seed=0, device=str(dev), logit_std=4.5)
blob, offs = weights.build_blob(cfg, sd, device=dev)
del sd
model = WhisperMedusaModel.from_blob(cfg, blob, offs, max_batch=32)
eng = model.engine
for rows in (352, 32):
    out = []
    for kern in (1, 2, 3, 4, 5, 6, 0):
        eng.profile_layer_gemms(rows, 5, kern)
        out.append(f"{NAMES[kern]} {eng.profile_layer_gemms(rows, 40, kern)[0] * 1e3:.2f}")
    print(f"[{tag}] rows={rows}: " + " | ".join(out), flush=True)
