"""Round 6: decode GEMMs of one decoder layer at 352 / 176 / 32 rows, microseconds per launch (wm_profile_kernel), for whichever library WM_LIB
points at (the product, or an arm of tests/microbench/r06_build_upper_bounds.sh).
    WM_LIB=... WM_ABI_ANY=1 python tests/microbench/r06_gemm_time.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "whisper-medusa_amd"), ROOT):
    sys.path.insert(0, p)
import torch  # noqa: E402
from whisper_medusa import MedusaConfig, WhisperMedusaModel, synth, weights  # noqa: E402

NAMES = {0: "layer(6)", 1: "LN1+QKV", 2: "out-proj", 3: "LN2+cross-q", 4: "cross-out", 5: "LN3+FC1", 6: "FC2", 7: "vocab"}
dev = torch.device("cuda", 0)
cfg = MedusaConfig.large_v2("base_head", K=10)
sd = synth.synth_state_dict(cfg, seed=0, device=str(dev), logit_std=4.5)
from whisper_medusa.engine import default_act_fp16  # noqa: E402
f16 = default_act_fp16()                # WM_ACT=f16: the fp16 single-plane decode contract (libwm_f16.so)
blob, offs = weights.build_blob(cfg, sd, device=dev, act_fp16=f16)
del sd
eng = WhisperMedusaModel.from_blob(cfg, blob, offs, max_batch=32, act_fp16=f16).engine
tag = os.path.basename(os.environ.get("WM_LIB", "libwm_f16.so" if f16 else "libwm.so"))
for rows in (352, 176, 32):
    out = []
    for kern in (1, 2, 3, 4, 5, 6, 0):
        eng.profile_layer_gemms(rows, 3, kern)
        out.append(f"{NAMES[kern]} {eng.profile_layer_gemms(rows, 40, kern)[0] * 1e3:.2f}")
    print(f"[{tag}] rows={rows}: " + " | ".join(out), flush=True)
