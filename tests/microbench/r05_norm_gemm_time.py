"""Round 5: the three LayerNorm-fed decode GEMMs of one decoder layer at 352 / 208 / 96 rows, microseconds per launch group (wm_profile_kernel):
LayerNorm launch + k_rows_gemm (WM_ROWS_NORM=0) against the LayerNorm in the GEMM's own prologue, k_rows_norm_gemm (WM_ROWS_NORM=1).
    python tests/microbench/r05_norm_gemm_time.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "whisper-medusa_amd"), ROOT):
    sys.path.insert(0, p)
import torch  # noqa: E402
from whisper_medusa import MedusaConfig, WhisperMedusaModel, synth, weights  # noqa: E402

NAMES = {0: "layer(6)", 1: "LN1+QKV", 2: "out-proj", 3: "LN2+cross-q", 4: "cross-out", 5: "LN3+FC1", 6: "FC2"}
dev = torch.device("cuda", 0)
cfg = MedusaConfig.large_v2("base_head", K=10)
sd = synth.synth_state_dict(cfg, seed=0, device=str(dev), logit_std=4.5)
fp8 = "--fp8" in sys.argv
blob, offs = weights.build_blob(cfg, sd, device=dev, dec_fp8=fp8)
del sd
model = WhisperMedusaModel.from_blob(cfg, blob, offs, max_batch=32, dec_weight_fp8=fp8)
eng = model.engine
for rows in (352, 208, 96):
    for tag, val in (("LN launch + k_rows_gemm", "0"), ("k_rows_norm_gemm", "1")):
        os.environ["WM_ROWS_NORM"] = val
        out = []
        for kern in (1, 3, 5, 0):
            eng.profile_layer_gemms(rows, 3, kern)
            out.append(f"{NAMES[kern]} {eng.profile_layer_gemms(rows, 30, kern)[0] * 1e3:.2f}")
        print(f"[{tag}{' fp8' if fp8 else ''}] rows={rows}: " + " | ".join(out), flush=True)
