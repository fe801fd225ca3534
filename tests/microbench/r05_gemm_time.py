"""Round 5: decode GEMMs of one decoder layer at 352 / 176 rows, microseconds per launch (wm_profile_kernel), register-blocked k_rows_gemm
against the LDS-shared token-tile kernel k_rows_lds and its block shapes (the knobs are read per launch).
    python tests/microbench/r05_gemm_time.py"""
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
fp8 = "--fp8" in sys.argv
blob, offs = weights.build_blob(cfg, sd, device=dev, dec_fp8=fp8)
del sd
model = WhisperMedusaModel.from_blob(cfg, blob, offs, max_batch=32, dec_weight_fp8=fp8)
eng = model.engine
ARMS = [("k_rows_gemm", dict(WM_ROWS_LDS="0")), ("k_rows_lds auto", dict(WM_ROWS_LDS="1"))]
for nw, ft in ((4, 1), (5, 1), (8, 1), (4, 2), (6, 2), (8, 2)):
    ARMS.append((f"k_rows_lds NW={nw} FT={ft}", dict(WM_ROWS_LDS="1", WM_RL_NW=str(nw), WM_RL_FT=str(ft))))
if "--short" in sys.argv:          # counter passes: the shipped kernel and the planned shapes only, 352 rows
    ARMS = ARMS[:2]
KEYS = ("WM_ROWS_LDS", "WM_RL_NW", "WM_RL_FT")
for rows in ((352,) if "--short" in sys.argv else (352, 176)):
    for tag, env in (ARMS if rows == 352 else ARMS[:2]):
        for k in KEYS:
            os.environ.pop(k, None)
        os.environ.update(env)
        out = []
        for kern in (1, 2, 3, 4, 5, 6, 0):
            eng.profile_layer_gemms(rows, 3, kern)
            out.append(f"{NAMES[kern]} {eng.profile_layer_gemms(rows, 30, kern)[0] * 1e3:.2f}")
        print(f"[{tag}] rows={rows}: " + " | ".join(out), flush=True)
