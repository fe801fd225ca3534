"""Round 5: what would a LayerNorm-free prologue be worth at one stream?  The three LayerNorm-fed GEMMs of a decoder layer at 11 rows (verify pass) and 1 row
(base pass), with the LayerNorm-fused loader (LdNorm: parameters -> LDS, statistics -> LDS -> barrier -> normalise) and with the identity loader (LdIdent:
fp32 rows -> hi / lo split, nothing else) in its place — the upper bound for any scheme that moves the statistics off the critical path.
Needs the variant library of this experiment (WM_LIB=libwm_nonorm.so: wm_profile_kernel honours WM_EXP_NONORM).  Results of the identity arm are numerically
meaningless; only the time counts."""
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
blob, offs = weights.build_blob(cfg, sd, device=dev)
del sd
eng = WhisperMedusaModel.from_blob(cfg, blob, offs, max_batch=1).engine
for rows in (11, 1):
    for tag, on in (("LdNorm", False), ("LdIdent", True)):
        if on:
            os.environ["WM_EXP_NONORM"] = "1"
        else:
            os.environ.pop("WM_EXP_NONORM", None)
        out = []
        for kern in (1, 3, 5, 2, 6, 0):
            eng.profile_layer_gemms(rows, 20, kern)
            out.append(f"{NAMES[kern]} {eng.profile_layer_gemms(rows, 200, kern)[0] * 1e3:.2f}")
        print(f"[{tag}] rows={rows}: " + " | ".join(out), flush=True)
