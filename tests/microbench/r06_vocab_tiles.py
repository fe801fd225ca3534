"""Round 6: tile shapes of the batched vocabulary projection (k_tile_gemm<F, TT>) under the fp16 single-plane operand, us per launch at 352 / 224 / 128 rows.
WM_TILE_F / WM_TILE_TT are read per launch."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "whisper-medusa_amd"), ROOT):
    sys.path.insert(0, p)
import torch  # noqa: E402
from whisper_medusa import MedusaConfig, WhisperMedusaModel, synth, weights  # noqa: E402

dev = torch.device("cuda", 0)
cfg = MedusaConfig.large_v2("base_head", K=10)
sd = synth.synth_state_dict(cfg, seed=0, device=str(dev), logit_std=4.5)
blob, offs = weights.build_blob(cfg, sd, device=dev, act_fp16=True)
del sd
eng = WhisperMedusaModel.from_blob(cfg, blob, offs, max_batch=32, act_fp16=True).engine
for rows in (352, 224, 128):
    out = []
    for F, TT in ((0, 0), (1, 4), (1, 8), (2, 4), (2, 8)):
        if F: os.environ["WM_TILE_F"] = str(F); os.environ["WM_TILE_TT"] = str(TT)
        else: os.environ.pop("WM_TILE_F", None); os.environ.pop("WM_TILE_TT", None)
        try:
            eng.profile_layer_gemms(rows, 3, 7)
            out.append(f"F{F}TT{TT} {eng.profile_layer_gemms(rows, 40, 7)[0] * 1e3:.2f}")
        except Exception as e:  # noqa: BLE001
            out.append(f"F{F}TT{TT} {e!r}"[:60])
    print(f"rows={rows} vocab projection us: " + " | ".join(out), flush=True)
