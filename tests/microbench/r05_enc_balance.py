"""Round 5 micro-benchmark: encoder pass at 1-4 clips with the two tile-shape changes of the round, each switchable per launch:

  WM_ENC_BALANCE  one balanced round of taller tiles for the LayerNorm-fed GEMMs (192 rows QKV, 256 rows FC1 at one clip): bit-identical
  WM_ENC_XSPLIT   residual GEMMs with the K loop split over two blocks per 128 x 128 tile, the second split's partial folded into the
                  residual stream by the next LayerNorm launch (1: FC2, 2: out-proj too): another fp32 summation order

Needs the variant library (tests/microbench/r05_build_enc_tiles.sh, WM_LIB=.../libwm_enctiles.so): the product sources do not carry these knobs
(measured, not shipped: profiles/r05_enc_few_clip_tiles.md).  One process, one model; prints ms per encoder pass (min of 5) and the difference of the encoder output against the round-4 dispatch.

    python tests/microbench/r05_enc_balance.py [--clips 1 2 3 4] [--profile] [--out gpurun_out/r05_enc_balance.json]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "whisper-medusa_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

VARIANTS = [("round4", dict(WM_ENC_BALANCE=0, WM_ENC_XSPLIT=0)),
            ("balance", dict(WM_ENC_BALANCE=1, WM_ENC_XSPLIT=0)),
            ("xsplit_fc2", dict(WM_ENC_BALANCE=0, WM_ENC_XSPLIT=1)),
            ("balance+xsplit_fc2", dict(WM_ENC_BALANCE=1, WM_ENC_XSPLIT=1)),
            ("balance+xsplit_fc2_outproj", dict(WM_ENC_BALANCE=1, WM_ENC_XSPLIT=2))]
# weight residency (GPU call 8): every layer on layer 0's weights = the pass with cache-resident weights (ceiling of a prefetch; not the model's
# output), and the side-stream prefetch one layer ahead (plain / nt loads)
PF_VARIANTS = [("round4", dict(WM_ENC_BALANCE=0, WM_ENC_XSPLIT=0)),
               ("round4_same_layer_weights", dict(WM_ENC_BALANCE=0, WM_ENC_XSPLIT=0, WM_ENC_SAME_LAYER=1)),
               ("round4_prefetch", dict(WM_ENC_BALANCE=0, WM_ENC_XSPLIT=0, WM_ENC_PREFETCH=1)),
               ("round4_prefetch_nt", dict(WM_ENC_BALANCE=0, WM_ENC_XSPLIT=0, WM_ENC_PREFETCH=2)),
               ("xsplit_fc2_prefetch", dict(WM_ENC_BALANCE=0, WM_ENC_XSPLIT=1, WM_ENC_PREFETCH=1)),
               ("balance+xsplit_fc2_same_layer_weights", dict(WM_ENC_BALANCE=1, WM_ENC_XSPLIT=1, WM_ENC_SAME_LAYER=1))]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, nargs="+", default=[1, 2, 3, 4])
    ap.add_argument("--profile", action="store_true", help="few passes of the first and the last-but-one variant only (under rocprofv3)")
    ap.add_argument("--prefetch", action="store_true", help="the weight-residency variants instead of the tile-shape ones")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r05_enc_balance.json"))
    args = ap.parse_args()
    from whisper_medusa import MedusaConfig, WhisperMedusaModel, synth, weights
    dev = torch.device("cuda", 0)
    cfg = MedusaConfig.large_v2("base_head", K=10)
    sd = synth.synth_state_dict(cfg, seed=0, device=str(dev), logit_std=4.5)
    blob, offs = weights.build_blob(cfg, sd, device=dev)
    del sd
    model = WhisperMedusaModel.from_blob(cfg, blob, offs, max_batch=max(args.clips))
    eng = model.engine
    n_samp = cfg.n_mel_frames * 160
    res = {}
    variants = [VARIANTS[0], VARIANTS[3]] if args.profile else VARIANTS
    if args.prefetch:
        variants = [PF_VARIANTS[0], PF_VARIANTS[1], PF_VARIANTS[2]] if args.profile else PF_VARIANTS
    for nb in args.clips:
        wav = torch.from_numpy(np.stack([synth.synth_clip(900 + j, n_samp) for j in range(nb)])).to(dev)
        feats = eng.logmel(wav)
        row, ref = {}, None
        for name, env in variants:
            for k, v in env.items():
                os.environ[k] = str(v)
            eng.encode(feats)
            ts = []
            for _ in range(3 if args.profile else 5):
                eng.encode(feats)
                ts.append(eng.stats()["ms_encode"])
            out = eng.encoder_output(nb).float().cpu()
            if ref is None:
                ref = out
            diff = (out - ref).abs()
            row[name] = {"ms": round(min(ts), 4), "tflops": round(2.587e12 * nb / (min(ts) * 1e-3) / 1e12, 1),
                         "max_abs_diff_vs_round4": float(diff.max()), "mean_abs_diff": float(diff.mean()), "out_abs_max": float(ref.abs().max())}
            for k in env:
                os.environ.pop(k, None)
            print(f"encoder {nb} clip(s) {name:28s} {row[name]}", flush=True)
        res[str(nb)] = row
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
