"""Round 3 micro-benchmark: the decode GEMMs of one decoder layer at B-stream row counts (old register-blocked kernel vs the
LDS-ring tile kernel in every tile shape), and the encoder pass at 1 / 32 clips (round-2 two-stage 256 x 256 kernel vs the
pipelined one, ring depths).  One process, one model: the knobs are environment variables read per launch.

    python tests/microbench/r03_sweep.py [--dec] [--enc] [--out gpurun_out/r03_sweep.json]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "whisper-medusa_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

NAMES = {0: "layer(6)", 1: "LN1+QKV", 2: "out-proj", 3: "LN2+cross-q", 4: "cross-out", 5: "LN3+FC1", 6: "FC2", 7: "vocab"}


def setenv(**kw):
    for k, v in kw.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dec", action="store_true")
    ap.add_argument("--enc", action="store_true")
    ap.add_argument("--streams", type=int, default=32)
    ap.add_argument("--tile-shapes", action="store_true", help="sweep every (F, TT) shape of the tile kernel at the full row count")
    ap.add_argument("--enc-only-batch", action="store_true", help="encoder legs at the full batch only")
    ap.add_argument("--enc-one-clip", action="store_true", help="encoder legs at one clip only (process-wide knobs come from the environment)")
    ap.add_argument("--enc-knobs", action="store_true", help="encoder legs: result-preserving knobs of the pipelined GEMM")
    ap.add_argument("--enc-now", action="store_true", help="encoder legs: the shipped configuration, and the V tiles feature-major (one launch, DPP transpose)")
    ap.add_argument("--enc-r3b", action="store_true", help="encoder legs: residual GEMMs with the classic epilogue; flash-attention variants")
    ap.add_argument("--enc-order", action="store_true", help="encoder legs: strip-major tile order against strips of one column (the old 32 x 1 patches)")
    ap.add_argument("--enc-default", action="store_true", help="encoder legs: the shipped configuration only (profiling runs)")
    ap.add_argument("--enc-stagger", action="store_true", help="encoder legs: every other block of the pipelined GEMM starts late (epilogues out of phase)")
    ap.add_argument("--enc-dbg", action="store_true", help="encoder legs with parts of the pipelined GEMM switched off (WM_ENC_GEMM_DBG)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r03_sweep.json"))
    args = ap.parse_args()
    if not (args.dec or args.enc):
        args.dec = args.enc = True
    from whisper_medusa import MedusaConfig, WhisperMedusaModel, synth, weights
    dev = torch.device("cuda", 0)
    cfg = MedusaConfig.large_v2("base_head", K=10)
    sd = synth.synth_state_dict(cfg, seed=0, device=str(dev), logit_std=4.5)
    blob, offs = weights.build_blob(cfg, sd, device=dev)
    del sd
    B = args.streams
    model = WhisperMedusaModel.from_blob(cfg, blob, offs, max_batch=B)
    eng = model.engine
    res = {"decode_gemms_us": {}, "encoder_ms": {}}

    if args.dec:
        for rows in (B * 11, 88, 32, 22):
            if rows > 16 * B:
                continue
            if rows <= 32:                       # two token tiles: register-blocked kernel vs the weight-streaming kernel with a second tile
                table = {}
                for kern in (1, 2, 3, 4, 5, 6, 7, 0):
                    row = {}
                    setenv(WM_SKINNY2=0)
                    row["rows_gemm"] = round(eng.profile_layer_gemms(rows, 30, kern)[0] * 1e3, 2)
                    setenv(WM_SKINNY2=None)
                    row["skinny2"] = round(eng.profile_layer_gemms(rows, 30, kern)[0] * 1e3, 2)
                    table[NAMES[kern]] = row
                    print(f"rows={rows} {NAMES[kern]:12s} {row}", flush=True)
                res["decode_gemms_us"][str(rows)] = table
                continue
            table = {}
            for kern in (1, 2, 3, 4, 5, 6, 7, 0):
                row = {}
                setenv(WM_TILE_GEMM_MIN_MT=0, WM_TILE_F=None, WM_TILE_TT=None, WM_LN_PREFETCH=None)
                row["old"] = round(eng.profile_layer_gemms(rows, 30, kern)[0] * 1e3, 2)
                setenv(WM_TILE_GEMM_MIN_MT=None, WM_TILE_GEMM_MIN_N16=0)
                row["tile"] = round(eng.profile_layer_gemms(rows, 30, kern)[0] * 1e3, 2)
                if kern in (1, 3, 5, 0):
                    setenv(WM_LN_PREFETCH=0)
                    row["tile_nopf"] = round(eng.profile_layer_gemms(rows, 30, kern)[0] * 1e3, 2)
                    setenv(WM_LN_PREFETCH=None)
                if args.tile_shapes and rows == B * 11 and kern != 0:
                    for F in (1, 2):
                        for TT in (2, 4, 6, 8):
                            setenv(WM_TILE_F=F, WM_TILE_TT=TT)
                            try:
                                row[f"F{F}TT{TT}"] = round(eng.profile_layer_gemms(rows, 20, kern)[0] * 1e3, 2)
                            except Exception as e:  # noqa: BLE001
                                row[f"F{F}TT{TT}"] = repr(e)
                    setenv(WM_TILE_F=None, WM_TILE_TT=None)
                setenv(WM_TILE_GEMM_MIN_N16=None)
                table[NAMES[kern]] = row
                print(f"rows={rows} {NAMES[kern]:12s} {row}", flush=True)
            res["decode_gemms_us"][str(rows)] = table

    if args.enc:
        n_samp = cfg.n_mel_frames * 160
        for nb in (B, 1):
            wav = torch.from_numpy(np.stack([synth.synth_clip(900 + j, n_samp) for j in range(nb)])).to(dev)
            feats = eng.logmel(wav)
            if nb == 1 and args.enc_only_batch:
                continue
            if nb > 1 and args.enc_one_clip:
                continue
            variants = [("r02_256", dict(WM_ENC_GEMM_256P=0)), ("p_tile_per_block", dict(WM_ENC_GEMM_256P=1, WM_ENC_GEMM_PERSIST=0)),
                        ("p_persist_ring4", dict(WM_ENC_GEMM_256P=1, WM_ENC_GEMM_PERSIST=1, WM_ENC_GEMM_RING=4)),
                        ("p_persist_ring5", dict(WM_ENC_GEMM_256P=1, WM_ENC_GEMM_PERSIST=1, WM_ENC_GEMM_RING=5))] if nb > 1 else \
                       [("stages2", dict(WM_ENC_GEMM_STAGES=2)), ("stages3", dict(WM_ENC_GEMM_STAGES=3))]
            if args.enc_knobs and nb > 1:
                variants = [("default", dict(WM_ENC_GEMM_DBG=0)), ("ring_fill_after_epilogue", dict(WM_ENC_GEMM_DBG=32)), ("setprio", dict(WM_ENC_GEMM_DBG=16)),
                            ("setprio_ring5", dict(WM_ENC_GEMM_DBG=16, WM_ENC_GEMM_RING=5)), ("one_tile_per_block", dict(WM_ENC_GEMM_PERSIST=0)),
                            ("no_epilogue", dict(WM_ENC_GEMM_DBG=4)),
                            ("wn2_two_blocks_per_cu", dict(WM_ENC_GEMM_WN=2)), ("wn2_one_tile_per_block", dict(WM_ENC_GEMM_WN=2, WM_ENC_GEMM_PERSIST=0)),
                            ("wn2_no_epilogue", dict(WM_ENC_GEMM_WN=2, WM_ENC_GEMM_DBG=4))]
            if args.enc_now:
                variants = [("default", dict(WM_ENC_GEMM_DBG=0)), ("v_tiles_feature_major", dict(WM_ENC_GEMM_DBG=64))]
            if args.enc_r3b:
                variants = [("default", dict(WM_ENC_GEMM_DBG=0)), ("residual_classic_epilogue", dict(WM_ENC_GEMM_DBG=128)),
                            ("flash_groups_of_4", dict(WM_FLASH_VARIANT=1)), ("flash_3_blocks_per_cu", dict(WM_FLASH_VARIANT=2)),
                            ("flash_3_blocks_per_cu_groups_of_1", dict(WM_FLASH_VARIANT=3))]
            if args.enc_order:
                variants = [("default", dict(WM_ENC_GEMM_PN=0)), ("strips_of_1_like_32x1_patches", dict(WM_ENC_GEMM_PN=1)), ("flash_groups_of_4", dict(WM_FLASH_VARIANT=1))]
            if args.enc_default:
                variants = [("default", dict(WM_ENC_GEMM_DBG=0))]
            if args.enc_stagger:
                variants = [("default", dict(WM_ENC_GEMM_DBG=0))] + [(f"odd_blocks_late_{n * 3.4:.0f}us", dict(WM_ENC_GEMM_DBG=n << 8)) for n in (3, 6, 10)]
            if args.enc_dbg and nb > 1:
                variants = [("full", dict(WM_ENC_GEMM_DBG=0)), ("no_mfma", dict(WM_ENC_GEMM_DBG=1)), ("no_refill", dict(WM_ENC_GEMM_DBG=2)),
                            ("no_epilogue", dict(WM_ENC_GEMM_DBG=4)), ("no_frag_reads", dict(WM_ENC_GEMM_DBG=8)),
                            ("mfma_only", dict(WM_ENC_GEMM_DBG=14)), ("refill_only", dict(WM_ENC_GEMM_DBG=13)), ("frag_reads_only", dict(WM_ENC_GEMM_DBG=7)),
                            ("barriers_only", dict(WM_ENC_GEMM_DBG=15))]
            row = {}
            for name, env in variants:
                setenv(**env)
                eng.encode(feats)
                ts = []
                for _ in range(3):
                    eng.encode(feats)
                    ts.append(eng.stats()["ms_encode"])
                row[name] = round(min(ts), 3)
                for k in env:
                    setenv(**{k: None})
            flops = 2.587e12 * nb
            row["tflops"] = {k: round(flops / (v * 1e-3) / 1e12, 1) for k, v in row.items()}
            print(f"encoder {nb} clip(s): {row}", flush=True)
            res["encoder_ms"][str(nb)] = row

    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
