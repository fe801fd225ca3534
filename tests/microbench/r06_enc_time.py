"""Round 6: the 32-clip encoder pass (large-v2) on whatever library WM_LIB_F16 / WM_LIB points at: ms per pass (median and best of N),
TFLOP/s, and a sha256 of the encoder output (schedule variants of the GEMM K loop must leave it bit for bit).

    WM_LIB_F16=.../libwm_gs2.so python tests/microbench/r06_enc_time.py [--clips 32] [--reps 7] [--fp8]
"""
import argparse
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "whisper-medusa_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=32)
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--fp8", action="store_true")
    args = ap.parse_args()
    from whisper_medusa import MedusaConfig, WhisperMedusaModel, synth, weights
    import bench
    dev = torch.device("cuda", 0)
    cfg = MedusaConfig.large_v2("base_head", K=10)
    sd = synth.synth_state_dict(cfg, seed=0, device=str(dev), logit_std=4.5)
    blob, offs = weights.build_blob(cfg, sd, device=dev, dec_fp8=args.fp8, enc_fp8=args.fp8, act_fp16=True)
    del sd
    B = args.clips
    model = WhisperMedusaModel.from_blob(cfg, blob, offs, max_batch=B, dec_weight_fp8=args.fp8, enc_fp8=args.fp8, act_fp16=True, cross_kv_fp8=args.fp8)
    eng = model.engine
    n_samp = cfg.n_mel_frames * 160
    wav = torch.from_numpy(np.stack([synth.synth_clip(500 + j, n_samp) for j in range(B)])).to(dev)
    feats = eng.logmel(wav)
    eng.encode(feats); eng.encode(feats)
    torch.cuda.synchronize()
    ts = []
    for _ in range(args.reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.encode(feats)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    enc = eng.encoder_output(B)
    sha = hashlib.sha256(enc.float().cpu().numpy().tobytes()).hexdigest()[:16]
    fl = bench.prefill_flops(cfg) * B
    print(f"lib={os.path.basename(os.environ.get('WM_LIB_F16', 'libwm_f16.so'))} clips={B} fp8={args.fp8} ms_median={ts[len(ts) // 2]:.3f} ms_best={ts[0]:.3f} "
          f"tflops_median={fl / (ts[len(ts) // 2] * 1e-3) / 1e12:.1f} tflops_best={fl / (ts[0] * 1e-3) / 1e12:.1f} enc_sha={sha}", flush=True)


if __name__ == "__main__":
    main()
