"""bench.py — headline benchmark of the MI355X Whisper-Medusa engine (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

One STEP = one pass of the whole hot path over one batch of synthetic 30 s clips per GPU:
log-mel -> Whisper-large-v2 encoder + cross-KV projection -> Medusa-Linear-10 decode loop
(base pass + verify pass + typical acceptance, replayed from a hipGraph) to a fixed budget of
128 new tokens per stream with EOS suppressed.  Clips are resident in HBM before
the timed region.  Streams shard data-parallel over ranks (no data-path collective); weights are
packed on rank 0 and broadcast once over RCCL (outside the timed region).

Prints ONE JSON line on rank 0: whole-job decoded tokens/s (+ RTF, iterations/s, tokens/iteration,
the vanilla-greedy anchor), a `roofline` object for the decode iteration (algorithmic bytes of
SURVEY.md §8d / hipEvent-timed iteration) and a `cpu_baseline` object (the oracle on the host cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "whisper-medusa_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402


def decode_iter_bytes(cfg, B, mean_len, dec_fp8=False, parts=False):
    """Algorithmic HBM bytes of one Medusa iteration: weights of the base pass (K+1 heads)
    + weights of the verify pass (1 head) + per stream 2 x cross-KV + self-KV read in both passes.
    ``dec_fp8``: the decoder-layer matrices are 1 byte per parameter (+ 4 bytes per output row of scales)."""
    d, f, L, V, K = cfg.d_model, cfg.decoder_ffn_dim, cfg.decoder_layers, cfg.vocab_size, cfg.medusa_num_heads
    layer = 6 * d * d + 2 * d * f            # q,k,v,out, cross-q, cross-out + fc1, fc2 (cross k/v proj not re-read)
    lb = (1.0 * layer + 4.0 * (7 * d + f)) if dec_fp8 else 2.0 * layer          # bytes of one layer's matrices
    if cfg.is_block:
        w_a = (L + 1) * lb + 2.0 * (V * d + K * d * d)
        w_v = L * lb + (lb * 3 * d * d / layer) + 2.0 * V * d   # block runs only its qkv projection in the verify pass
        nkv = L + 1
    else:
        w_a = L * lb + 2.0 * (V * d + (K + 1) * d * d)
        w_v = L * lb + 2.0 * (V * d + d * d)
        nkv = L
    X = nkv * 2 * cfg.max_source_positions * d * 2
    skv = 2 * (nkv * 2 * mean_len * d * 2)
    if parts:
        # vanilla greedy step: the layers + the base head only (Linear: head 0 of the stacked heads; Block: plain proj_out)
        w_van = L * lb + 2.0 * V * d + (0.0 if cfg.is_block else 2.0 * d * d)
        return {"w_base": w_a, "w_verify": w_v, "w_vanilla": w_van, "cross_kv_per_stream_pass": float(X), "self_kv_per_stream_pass": skv / 2.0}
    return (w_a + w_v) + B * (2.0 * X + skv)


def executed_bytes(cfg, B, mean_len, dec_fp8, p_base_pass, p_base_attn):
    """Bytes the engine actually had to move per iteration given what ran: every iteration has a verify pass; a base pass (its weights)
    ran with probability `p_base_pass` (one stream: only after an accept length of 0 — otherwise the hidden state was carried and the
    host skipped the pass; several streams: always, the rows of carrying streams ride along) and a stream's base-pass attention (its
    cross- and self-K/V reads) with probability `p_base_attn`.  SURVEY §8d's figure (decode_iter_bytes) assumes both are 1."""
    p = decode_iter_bytes(cfg, B, mean_len, dec_fp8, parts=True)
    kv = p["cross_kv_per_stream_pass"] + p["self_kv_per_stream_pass"]
    return p["w_verify"] + B * kv + p_base_pass * p["w_base"] + p_base_attn * B * kv


def prefill_flops(cfg, parts=False):
    """Algorithmic FLOPs of the encoder + cross-KV projection for one 30 s clip.
    ``parts``: (flops of the GEMMs the fp8 configuration runs on the fp8 MFMA — the LayerNorm-fed QKV and FC1 of every encoder layer and
    the cross-K/V projection —, flops that stay bf16 there: out-proj, FC2, attention, convolutions)."""
    d, f, S, L = cfg.d_model, cfg.encoder_ffn_dim, cfg.max_source_positions, cfg.encoder_layers
    per_layer = 4 * 2 * S * d * d + 2 * 2 * S * S * d + 2 * 2 * S * d * f
    conv = 2 * (2 * S) * d * 3 * cfg.num_mel_bins + 2 * S * d * 3 * d
    cross = cfg.n_kv_layers * 2 * 2 * S * d * d
    if parts:
        f8 = L * (3 * 2 * S * d * d + 2 * S * d * f) + cross
        return f8, L * per_layer + conv + cross - f8
    return L * per_layer + conv + cross


MFMA_PEAK_BF16, MFMA_PEAK_FP8 = 2.5e15, 5.0e15          # dense, BASELINE.md §2 (AMD's headline figures include 2:1 sparsity)


def prefill_roofline(cfg, clips, ms, fp8):
    """Prefill priced against the matrix-core peak.  bf16: 2.5 PFLOP/s.  The fp8 configuration runs 7 / 12 of its GEMM flops on the fp8 MFMA
    (peak 5 PFLOP/s): its floor is flops_fp8 / 5e15 + flops_bf16 / 2.5e15 and `frac` is floor / measured; the same time against the bf16
    peak alone is kept as `frac_vs_bf16_peak` (the number earlier rounds reported for this leg)."""
    flops = prefill_flops(cfg) * clips
    t = ms * 1e-3
    out = {"bound": "mfma", "tflops_per_clip": round(prefill_flops(cfg) / 1e12, 3), "achieved": round(flops / t / 1e12, 1), "unit": "TFLOP/s"}
    if fp8:
        f8, bf = prefill_flops(cfg, parts=True)
        floor = clips * (f8 / MFMA_PEAK_FP8 + bf / MFMA_PEAK_BF16)
        out.update({"peak": round(flops / floor / 1e12, 1), "peak_note": "blended: fp8 share of the flops at 5 PFLOP/s, the rest at 2.5 PFLOP/s",
                    "fp8_flop_share": round(f8 / (f8 + bf), 4), "frac": round(floor / t, 4),
                    "frac_vs_bf16_peak": round(flops / t / MFMA_PEAK_BF16, 4)})
    else:
        out.update({"peak": MFMA_PEAK_BF16 / 1e12, "frac": round(flops / t / MFMA_PEAK_BF16, 4)})
    return out


DECODE_SOURCES = ("wm_common.h", "wm_decoder.hip", "wm_engine.hip", "wm_epilogues.h", "wm_internal.h", "wm_skinny_gemm.h")


def kernels_sha():
    """sha1 over the HIP sources of the DECODE path (the traffic figure is the decode iteration's): a PMC traffic file measured
    on other kernels is refused as stale.  tests/pmc_summary.py stamps the same hash into the file it writes."""
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(ROOT, "whisper-medusa_amd", "csrc")
    for f in DECODE_SOURCES:
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def median(v):
    v = sorted(v)
    return v[len(v) // 2]


def cpu_baseline_leg(cfg, sd, eng, gp, wav0, engine_ids, n_iters, fp8, full_budget, act="hilo", xkv8=None):
    """The oracle (a CPU port of the reference algorithm, oracle/) on the host cores of this box, SURVEY.md §8d: split
    log-mel / encoder / decode timers, 1 warm-up + 3 timed runs each, median.  The decode sample is `n_iters` Medusa iterations
    (the whole 128-token budget with --cpu-full).  Parity: the oracle in the engine's numeric contract (sim="bf16"), fed with
    the engine's encoder output, must emit exactly the engine's token ids for those iterations."""
    from oracle.whisper_medusa_oracle import Oracle, log_mel
    sd_cpu = {k: v.float().cpu() for k, v in sd.items()}
    xkv8 = fp8 if xkv8 is None else xkv8
    orc = Oracle(cfg, sd_cpu, sim="fp32", dec_fp8=fp8, enc_fp8=fp8, xkv_fp8=xkv8)             # the reference's default dtype
    n = cfg.n_mel_frames * 160
    wav = wav0.cpu().numpy()
    # The host thread count is calibrated, not assumed: the decode loop is a chain of small ops and skinny GEMVs, and on the GPU box's 256
    # hardware threads torch's default pool (128) and the 64 threads of earlier rounds are ~3x SLOWER than 16 (profiles/r05_oracle_threads.log).
    # The baseline is the best of {8, 16, 32, 64} threads: per count one warm-up iteration, then the median of three 2-iteration samples (a
    # single sample moved the pick between runs).  `cores` = the box's hardware threads (os.cpu_count()), `threads_used` = the pool of the sample.
    enc0 = eng.encoder_output(1)[0]
    hw = os.cpu_count() or 1
    cand = [c for c in (8, 16, 32, 64) if c <= hw] or [hw]
    calib = {}
    for c in cand:
        torch.set_num_threads(c)
        orc.decode(enc0, gp, max_iters=1)
        ts = []
        for _ in range(3):
            t = time.perf_counter(); orc.decode(enc0, gp, max_iters=2); ts.append(time.perf_counter() - t)
        calib[c] = median(ts)
    ncore = min(calib, key=calib.get)
    torch.set_num_threads(ncore)

    def timed(fn, reps=3):
        fn()                                                         # warm-up
        ts = []
        for _ in range(reps):
            t = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t)
        return median(ts), r
    t_mel, feats = timed(lambda: log_mel(wav, cfg.num_mel_bins, n))
    t_enc, _ = timed(lambda: orc.encode(torch.from_numpy(feats)), reps=3)
    enc = eng.encoder_output(1)[0]                                   # decode legs start from the GPU encoder output
    iters = None if full_budget else n_iters
    t_dec, r = timed(lambda: orc.decode(enc, gp, max_iters=iters))
    ntok = len(r.ids) - len(gp.prompt)
    # parity on the same iterations, engine contract
    orc16 = Oracle(cfg, sd_cpu, sim="bf16", dec_fp8=fp8, enc_fp8=fp8, act=act, xkv_fp8=xkv8)
    r16 = orc16.decode(enc, gp, max_iters=iters)
    ok = engine_ids[: len(r16.ids)] == r16.ids
    first = next((i for i, (a, b) in enumerate(zip(engine_ids, r16.ids)) if a != b), min(len(engine_ids), len(r16.ids)))
    agree32 = next((i for i, (a, b) in enumerate(zip(engine_ids, r.ids)) if a != b), min(len(engine_ids), len(r.ids))) - len(gp.prompt)
    audio_s = 30.0 * cfg.max_source_positions / 1500.0
    return {"value": round(ntok / t_dec, 3), "unit": "tokens/s", "cores": hw, "threads_used": ncore, "kind": "port",
            "threads_calibration_s": {str(k): round(v, 3) for k, v in calib.items()},
            "sample": f"oracle (PyTorch CPU fp32 restatement of the reference loop) on clip 0, 1 warm-up + 3 runs, median: log-mel "
                      f"{t_mel:.2f} s, encoder {t_enc:.2f} s, decode (cross-KV projection + {r.n_iters} Medusa iterations = {ntok} tokens, "
                      f"from the GPU encoder output) {t_dec:.2f} s",
            "s_logmel": round(t_mel, 3), "s_encoder": round(t_enc, 3), "s_decode_sample": round(t_dec, 3),
            "decode_sample_iters": r.n_iters, "decode_sample_tokens": ntok,
            "whole_clip_rtf_estimate": round((t_mel + t_enc + t_dec * (gp.max_length - len(gp.prompt)) / max(ntok, 1)) / audio_s, 3),
            "parity_checked": bool(ok), "parity_tokens_compared": len(r16.ids) - len(gp.prompt),
            "parity_first_divergence": None if ok else first - len(gp.prompt),
            "fp32_oracle_agrees_for_tokens": agree32}


def acceptance_sensitivity(eng, cfg, gp_base, B, max_new, fp8=False, accepts=None, vanilla_tps=None):
    """Decode-only tokens/s with the accept length of every iteration FORCED to a (wm.h force_accept): the cost side of the
    headline, independent of what the random-init heads happen to accept.  a = 0 is the floor (2 tokens, 2 passes per iteration).
    Each row also carries the HBM fraction of what that iteration executes: a = 0 runs both passes (SURVEY §8d's bytes), a >= 1 at one
    stream runs the verify pass only (the carry skips the base pass), at several streams the base pass's weights but not its attention."""
    import copy
    out = {}
    mean_len = len(gp_base.prompt) + max_new / 2
    for a in (accepts or (0, 1, 2, 3, 5, cfg.medusa_num_heads)):
        g = copy.copy(gp_base); g.force_accept = a
        eng.decode(g, B)
        st = eng.stats()
        ms_it = st["ms_decode"] / max(st["iterations"], 1)
        nbytes = executed_bytes(cfg, B, mean_len, fp8, 1.0 if (a == 0 or B > 1) else 0.0, 1.0 if a == 0 else 0.0)
        passes = "verify + base" if a == 0 else ("verify only" if B == 1 else "verify + base weights (attention of carried streams skipped)")
        spi = None
        if st.get("schedule_steps", 0) > 0:      # merged-step schedule: one pass per step, an accept length of 0 costs the stream a second step
            spi = st["schedule_steps"] / max(st["iterations"], 1)
            nbytes = spi * executed_bytes(cfg, B, mean_len, fp8, 0.0, 0.0)
            passes = f"{spi:.2f} merged steps (one pass each)"
        out[f"a={a}"] = {"tokens_per_sec": round(st["tokens_emitted"] / (st["ms_decode"] * 1e-3), 1),
                         "ms_per_iteration": round(ms_it, 4),
                         "tokens_per_iteration": round(st["tokens_emitted"] / max(st["iterations"], 1) / B, 3),
                         "passes": passes,
                         "bytes_executed": round(nbytes), "frac_hbm_executed": round(nbytes / (ms_it * 1e-3) / 8e12, 4)}
        if spi is not None:
            out[f"a={a}"]["steps_per_iteration"] = round(spi, 3)
        if vanilla_tps:        # the leg's Medusa / vanilla ratio AT this accept length (the cost side, whatever the random-init heads accept)
            out[f"a={a}"]["medusa_over_vanilla"] = round(out[f"a={a}"]["tokens_per_sec"] / vanilla_tps, 3)
    return out


def leg_parity(cfg, sd, eng, gp, fp8, iters=8, act="hilo"):
    """Stream 0 of the leg's LAST decoded batch against the oracle in the engine's numeric contract, fed with the engine's encoder
    output, for the first `iters` Medusa iterations (the checker, never the thing measured)."""
    from oracle.whisper_medusa_oracle import Oracle
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    orc = Oracle(cfg, {k: v.float().cpu() for k, v in sd.items()}, sim="bf16", dec_fp8=fp8, enc_fp8=fp8, act=act, xkv_fp8=fp8)
    enc = eng.encoder_output(1)[0]
    ref = orc.decode(enc, gp, max_iters=iters)
    got = eng.tokens(0)
    n = len(ref.ids)
    out = {"parity_checked": bool(got[:n] == ref.ids), "parity_strict": bool(got[:n] == ref.ids),
           "parity_tokens_compared": n - len(gp.prompt), "parity_iterations": ref.n_iters}
    if not out["parity_strict"]:
        # Two correct implementations of one contract still differ in fp32 summation order; behind the fp16 operand rounding that is a decision
        # flipped where the ORACLE's own margin is below that difference.  Walk the oracle along the engine's ids (tests/helpers.py:check_tokens):
        # every decision outside such a margin must match; the ones inside are listed, never hidden.
        ok, ties, _ = orc.decode_following(enc, gp, list(got), tol_logit=2e-3, max_iters=iters)
        out["parity_first_difference"] = next((i for i, (a, b) in enumerate(zip(got, ref.ids)) if a != b), n) - len(gp.prompt)
        out["parity_ties_followed"] = [{k: (float(v) if isinstance(v, (int, float)) else str(v)) for k, v in t.items()} for t in ties]
        out["parity_checked"] = bool(ok and 0 < len(ties) <= 2)
    return out


def extra_config(name, heads, B, fp8, dev, logit_std, max_new, steps=8, f16=False):
    """One more BASELINE.json config measured in the same process (B streams, one context): whole-step tokens/s, decode
    iteration time, vanilla anchor, roofline fractions, and a parity check of stream 0 against the oracle."""
    from whisper_medusa import MedusaConfig, WhisperMedusaModel, ACCEPT_TYPICAL, synth, weights
    cfg = MedusaConfig.large_v2(heads, K=10)
    sd = synth.synth_state_dict(cfg, seed=0, device=str(dev), logit_std=logit_std)
    blob, offs = weights.build_blob(cfg, sd, device=dev, dec_fp8=fp8, enc_fp8=fp8, act_fp16=f16)
    sd_cpu = {k: v.cpu() for k, v in sd.items()}
    del sd
    model = WhisperMedusaModel.from_blob(cfg, blob, offs, max_batch=B, dec_weight_fp8=fp8, enc_fp8=fp8, act_fp16=f16, cross_kv_fp8=fp8)
    eng = model.engine
    n_samp = cfg.n_mel_frames * 160
    wav = torch.from_numpy(np.stack([synth.synth_clip(500 + j, n_samp) for j in range(B)])).to(dev)
    gp = synth.bench_gen_params(cfg, max_new_tokens=max_new, accept_mode=ACCEPT_TYPICAL)

    def step():
        eng.encode(eng.logmel(wav))
        seqs = eng.decode(gp, B)
        return sum(len(s) - len(gp.prompt) for s in seqs), eng.stats()
    step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tok = it = sched = 0; ms_dec = ms_enc = 0.0
    for _ in range(steps):
        n, st = step(); tok += n; it += st["iterations"]; ms_dec += st["ms_decode"]; ms_enc += st["ms_encode"]; sched += st.get("schedule_steps", 0)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    try:
        parity = leg_parity(cfg, sd_cpu, eng, gp, fp8, act="f16" if f16 else "hilo")
    except Exception as e:  # noqa: BLE001
        parity = {"parity_checked": False, "parity_error": repr(e)}
    del sd_cpu
    gpv = synth.bench_gen_params(cfg, max_new_tokens=max_new, vanilla=True)
    eng.decode(gpv, B); eng.decode(gpv, B)
    stv = eng.stats()
    van = B * max_new / (stv["ms_decode"] * 1e-3)
    t_iter = ms_dec / max(it, 1)
    bytes_iter = decode_iter_bytes(cfg, B, len(gp.prompt) + max_new / 2, fp8)
    out = {"config": name, "streams": B, "heads": heads, "fp8_decoder_weights": fp8, "fp8_cross_kv": fp8, "steps": steps,
           "decode_operands": "fp16 single plane" if f16 else "bf16 hi/lo pair",
           "tokens_per_sec": round(tok / el, 1), "decode_tokens_per_sec": round(tok / (ms_dec * 1e-3), 1),
           "ms_per_iteration": round(t_iter, 4), "tokens_per_iteration": round(tok / max(it, 1) / B, 3),
           "vanilla_tokens_per_sec": round(van, 1), "medusa_over_vanilla": round(tok / (ms_dec * 1e-3) / van, 3),
           "roofline": {"bound": "hbm", "bytes_per_launch": round(bytes_iter), "ms_per_launch": round(t_iter, 4),
                        "achieved": round(bytes_iter / (t_iter * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                        "frac": round(bytes_iter / (t_iter * 1e-3) / 8e12, 4)},
           "roofline_frac_hbm": round(bytes_iter / (t_iter * 1e-3) / 8e12, 4),
           "merged_steps_per_iteration": round(sched / max(it, 1), 3) if sched else None,
           "prefill_tflops": round(prefill_flops(cfg) * B / (ms_enc / steps * 1e-3) / 1e12, 1),
           "prefill": prefill_roofline(cfg, B, ms_enc / steps, fp8)}
    out["prefill_frac_mfma"] = out["prefill"]["frac"]        # fp8 leg: against the blended fp8 / bf16 peak (prefill_roofline)
    out.update(parity)
    if B > 1:
        # the ratio at a STATED acceptance (the reference's x1.5 implies ~3 tokens per iteration, /root/reference/README.md:34-35): forced accept
        # lengths 0..3 on this leg's encoder state — ms per iteration, merged steps per iteration, Medusa / vanilla at that accept length
        try:
            out["acceptance_sensitivity"] = acceptance_sensitivity(eng, cfg, gp, B, max_new, fp8, accepts=(0, 1, 2, 3), vanilla_tps=van)
        except Exception as e:  # noqa: BLE001
            out["acceptance_sensitivity"] = {"failed": repr(e)}
    eng.close()
    del model, blob
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1, help="streams per GPU (BASELINE configs[1] = 1)")
    ap.add_argument("--heads", default="linear", choices=["linear", "block"])
    ap.add_argument("--model", default="large-v2", choices=["large-v2", "tiny.en", "micro"])
    ap.add_argument("--max-new", type=int, default=128)
    ap.add_argument("--logit-std", type=float, default=4.5,
                    help="std of the vocabulary logits of the random-init checkpoint; 4.5 gives ~3.5 tokens/iteration "
                         "under typical acceptance, close to the ~3 implied by the reference's x1.5 speed-up")
    ap.add_argument("--micro-batches", type=int, default=1,
                    help="decode the per-GPU batch as this many concurrent micro-batches (whisper_medusa/pool.py); 1 = one context")
    ap.add_argument("--fp8-weights", action="store_true",
                    help="BASELINE configs[4]: encoder QKV / FC1 / cross-K/V projection on the fp8 MFMA (e4m3 x e4m3, per-row scales) and the "
                         "decoder-layer matrices stored as fp8 e4m3 + per-row scale (widened to bf16 in registers: the decode step is HBM-bound)")
    ap.add_argument("--act", choices=["hilo", "f16"], default=None,
                    help="decode numerics contract (include/wm.h wm_config.act_fp16): bf16 hi/lo operand pairs (libwm.so) or one fp16 plane "
                         "(libwm_f16.so); default: the engine's (WM_ACT)")
    ap.add_argument("--bf16-cross-kv", action="store_true", help="with --fp8-weights: keep the decode loop on the bf16 cross-K/V cache (round 5's configs[4] leg; A/B)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-vanilla", action="store_true",
                    help="skip the vanilla-greedy anchor (PMC passes: only Medusa iterations in the counter totals)")
    ap.add_argument("--cpu-iters", type=int, default=8, help="Medusa iterations of the CPU-baseline decode sample")
    ap.add_argument("--cpu-full", action="store_true", help="CPU baseline decodes the whole max-new budget (minutes)")
    ap.add_argument("--test-setup-delay-s", type=float, default=0.0, help=argparse.SUPPRESS)     # test harness only (tests/test_bench_dist.py)
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="skip the BASELINE configs[2] (Block, 32 streams) and configs[4] (fp8, 32 streams) legs and the sensitivity rows")
    args = ap.parse_args()

    from whisper_medusa import MedusaConfig, WhisperMedusaModel, ACCEPT_TYPICAL
    from whisper_medusa import synth, weights, dist as wd

    from whisper_medusa.engine import default_act_fp16
    f16 = default_act_fp16() if args.act is None else args.act == "f16"
    rank, local, world = wd.init_from_env()
    assert world == args.gpus or (world == 1 and args.gpus == 1), f"WORLD_SIZE={world} but --gpus {args.gpus}"
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs a HIP device: the engine has no CPU path")
    dev = torch.device("cuda", local % torch.cuda.device_count())     # a launcher may narrow the visible devices per rank
    torch.cuda.set_device(dev)

    heads = "medusa_block" if args.heads == "block" else "base_head"
    if args.model == "large-v2":
        cfg = MedusaConfig.large_v2(heads, K=10)
    elif args.model == "tiny.en":
        cfg = MedusaConfig.tiny_en(heads, K=4)
    else:
        cfg = MedusaConfig.micro(heads_type=heads, K=4)
    B = args.batch

    # ---- weights: rank 0 builds the packed blob, RCCL broadcast over xGMI (one time, untimed) ----
    blob = offs = sd = None
    if rank == 0:
        sd = synth.synth_state_dict(cfg, seed=0, device=str(dev), logit_std=args.logit_std)
        blob, offs = weights.build_blob(cfg, sd, device=dev, dec_fp8=args.fp8_weights, enc_fp8=args.fp8_weights, act_fp16=f16)
    t0 = time.time()
    blob, offs = wd.broadcast_blob(blob, offs, device=dev)
    torch.cuda.synchronize()
    if args.test_setup_delay_s > 0:      # tests/test_bench_dist.py stretches the one-time set-up phase to show that it lies outside the timed region
        time.sleep(args.test_setup_delay_s)
    t_bcast = time.time() - t0
    xkv8 = bool(args.fp8_weights) and not args.bf16_cross_kv
    model = WhisperMedusaModel.from_blob(cfg, blob, offs, max_batch=B, dec_weight_fp8=args.fp8_weights, enc_fp8=args.fp8_weights, act_fp16=f16,
                                         cross_kv_fp8=xkv8)
    eng = model.engine

    # ---- inputs resident in HBM ----
    n_samp = cfg.n_mel_frames * 160
    n_sets = 4                               # different clips on successive steps (4 sets, cycled), all resident in HBM
    wavs = [torch.from_numpy(np.stack([synth.synth_clip((k * world + rank) * B + j, n_samp) for j in range(B)])).to(dev)
            for k in range(n_sets)]
    gp = synth.bench_gen_params(cfg, max_new_tokens=args.max_new, accept_mode=ACCEPT_TYPICAL)
    step_no = [0]

    pool = None
    if args.micro_batches > 1:
        from whisper_medusa.pool import ContextPool
        pool = ContextPool(cfg, blob, offs, args.micro_batches, B, dec_weight_fp8=args.fp8_weights, enc_fp8=args.fp8_weights, act_fp16=f16, cross_kv_fp8=bool(args.fp8_weights) and not args.bf16_cross_kv)

    def step():
        wav = wavs[step_no[0] % n_sets]
        step_no[0] += 1
        if pool is not None:
            seqs = pool.run(wav, gp, from_wav=True)
            return sum(len(s) - len(gp.prompt) for s in seqs), pool.last_stats
        feats = eng.logmel(wav)
        eng.encode(feats)
        seqs = eng.decode(gp, B)
        st = eng.stats()
        return sum(len(s) - len(gp.prompt) for s in seqs), st

    for _ in range(args.warmup):
        step()
    wd.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tokens = 0
    iters = 0
    ms_dec = ms_enc = ms_mel = 0.0
    hist = np.zeros(cfg.medusa_num_heads + 1, dtype=np.int64)
    replays = 0
    sched_steps = 0
    sib_hits = 0
    for _ in range(args.steps):
        n, st = step()
        tokens += n
        iters += st["iterations"]
        ms_dec += st["ms_decode"]; ms_enc += st["ms_encode"]; ms_mel += st["ms_logmel"]
        hist += np.asarray(st["accept_hist"], dtype=np.int64)
        replays += st["graph_replays"]
        sched_steps += st.get("schedule_steps", 0)
        sib_hits += st.get("sibling_hits", 0)
    torch.cuda.synchronize()
    wd.barrier()
    elapsed = wd.max_over_ranks(time.perf_counter() - t0, dev)
    tokens_all = wd.sum_over_ranks(float(tokens), dev)
    per_rank_tokens = wd.gather_floats(float(tokens), dev)          # one entry per rank: every rank of the launch really decoded

    # ---- anchor: vanilla greedy decoding on the same engine / clips / budget (one untimed-region step) ----
    vanilla_tps = vanilla_ms_step = None
    if not args.no_vanilla:
        gpv = synth.bench_gen_params(cfg, max_new_tokens=args.max_new, vanilla=True)
        if pool is not None:
            eng.encode(eng.logmel(wavs[0]))      # the timed steps ran on the pool's contexts
        eng.decode(gpv, B)                       # warm
        eng.decode(gpv, B)
        stv = eng.stats()
        vanilla_tps = B * args.max_new / (stv["ms_decode"] * 1e-3)
        vanilla_ms_step = stv["ms_decode"] / max(stv["iterations"], 1)

    if rank != 0:
        return

    # HBM traffic per iteration from the PMC pass (profiles/): only if that pass was taken on THESE kernels (sha over csrc/)
    traffic = None; traffic_src = None
    tname = next((n for n in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json") if os.path.exists(os.path.join(ROOT, "profiles", n))), "r05_pmc_traffic.json")
    tpath = os.path.join(ROOT, "profiles", tname)
    if args.model == "large-v2" and B == 1 and args.heads == "linear" and not args.fp8_weights and os.path.exists(tpath):
        tj = json.load(open(tpath))
        if tj.get("kernels_sha") == kernels_sha():
            traffic = tj["medusa_iteration_bytes"]      # PMC FETCH_SIZE x2, see profiles/
            traffic_src = {"file": "profiles/" + tname, "command": tj.get("command"), "kernels_sha": tj.get("kernels_sha"),
                           "counts": tj.get("counts"), "without_prefetch_blocks": tj.get("without_prefetch_blocks")}
        else:
            traffic_src = {"file": "profiles/" + tname, "stale": True, "file_kernels_sha": tj.get("kernels_sha"), "kernels_sha": kernels_sha()}
    t_iter_ms = ms_dec / max(iters, 1)
    mean_len = len(gp.prompt) + args.max_new / 2
    bytes_iter = decode_iter_bytes(cfg, B, mean_len, args.fp8_weights)
    achieved = bytes_iter / (t_iter_ms * 1e-3) / 1e9
    # what actually ran (the carry skips work SURVEY §8d's figure counts): a stream's base pass follows an accept length of 0 (and opens
    # every decode call); one stream: the host then skips the whole pass, several streams: only that stream's attention
    n_it_streams = max(int(hist.sum()), 1)
    # (wm_config.sibling_rows, one stream: an accept length of 0 whose next root rode along as a sibling row needs no base pass either)
    p0 = min(1.0, (float(hist[0]) - sib_hits + args.steps * B) / n_it_streams)
    p_base_pass = p0 if (B == 1 and args.micro_batches == 1) else 1.0
    bytes_exec = executed_bytes(cfg, B, mean_len, args.fp8_weights, p_base_pass, p0)
    passes_per_iter = 1.0 + p_base_pass
    schedule = "base pass + verify pass per iteration (hidden-state carry skips the base pass / its attention)"
    if sched_steps > 0:
        # merged-step schedule (several streams): every step is ONE pass — the weights once, each live stream's cross- / self-K/V once —
        # and the slowest stream's iteration takes 1 + P(accept length 0) steps
        passes_per_iter = sched_steps / max(args.micro_batches, 1) / max(iters, 1)
        bytes_exec = passes_per_iter * executed_bytes(cfg, B, mean_len, args.fp8_weights, 0.0, 0.0)
        schedule = "merged step: one pass per step; a stream that accepted nothing contributes its one base row to the others' verify pass"
    parts = decode_iter_bytes(cfg, B, mean_len, args.fp8_weights, parts=True)
    van_bytes = parts["w_vanilla"] + B * (parts["cross_kv_per_stream_pass"] + parts["self_kv_per_stream_pass"])
    gemm_rows = min(16, B * (cfg.medusa_num_heads + 1))
    gemm_ms, gemm_bytes = (None, None) if args.no_vanilla else eng.profile_layer_gemms(rows=gemm_rows, reps=50)
    audio_s = args.steps * B * world * 30.0 * cfg.max_source_positions / 1500.0
    tok_per_iter = tokens / max(iters, 1) / B
    out = {
        "metric": "decoded_tokens_per_sec", "value": round(tokens_all / elapsed, 2), "unit": "tokens/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": ("bf16" if not args.fp8_weights else "bf16 + fp8 e4m3 (fp8 MFMA in the encoder's LayerNorm-fed GEMMs and the cross-K/V projection, fp8-stored decoder-layer weights)")
                                      + (" + fp16 decoder GEMM operands (one plane; fp32 accumulate)" if f16 else ""), "data": "synthetic",
        "config": {"workload": f"whisper-{args.model} + medusa-{args.heads} K={cfg.medusa_num_heads}, "
                               f"{B} x 30 s clip(s) per GPU, log-mel+encoder+decode, max_new_tokens={args.max_new}, "
                               f"typical acceptance (T=1.0), hipGraph decode loop, random-init weights (logit_std={args.logit_std})",
                   "streams_per_gpu": B, "micro_batches": args.micro_batches, "fp8_decoder_weights": bool(args.fp8_weights), "fp8_cross_kv": xkv8, "parallelism": f"dp{world}",
                   "decode_operands": "fp16 single plane (wm_config.act_fp16 = 1)" if f16 else "bf16 hi/lo pair (wm_config.act_fp16 = 0)",
                   "max_new_tokens": args.max_new},
        "tokens_per_sec_per_gpu": round(tokens_all / elapsed / world, 2),
        "ranks_seen": len(per_rank_tokens), "tokens_per_rank": [int(t) for t in per_rank_tokens],
        "rtf": round(elapsed / audio_s, 6), "x_realtime": round(audio_s / elapsed, 2),
        "decode_tokens_per_sec_per_gpu": round(tokens / (ms_dec * 1e-3), 2),
        "iters_per_sec": round(iters / (ms_dec * 1e-3), 2), "tokens_per_iter": round(tok_per_iter, 3),
        "accept_hist": hist.tolist(), "graph_replays": int(replays),
        "sibling_rows": int(getattr(eng, "sibling_rows", 0)) if B == 1 else 0, "sibling_hits": int(sib_hits),
        "ms_logmel_per_step": round(ms_mel / args.steps, 3), "ms_encode_per_step": round(ms_enc / args.steps, 3),
        "ms_decode_per_step": round(ms_dec / args.steps, 3), "weight_broadcast_s": round(t_bcast, 3),
        "vanilla_anchor": None if vanilla_tps is None else
                          {"tokens_per_sec_per_gpu": round(vanilla_tps, 2), "ms_per_token_step": round(vanilla_ms_step, 4),
                           "medusa_over_vanilla": round(tokens / (ms_dec * 1e-3) / vanilla_tps, 3)},
        "roofline": {"bound": "hbm", "kernel": "decode iteration (verify pass + base pass unless the hidden state was carried; hipGraph replays)",
                     "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 4),
                     "traffic": traffic, "traffic_source": traffic_src, "bytes_per_launch": round(bytes_iter), "ms_per_launch": round(t_iter_ms, 4),
                     # the same iteration priced on the bytes it executed (frac above keeps SURVEY §8d's two-pass numerator)
                     "schedule": schedule, "passes_per_iteration": round(passes_per_iter, 3), "base_attention_per_stream_iteration": round(p0, 3),
                     "bytes_executed": round(bytes_exec), "frac_executed": round(bytes_exec / (t_iter_ms * 1e-3) / 8e12, 4),
                     "vanilla_step": None if vanilla_ms_step is None else
                                     {"bytes": round(van_bytes), "ms": round(vanilla_ms_step, 4),
                                      "frac": round(van_bytes / (vanilla_ms_step * 1e-3) / 8e12, 4)},
                     "prefill": prefill_roofline(cfg, B, ms_enc / args.steps, args.fp8_weights),
                     "layer_gemms": None if gemm_ms is None else
                                    {"rows": gemm_rows, "ms": round(gemm_ms, 5), "bytes": round(gemm_bytes),
                                     "achieved_gbs": round(gemm_bytes / (gemm_ms * 1e-3) / 1e9, 1)}},
    }

    # ---- acceptance sensitivity + the other single-GPU BASELINE configs (untimed by the driver, same process) ----
    if not args.no_extra_configs and world == 1 and not args.no_vanilla:
        try:
            out["acceptance_sensitivity"] = acceptance_sensitivity(eng, cfg, gp, B, args.max_new, args.fp8_weights)
        except Exception as e:  # noqa: BLE001
            out["acceptance_sensitivity"] = {"failed": repr(e)}
        if args.model == "large-v2" and B == 1 and args.heads == "linear" and not args.fp8_weights:
            extra = []
            for name, heads_x, Bx, fp8x in (("large-v2 + Medusa-Block K=10, 1 stream (the batch the reference's x1.37 is quoted on)", "medusa_block", 1, False),
                                            ("configs[2] large-v2 + Medusa-Block K=10, 32 streams", "medusa_block", 32, False),
                                            ("configs[1] shape at 32 streams (Medusa-Linear)", "base_head", 32, False),
                                            ("configs[4] fp8 (encoder fp8 MFMA + fp8 decoder weights) + Medusa-Linear, 32 streams", "base_head", 32, True)):
                try:
                    extra.append(extra_config(name, heads_x, Bx, fp8x, dev, args.logit_std, args.max_new, f16=f16))
                except Exception as e:  # noqa: BLE001
                    extra.append({"config": name, "failed": repr(e)})
            out["configs"] = extra

    # ---- CPU baseline: the oracle (a port of the reference algorithm) on the host cores, bounded sample ----
    if not args.no_cpu_baseline and world == 1:
        try:
            eng.encode(eng.logmel(wavs[0][:1].contiguous()))
            ids0 = eng.decode(gp, 1)[0]
            out["cpu_baseline"] = cpu_baseline_leg(cfg, sd, eng, gp, wavs[0][0], ids0, args.cpu_iters, args.fp8_weights, args.cpu_full,
                                                   act="f16" if f16 else "hilo", xkv8=xkv8)
        except Exception as e:  # noqa: BLE001
            out["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": os.cpu_count(), "kind": "port",
                                   "sample": f"failed: {e!r}", "parity_checked": False}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
