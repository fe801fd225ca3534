"""The two decode numerics contracts side by side (include/wm.h wm_config.act_fp16): bf16 hi / lo operand pairs (libwm.so) and the
fp16 single-plane operand (libwm_f16.so).  The rest of the GPU suite runs on the process default (WM_ACT); this file pins BOTH explicitly, each
against the oracle in the same contract (Oracle(act=...)): decode loops (Linear / Block, typical / exact-match), batch invariance, the large-v2
loop and logits, and the agreement of the two contracts with each other and with the fp32-pinned table."""
import numpy as np
import pytest
import torch

from helpers import MedusaConfig, synth, check_tokens, golden_gen_params, record_table, ACCEPT_GREEDY, ACCEPT_TYPICAL
from whisper_medusa import WhisperMedusaModel

pytestmark = pytest.mark.gpu
ACTS = [pytest.param(False, id="hilo"), pytest.param(True, id="f16")]


def _cpu(sd):
    return {k: v.float().cpu() for k, v in sd.items()}


def _orc(cfg, sd, f16, **kw):
    from oracle.whisper_medusa_oracle import Oracle
    return Oracle(cfg, _cpu(sd), sim="bf16", act="f16" if f16 else "hilo", **kw)


@pytest.mark.parametrize("f16", ACTS)
@pytest.mark.parametrize("shape", ["micro", "microblock", "tiny", "tinyblock"])
def test_decode_loop_matches_the_oracle_in_its_contract(gpu, shape, f16):
    heads = "medusa_block" if shape.endswith("block") else "base_head"
    cfg = MedusaConfig.micro(K=4, heads_type=heads) if shape.startswith("micro") else MedusaConfig.tiny_en(heads, K=4)
    sd = synth.synth_state_dict(cfg, seed=21, device=str(gpu))
    model = WhisperMedusaModel(cfg, sd, device=gpu, max_batch=3, act_fp16=f16)
    eng = model.engine
    assert bool(eng.lib.wm_build_act_fp16()) == f16
    n = cfg.n_mel_frames * 160
    feats = model.extract_features(np.stack([synth.synth_clip(70 + i, n) for i in range(3)]))
    orc = _orc(cfg, sd, f16)
    for mode in (ACCEPT_TYPICAL, ACCEPT_GREEDY):
        gp = golden_gen_params(cfg, mode, 32)
        eng.encode(feats)
        enc = eng.encoder_output(3)
        both = eng.decode(gp, 3)
        for b in range(3):
            check_tokens(orc, enc[b], gp, both[b], label=f"{shape} act={'f16' if f16 else 'hilo'} mode={mode} b={b}")
        # batch invariance: a stream's tokens do not depend on the streams it shares launches with
        eng.encode(feats[1:2].contiguous())
        assert eng.decode(gp, 1)[0] == both[1]
    model.engine.close()


@pytest.mark.parametrize("f16", ACTS)
@pytest.mark.parametrize("heads", ["base_head", "medusa_block"])
def test_two_tile_passes_leave_the_single_stream_cache_bit_for_bit(gpu, heads, f16):
    """Batch invariance below the token level.  A pass of 17..32 rows (two streams with a long prompt: the 2 x 16-row prefix chunks, the 2 x 9-row first
    base pass) runs the two-tile kernels where a single stream runs the 16-row kernel; the self-K/V cache they leave must be the single-stream
    run's bit for bit.  Probed through wm_forward_logits at position P (that entry point walks streams one at a time: the probe itself has no
    batch): logits from the cache of the two-stream run == logits from the cache of each stream alone, exactly.
    (Round 6: with the LayerNorm folded, hipcc contracted rstd * (acc - mean c) + bias into an fma in one GEMM kernel and not in another — the
    cross-q rows of a two-tile pass differed in the last bit of 28 % of their elements, and a token flipped now and then; csrc/wm_common.h nofuse.)"""
    cfg = MedusaConfig.micro(K=4, heads_type=heads, n_tgt=96)
    sd = synth.synth_state_dict(cfg, seed=41)
    model = WhisperMedusaModel(cfg, sd, device=gpu, max_batch=2, act_fp16=f16)
    eng = model.engine
    n = cfg.n_mel_frames * 160
    feats = model.extract_features([synth.synth_clip(3, n), synth.synth_clip(4, n)[: n // 2]])
    eng.encode(feats)
    enc = eng.encoder_output(2)
    for plen in (23, 15, 7, 3):                    # 2 x (16 + 9), 2 x (16 + 1), 2 x 9, 2 x 5 rows
        pid = torch.tensor([cfg.vocab_size - 5] + [10 + (7 * i) % 900 for i in range(plen - 1)])
        gp = model._gen_params(None, None, (6, 1.3), 20, None, None, False, None, None, None, None, pid)
        P = len(gp.prompt)
        eng.set_encoder_output(enc)
        eng.decode(gp, 2, max_iters=1)
        two = eng.forward_logits([[5], [5]], P, False)
        for b in range(2):
            eng.set_encoder_output(enc[b: b + 1])
            eng.decode(gp, 1, max_iters=1)
            one = eng.forward_logits([[5]], P, False)[:, 0]
            assert torch.equal(two[:, b], one), (heads, f16, P, b, float((two[:, b] - one).abs().max()))
    eng.close()


@pytest.mark.parametrize("f16", ACTS)
def test_prompt_pass_logits_in_its_contract(gpu, f16):
    """all heads' logits of one prompt pass (tiny.en) against the oracle in the same contract"""
    cfg = MedusaConfig.tiny_en(K=4)
    sd = synth.synth_state_dict(cfg, seed=5, device=str(gpu))
    model = WhisperMedusaModel(cfg, sd, device=gpu, max_batch=1, act_fp16=f16)
    eng = model.engine
    n = cfg.n_mel_frames * 160
    eng.encode(model.extract_features(synth.synth_clip(9, n)[None]))
    enc = eng.encoder_output(1)[0]
    prompt = synth.default_prompt(cfg)
    z = eng.forward_logits([prompt], 0, False)[:, 0]
    orc = _orc(cfg, sd, f16)
    ref = orc.decoder_pass(orc.new_state(enc), prompt, 0, disable_medusa=False)
    scale = float(ref.abs().max())
    rel = float((z - ref).abs().max()) / scale
    record_table(f"tiny.en prompt pass, engine vs oracle, contract {'f16' if f16 else 'hilo'}", max_rel_to_scale=round(rel, 7), logit_scale=round(scale, 3))
    print("tiny prompt pass", "f16" if f16 else "hilo", "max rel to scale", rel)
    assert rel <= (2e-3 if f16 else 2e-4)
    model.engine.close()


def test_large_f16_contract_against_oracle_and_the_fp32_table(gpu):
    """large-v2 + Medusa-Linear K=10 on the fp16 single-plane contract: decode loop against the oracle in that contract (one stream and one of
    four), prompt-pass logits (recorded next to the hi / lo figure), the fp32-pinned table (ids minted offline from the oracle's reference-pinned
    mode, audio -> tokens), and the tokens of the hi / lo contract on the same clips."""
    from helpers import fp32_golden, fp32_agreement_rows
    cfg = MedusaConfig.large_v2("base_head", K=10)
    sd = synth.synth_state_dict(cfg, seed=0, device="cpu", logit_std=4.5)          # the checkpoint of the fp32-pinned table (CPU generator)
    sd_gpu = {k: v.to(gpu) for k, v in sd.items()}
    g = fp32_golden("large", sd)
    seed, clip0, N, NEW = (int(x) for x in g["large_meta"])
    n = cfg.n_mel_frames * 160
    wav = np.stack([synth.synth_clip(clip0 + i, n) for i in range(N)])
    out = {}
    for f16 in (True, False):
        model = WhisperMedusaModel(cfg, sd_gpu, device=gpu, max_batch=N, act_fp16=f16)
        eng = model.engine
        feats = model.extract_features(wav)
        for mode in (ACCEPT_TYPICAL, ACCEPT_GREEDY):
            gp = synth.bench_gen_params(cfg, max_new_tokens=NEW, accept_mode=mode)
            eng.encode(feats)
            out[(f16, mode)] = eng.decode(gp, N)
        if f16:
            gp = synth.bench_gen_params(cfg, max_new_tokens=NEW, accept_mode=ACCEPT_TYPICAL)
            orc = _orc(cfg, sd, True)
            eng.encode(feats[:4].contiguous())
            enc = eng.encoder_output(4)
            four = eng.decode(gp, 4)
            check_tokens(orc, enc[2], gp, four[2], label="large f16 B=4 stream 2")
            eng.encode(feats[:1].contiguous())
            one = eng.decode(gp, 1)[0]
            assert one == four[0]
            check_tokens(orc, enc[0], gp, one, label="large f16 B=1")
            prompt = synth.default_prompt(cfg)
            z = eng.forward_logits([prompt], 0, False)[:, 0]
            ref = orc.decoder_pass(orc.new_state(eng.encoder_output(1)[0]), prompt, 0, disable_medusa=False)
            scale = float(ref.abs().max())
            rel, mean = float((z - ref).abs().max()) / scale, float((z - ref).abs().mean()) / scale
            record_table("large-v2 prompt pass, all 11 heads: engine logits vs oracle, fp16 single-plane contract", logit_scale=round(scale, 3),
                         max_rel_to_scale=round(rel, 6), mean_rel_to_scale=round(mean, 7))
            print("large f16 prompt pass: max rel", rel, "mean rel", mean)
            assert rel <= 1.5e-3 and mean <= 3e-4
        model.engine.close()
        del model
        torch.cuda.empty_cache()
    for mode, m in ((ACCEPT_GREEDY, "greedy"), (ACCEPT_TYPICAL, "typical")):
        rows = fp32_agreement_rows(g, "large", mode, out[(True, mode)])
        agree, total = sum(r[1] for r in rows), sum(r[2] for r in rows)
        same = sum(int(a == b) for a, b in zip(out[(True, mode)], out[(False, mode)]))
        record_table(f"fp32-pinned end-to-end large-v2 {m}, fp16 single-plane contract", agree=agree, total=total, frac=round(agree / total, 4),
                     clips_equal_to_hilo_contract=same, clips=N)
        print(f"large f16 {m}: {agree} / {total} ids agree with the fp32-pinned table; {same} / {N} clips identical to the hi / lo contract")
        assert agree >= 0.97 * total


@pytest.mark.parametrize("f16", ACTS)
@pytest.mark.parametrize("heads", ["base_head", "medusa_block"])
def test_fp8_cross_kv_matches_the_oracle(gpu, heads, f16):
    """wm_config.cross_kv_fp8 on the tiny.en shape, both contracts: the decode loop reads the encoder cross-K/V from its e4m3 copy (per-head
    scales on the query / the output); the oracle quantises its cross-K/V the same way (Oracle(xkv_fp8=True)).  Also: the bf16 tap still returns
    the unquantised projection, and the quantised run differs from the bf16-cache run (the switch does something)."""
    cfg = MedusaConfig.tiny_en(heads, K=4)
    sd = synth.synth_state_dict(cfg, seed=23, device=str(gpu))
    n = cfg.n_mel_frames * 160
    wav = np.stack([synth.synth_clip(120 + i, n) for i in range(3)])
    from oracle.whisper_medusa_oracle import Oracle
    outs = {}
    for x8 in (True, False):
        model = WhisperMedusaModel(cfg, sd, device=gpu, max_batch=3, act_fp16=f16, cross_kv_fp8=x8)
        eng = model.engine
        feats = model.extract_features(wav)
        orc = Oracle(cfg, _cpu(sd), sim="bf16", act="f16" if f16 else "hilo", xkv_fp8=x8)
        gp = golden_gen_params(cfg, ACCEPT_TYPICAL, 32)
        eng.encode(feats)
        enc = eng.encoder_output(3)
        both = eng.decode(gp, 3)
        for b in range(3):
            check_tokens(orc, enc[b], gp, both[b], label=f"tiny {heads} xkv8={x8} act={'f16' if f16 else 'hilo'} b={b}")
        prompt = synth.default_prompt(cfg)
        eng.encode(feats[:1].contiguous())
        z = eng.forward_logits([prompt], 0, False)[:, 0]
        ref = orc.decoder_pass(orc.new_state(eng.encoder_output(1)[0]), prompt, 0, disable_medusa=False)
        rel = float((z - ref).abs().max()) / float(ref.abs().max())
        print("tiny", heads, "xkv8", x8, "f16", f16, "prompt pass max rel to scale", rel)
        assert rel <= 3e-3
        outs[x8] = z
        model.engine.close()
    assert float((outs[True] - outs[False]).abs().max()) > 1e-4
