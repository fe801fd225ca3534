"""Whisper-large-v2 shape on the GPU: size-independent properties at BASELINE.json's full size
(the oracle would need minutes per pass here, so it only spot-checks one prompt pass)."""
import numpy as np
import pytest
import torch

from helpers import MedusaConfig, synth, ACCEPT_GREEDY, ACCEPT_TYPICAL
from whisper_medusa import WhisperMedusaModel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def large(gpu):
    cfg = MedusaConfig.large_v2("base_head", K=10)
    sd = synth.synth_state_dict(cfg, seed=0, device=str(gpu))
    model = WhisperMedusaModel(cfg, sd, device=gpu, max_batch=2)
    wav = np.stack([synth.synth_clip(i) for i in range(2)])
    feats = model.extract_features(wav)
    yield cfg, sd, model, feats
    model.engine.close()


def test_large_greedy_equals_vanilla_and_batch_consistency(large):
    cfg, sd, model, feats = large
    eng = model.engine
    gp = synth.bench_gen_params(cfg, max_new_tokens=48, accept_mode=ACCEPT_GREEDY)
    eng.encode(feats)
    med = eng.decode(gp, 2)
    gp.vanilla = True
    van = eng.decode(gp, 2)
    for b in range(2):
        n = min(len(med[b]), len(van[b]))
        assert n >= 4 + 40 and med[b][:n] == van[b][:n]
        assert all(0 <= t < cfg.vocab_size for t in med[b])
    gp = synth.bench_gen_params(cfg, max_new_tokens=48, accept_mode=ACCEPT_TYPICAL)
    both = eng.decode(gp, 2)
    eng.encode(feats[1:2].contiguous())
    alone = eng.decode(gp, 1)[0]
    assert alone == both[1]
    st = eng.stats()
    assert st["graph_replays"] > 0          # the steady state ran as hipGraph replays


def test_large_prompt_pass_against_oracle(large):
    """one 4-token prompt pass of all 11 heads vs the oracle fed with the engine's encoder output"""
    from oracle.whisper_medusa_oracle import Oracle
    cfg, sd, model, feats = large
    eng = model.engine
    eng.encode(feats)
    enc = eng.encoder_output(1)[0]
    prompt = synth.default_prompt(cfg)
    z = eng.forward_logits([prompt], 0, False)[:, 0]
    orc = Oracle(cfg, {k: v.float().cpu() for k, v in sd.items()}, sim="bf16")
    ref = orc.decoder_pass(orc.new_state(enc), prompt, 0, disable_medusa=False)
    d = (z - ref).abs()
    print('large prompt pass: max|d|', float(d.max()), 'mean|d|', float(d.mean()), 'ref max', float(ref.abs().max()))
    assert d.max() <= 8e-2 and d.mean() <= 5e-3, (float(d.max()), float(d.mean()))
    assert (z[:, -1].argmax(-1) == ref[:, -1].argmax(-1)).all()
