"""Sibling rows (include/wm.h: wm_config.sibling_rows; VERDICT r05 item 6, SURVEY §8f row 4 "single-pass-per-iteration fusion").

One stream, candidate chain: the verify pass's 16-row tile has 15 - K spare rows; up to five of them carry head 1's top-2 .. top-6 tokens as leaves
under the root.  The acceptance rule never sees them (medusa_utils.py:526-641 runs on the chain: emitted ids are the reference's); when the chain accepts
nothing and the next root (argmax v_0, model.py:710-713) is one of them, that row's hidden state and K/V rows replace the next base pass.
What must hold: ids == the oracle's chain ids (tie-aware like every decode-loop comparison), with the rows on and off; the engine's hit count == the
oracle's count of such iterations whenever the ids are strictly equal; hits do occur."""
import numpy as np
import pytest
import torch

from helpers import MedusaConfig, synth, golden_gen_params, clip_for, check_tokens, ACCEPT_TYPICAL, ACCEPT_GREEDY
from oracle.whisper_medusa_oracle import Oracle
from whisper_medusa import WhisperMedusaModel

pytestmark = pytest.mark.gpu

SHAPES = {
    "micro": (lambda: MedusaConfig.micro(K=4), 11),
    "micro10": (lambda: MedusaConfig.micro(K=10, d_model=128, layers=2), 12),
    "microblock": (lambda: MedusaConfig.micro(K=4, heads_type="medusa_block"), 13),
    "tiny": (lambda: MedusaConfig.tiny_en(K=4), 0),
}


def _models(gpu, monkeypatch, cfg, sd):
    monkeypatch.setenv("WM_SIBLINGS", "5")
    on = WhisperMedusaModel(cfg, sd, device=gpu, max_batch=1)
    assert on.engine.sibling_rows == 5
    monkeypatch.setenv("WM_SIBLINGS", "0")
    off = WhisperMedusaModel(cfg, sd, device=gpu, max_batch=1)
    assert off.engine.sibling_rows == 0
    return on, off


@pytest.mark.parametrize("mode", [ACCEPT_TYPICAL, ACCEPT_GREEDY])
@pytest.mark.parametrize("shape", list(SHAPES))
def test_sibling_rows_emit_the_chain_ids(gpu, monkeypatch, shape, mode):
    mk, seed = SHAPES[shape]
    cfg = mk()
    sd = synth.synth_state_dict(cfg, seed=seed)
    on, off = _models(gpu, monkeypatch, cfg, sd)
    orc = Oracle(cfg, sd, sim="bf16")
    S = min(5, 15 - cfg.medusa_num_heads)
    gp = golden_gen_params(cfg, mode, 40)
    hits = 0
    for i in range(3):
        feats = on.extract_features(clip_for(cfg, i))
        on.engine.encode(feats)
        enc = on.engine.encoder_output(1)
        got_on = on.engine.decode(gp, 1)[0]
        st = on.engine.stats()
        off.engine.set_encoder_output(enc)
        got_off = off.engine.decode(gp, 1)[0]
        assert off.engine.stats()["sibling_hits"] == 0
        check_tokens(orc, enc[0], gp, got_on, f"siblings on {shape} clip {i}")
        if got_off != got_on:                    # (a sibling row's state equals the base pass's up to fp32 summation order: a near-tie may resolve differently)
            check_tokens(orc, enc[0], gp, got_off, f"siblings off {shape} clip {i}")
        ref = orc.decode(enc[0], gp, siblings=S)
        if got_on == ref.ids:
            assert st["sibling_hits"] == ref.sibling_hits, (shape, i, st["sibling_hits"], ref.sibling_hits, ref.accept_lengths)
            assert st["accept_hist"][0] >= st["sibling_hits"]
        hits += st["sibling_hits"]
    print(f"sibling rows {shape} mode {mode}: {hits} hits over 3 clips")
    on.engine.close(); off.engine.close()


def test_sibling_rows_do_hit_and_skip_base_passes(gpu, monkeypatch):
    """micro10 (K = 10: five spare rows), exact-match acceptance, six clips: hits occur — the oracle counts them too — and the engine's count is the
    oracle's on every clip whose ids are strictly the oracle's."""
    cfg = MedusaConfig.micro(K=10, d_model=128, layers=2)
    sd = synth.synth_state_dict(cfg, seed=12)
    on, off = _models(gpu, monkeypatch, cfg, sd)
    orc = Oracle(cfg, sd, sim="bf16")
    gp = golden_gen_params(cfg, ACCEPT_GREEDY, 40)
    total = want = 0
    for i in range(6):
        feats = on.extract_features(clip_for(cfg, i))
        on.engine.encode(feats)
        enc = on.engine.encoder_output(1)
        got = on.engine.decode(gp, 1)[0]
        st = on.engine.stats()
        check_tokens(orc, enc[0], gp, got, f"sibling hits micro10 clip {i}")
        ref = orc.decode(enc[0], gp, siblings=5)
        if got == ref.ids:
            assert st["sibling_hits"] == ref.sibling_hits, (i, st["sibling_hits"], ref.sibling_hits)
        total += st["sibling_hits"]; want += ref.sibling_hits
    assert total > 0 and want > 0, (total, want)
    on.engine.close(); off.engine.close()


def test_sibling_rows_are_off_outside_single_stream_chain_decodes(gpu, monkeypatch):
    """two streams (merged-step schedule: the verify rows are dense), vanilla greedy, a candidate tree: no sibling rows, no hits"""
    monkeypatch.setenv("WM_SIBLINGS", "5")
    cfg = MedusaConfig.micro(K=4)
    sd = synth.synth_state_dict(cfg, seed=11)
    m = WhisperMedusaModel(cfg, sd, device=gpu, max_batch=2)
    feats = m.extract_features([clip_for(cfg, 0), clip_for(cfg, 1)])
    m.engine.encode(feats)
    gp = golden_gen_params(cfg, ACCEPT_TYPICAL, 30)
    m.engine.decode(gp, 2)
    assert m.engine.stats()["sibling_hits"] == 0
    gpv = golden_gen_params(cfg, ACCEPT_GREEDY, 30); gpv.vanilla = True
    m.engine.encode(m.extract_features(clip_for(cfg, 0)))
    m.engine.decode(gpv, 1)
    assert m.engine.stats()["sibling_hits"] == 0
    m.engine.close()
    cfgt = MedusaConfig.micro(K=4, medusa_choices=[1, 3, 2, 1, 1])
    mt = WhisperMedusaModel(cfgt, synth.synth_state_dict(cfgt, seed=11), device=gpu, max_batch=1)
    mt.engine.encode(mt.extract_features(clip_for(cfgt, 0)))
    mt.engine.decode(golden_gen_params(cfgt, ACCEPT_TYPICAL, 30), 1)
    assert mt.engine.stats()["sibling_hits"] == 0
    mt.engine.close()
