"""The oracle against the golden vectors minted FROM THE REFERENCE'S OWN CODE (oracle/make_golden.py):
medusa_utils known answers, reference forward()/loop runs; plus live cross-checks against the installed
HF Whisper modules.  CPU only."""
import numpy as np
import pytest
import torch

from helpers import GOLD, GOLDEN_MODELS, GOLDEN_BLOCK_MODELS, golden_gen_params, clip_for, synth, ACCEPT_TYPICAL, ACCEPT_GREEDY, GenParams, MedusaConfig
from oracle.whisper_medusa_oracle import Oracle, log_mel, mel_filter_bank, evaluate_posterior_chain, process_logits

KAT = np.load(f"{GOLD}/medusa_utils_kat.npz")
RUNS = {**np.load(f"{GOLD}/reference_linear_runs.npz"), **np.load(f"{GOLD}/reference_block_runs.npz")}


def _gp(mode):
    return GenParams(prompt=[1, 2], eos_token_id=0, pad_token_id=0, accept_mode=mode,
                     temperature=1.0 if mode == ACCEPT_TYPICAL else 0.0)


def test_chain_buffers_are_identity():
    # generate_medusa_buffers([1]*(K+1)) -> all index buffers are 0..K (medusa_utils.py:305-421)
    assert KAT["buf11_tree_indices"].tolist() == list(range(11))
    assert KAT["buf11_retrieve_indices"].tolist() == [list(range(11))]
    assert KAT["buf11_position_ids"].tolist() == list(range(11))
    assert KAT["buf132_retrieve_indices"].tolist() == [[0, 1, 4], [0, 1, 5], [0, 2, 6], [0, 2, 7], [0, 3, 8], [0, 3, 9]]


@pytest.mark.parametrize("mode,key", [(ACCEPT_TYPICAL, "post_accept_typical"), (ACCEPT_GREEDY, "post_accept_greedy")])
def test_evaluate_posterior_matches_reference(mode, key):
    logits, cand, want = KAT["post_logits"], KAT["post_cand"], KAT[key]
    assert len(set(want.tolist())) > 3          # the vectors exercise several accept lengths
    for i in range(len(want)):
        a, _ = evaluate_posterior_chain(torch.from_numpy(logits[i]), torch.from_numpy(cand[i]), _gp(mode))
        assert a == int(want[i]), (i, a, want[i])


def test_generate_candidates_matches_reference():
    base, med = torch.from_numpy(KAT["cand_base"]), torch.from_numpy(KAT["cand_med"])
    z = torch.cat([base[0], med[:, 0, 0]], dim=0)               # [K+1, V] rows = heads
    got = torch.argmax(z, dim=-1)
    assert got.tolist() == KAT["cand_out"][0].tolist() == KAT["cand_tree_out"][0].tolist()


@pytest.mark.parametrize("tag", list(GOLDEN_MODELS) + list(GOLDEN_BLOCK_MODELS))
def test_decode_matches_reference_runs(tag):
    """oracle.decode == reference forward() + reference candidates/posterior, token for token (Linear and Block heads)."""
    mk, seed, max_new = {**GOLDEN_MODELS, **GOLDEN_BLOCK_MODELS}[tag]
    cfg = mk()
    sd = synth.synth_state_dict(cfg, seed=seed)
    orc = Oracle(cfg, sd, sim="fp32")
    feats = torch.from_numpy(log_mel(clip_for(cfg), cfg.num_mel_bins, cfg.n_mel_frames * 160))
    enc = orc.encode(feats)
    probe = enc[::max(1, enc.shape[0] // 8), :16].numpy()
    np.testing.assert_allclose(probe, RUNS[f"{tag}_enc_probe"], atol=2e-4)
    for mode, mname in ((ACCEPT_TYPICAL, "typical"), (ACCEPT_GREEDY, "greedy")):
        for eos_free, ename in ((True, "noeos"), (False, "eos")):
            gp = golden_gen_params(cfg, mode, max_new, suppress_eos=eos_free)
            r = orc.decode(enc, gp)
            want_ids = RUNS[f"{tag}_{mname}_{ename}_ids"].tolist()
            assert r.ids == want_ids, (tag, mname, ename)
            assert r.accept_lengths == RUNS[f"{tag}_{mname}_{ename}_accepts"].tolist()
    st = orc.new_state(enc)
    z = orc.decoder_pass(st, synth.default_prompt(cfg), 0, disable_medusa=False, last_only=True)[:, 0, :64]
    np.testing.assert_allclose(z.numpy(), RUNS[f"{tag}_first_logits_probe"], atol=5e-4)


def test_greedy_mode_equals_vanilla_greedy():
    """temperature==0 verification reproduces plain greedy decoding of head 0."""
    cfg = MedusaConfig.micro(K=4)
    sd = synth.synth_state_dict(cfg, seed=3)
    orc = Oracle(cfg, sd)
    enc = orc.encode(torch.from_numpy(log_mel(clip_for(cfg, 1), 80, cfg.n_mel_frames * 160)))
    gp = golden_gen_params(cfg, ACCEPT_GREEDY, 30)
    med = orc.decode(enc, gp).new_tokens
    gp.vanilla = True
    van = orc.decode(enc, gp).new_tokens
    n = min(len(med), len(van))
    assert med[:n] == van[:n] and n >= 28


@pytest.mark.parametrize("heads", ["base_head", "medusa_block"])
def test_bf16_contract_close_to_fp32(heads):
    cfg = MedusaConfig.micro(K=4, heads_type=heads)
    sd = synth.synth_state_dict(cfg, seed=5)
    feats = torch.from_numpy(log_mel(clip_for(cfg), 80, cfg.n_mel_frames * 160))
    a, b = Oracle(cfg, sd, "fp32"), Oracle(cfg, sd, "bf16")
    ea, eb = a.encode(feats), b.encode(feats)
    assert (ea - eb).abs().max() < 0.15 and (ea - eb).abs().mean() < 0.01
    za = a.decoder_pass(a.new_state(ea), [1, 2, 3], 0, False)
    zb = b.decoder_pass(b.new_state(ea), [1, 2, 3], 0, False)
    assert za.shape == (5, 3, cfg.vocab_size) and (za - zb).abs().max() < 0.25


# ---- live cross-checks against the installed HF Whisper (same arithmetic as the pinned 4.49) ----
@pytest.fixture(scope="module")
def hf_tiny():
    from transformers import WhisperConfig
    from transformers.models.whisper.modeling_whisper import WhisperForConditionalGeneration
    cfg = MedusaConfig.tiny_en(K=4)
    sd = synth.synth_state_dict(cfg, seed=0)
    hc = WhisperConfig(d_model=384, encoder_layers=4, decoder_layers=4, encoder_attention_heads=6,
                       decoder_attention_heads=6, encoder_ffn_dim=1536, decoder_ffn_dim=1536, vocab_size=51864)
    m = WhisperForConditionalGeneration(hc).eval()
    m.load_state_dict({k[len("whisper_model."):]: v for k, v in sd.items() if k.startswith("whisper_model.")}, strict=False)
    return cfg, sd, m


def test_logmel_matches_hf_feature_extractor():
    from transformers import WhisperFeatureExtractor
    fe = WhisperFeatureExtractor()
    np.testing.assert_allclose(mel_filter_bank(), fe.mel_filters, atol=1e-7)
    for i in (0, 3):
        wav = synth.synth_clip(i)
        ref = fe(wav, sampling_rate=16000, return_tensors="np").input_features[0]
        np.testing.assert_allclose(log_mel(wav), ref, atol=5e-5)
    short = synth.synth_clip(1)[:100000]                      # ragged input: padded to 30 s
    ref = fe(short, sampling_rate=16000, return_tensors="np").input_features[0]
    np.testing.assert_allclose(log_mel(short), ref, atol=5e-5)


def test_encoder_and_decoder_match_hf(hf_tiny):
    cfg, sd, m = hf_tiny
    orc = Oracle(cfg, sd)
    feats = torch.from_numpy(log_mel(synth.synth_clip(0)))
    with torch.no_grad():
        e_ref = m.model.encoder(feats[None]).last_hidden_state[0]
    e = orc.encode(feats)
    assert (e - e_ref).abs().max() < 5e-5
    # decoder stack + tied proj_out == Medusa-Block base logits; incremental KV == full forward
    cfgb = MedusaConfig.tiny_en(K=4, heads_type="medusa_block")
    sdb = dict(sd)
    sdb.update({k.replace("whisper_model.model.decoder.layers.3", "medusa_block"): v for k, v in sd.items() if "decoder.layers.3." in k})
    orb = Oracle(cfgb, sdb)
    ids = [50257, 50362, 11, 22, 333, 4444]
    with torch.no_grad():
        ref = m(encoder_outputs=(e_ref[None],), decoder_input_ids=torch.tensor([ids]), use_cache=False).logits[0]
    st = orb.new_state(e_ref)
    z = orb.decoder_pass(st, ids, 0, disable_medusa=True)[0]
    assert (z - ref).abs().max() < 5e-5
    st = orb.new_state(e_ref)
    z1 = orb.decoder_pass(st, ids[:2], 0, True)[0]
    st["kv_len"] = 2
    z2 = orb.decoder_pass(st, ids[2:], 2, True)[0]
    assert (torch.cat([z1, z2]) - ref).abs().max() < 5e-5


def test_processors_match_hf():
    from transformers.generation.logits_process import (SuppressTokensLogitsProcessor,
                                                        SuppressTokensAtBeginLogitsProcessor, ExponentialDecayLengthPenalty)
    g = torch.Generator().manual_seed(0)
    V = 300
    gp = GenParams(prompt=[7, 8, 9], eos_token_id=11, pad_token_id=11, suppress_tokens=[1, 5, 250],
                   begin_suppress_tokens=[220, 11], exp_decay=(4, 1.2))
    for cur_len in (3, 4, 7, 8, 20):
        scores = torch.randn(5, V, generator=g) * 3
        ids = torch.zeros(1, cur_len, dtype=torch.long)
        ref = ExponentialDecayLengthPenalty(gp.exp_decay, gp.eos_token_id, len(gp.prompt))(ids, scores)
        ref = SuppressTokensAtBeginLogitsProcessor(gp.begin_suppress_tokens, begin_index=len(gp.prompt))(ids, ref)
        ref = SuppressTokensLogitsProcessor(gp.suppress_tokens)(ids, ref)
        got = process_logits(scores, cur_len, gp)
        assert torch.equal(torch.isinf(got), torch.isinf(ref))
        fin = ~torch.isinf(ref)
        assert torch.equal(got[fin], ref[fin])


def test_language_detection_matches_hf(hf_tiny):
    """Oracle.detect_language == HF detect_language on the same weights (the product's detect_language is checked against the
    oracle on the GPU, tests/test_gpu_features.py)."""
    from transformers import GenerationConfig
    from transformers.modeling_outputs import BaseModelOutput
    cfg, sd, m = hf_tiny
    orc = Oracle(MedusaConfig.tiny_en("medusa_block", K=4), {**sd, **{k.replace("whisper_model.model.decoder.layers.3", "medusa_block"): v
                                                                      for k, v in sd.items() if "decoder.layers.3." in k}})
    lang_to_id = {f"<|l{i}|>": 50259 + i for i in range(12)}
    gc = GenerationConfig(decoder_start_token_id=50257, lang_to_id=lang_to_id, is_multilingual=True, task_to_id={"transcribe": 50359})
    picks = set()
    for clip in range(4):
        feats = torch.from_numpy(log_mel(synth.synth_clip(clip)))
        with torch.no_grad():
            e_ref = m.model.encoder(feats[None]).last_hidden_state
            want = int(m.detect_language(encoder_outputs=BaseModelOutput(last_hidden_state=e_ref), generation_config=gc)[0])
        got = orc.detect_language(e_ref[0], list(lang_to_id.values()), start_token=50257)
        assert got == want and got in lang_to_id.values()
        picks.add(got)
    assert len(picks) >= 1


@pytest.mark.parametrize("tag,heads,seed", [("microprompt", "base_head", 41), ("microblockprompt", "medusa_block", 42)])
def test_long_prompt_runs_match_reference(tag, heads, seed):
    """`prompt_ids` conditioning: decode runs started from decoder prompts of 7 / 25 / 42 tokens (previous-text token + text +
    init tokens), minted from the reference forward() (oracle/make_golden.py prompt) — the oracle reproduces tokens and accept
    lengths; the engine's chunked prompt pass is compared with the oracle on the GPU (tests/test_gpu_features.py)."""
    runs = np.load(f"{GOLD}/reference_prompt_runs.npz")
    cfg = MedusaConfig.micro(K=4, heads_type=heads, n_tgt=96)
    sd = synth.synth_state_dict(cfg, seed=seed)
    orc = Oracle(cfg, sd, sim="fp32")
    enc = orc.encode(torch.from_numpy(log_mel(clip_for(cfg, 2), cfg.num_mel_bins, cfg.n_mel_frames * 160)))
    for plen in (5, 23, 40):
        gp = golden_gen_params(cfg, ACCEPT_TYPICAL, 24)
        gp.begin_suppress_index = int(runs[f"{tag}_{plen}_begin_index"])       # = number of init tokens (reference model.py:1537)
        assert gp.begin_suppress_index == len(gp.prompt)
        gp.prompt = runs[f"{tag}_{plen}_prompt"].tolist()
        gp.max_length = min(len(gp.prompt) + 24, cfg.max_target_positions)
        r = orc.decode(enc, gp)
        assert r.ids == runs[f"{tag}_{plen}_ids"].tolist() and r.accept_lengths == runs[f"{tag}_{plen}_accepts"].tolist(), (tag, plen)


def test_offline_fp32_tables_match_the_live_oracle():
    """tests/golden/fp32_pinned_runs.npz (oracle/make_fp32_golden.py: the pinned fp32 mode run audio -> tokens offline, compared with the
    engine on the GPU box) cannot drift from the oracle: three tiny.en clips re-run live here in both acceptance modes, id for id, and the
    large-v2 table's bookkeeping (8 clips x 48 new tokens, emitted ids per iteration sum to the ids stored) checked."""
    from helpers import fp32_golden
    cfg = MedusaConfig.tiny_en(K=4)
    sd = synth.synth_state_dict(cfg, seed=0)
    g = fp32_golden("tiny", sd)
    seed, clip0, N, NEW = (int(x) for x in g["tiny_meta"])
    orc = Oracle(cfg, sd, sim="fp32")
    for mode, m in ((ACCEPT_GREEDY, "greedy"), (ACCEPT_TYPICAL, "typical")):
        gp = golden_gen_params(cfg, mode, NEW)
        for i in (0, 7, 15):
            ref = orc.transcribe(synth.synth_clip(clip0 + i, cfg.n_mel_frames * 160), gp)
            assert ref.ids == [int(t) for t in g[f"tiny_{m}_ids"][i][: int(g[f"tiny_{m}_len"][i])]], (m, i)
    assert [int(x) for x in g["large_meta"]] == [0, 300, 8, 48]
    for m in ("greedy", "typical"):
        P = int(g[f"large_{m}_prompt_len"])
        for i in range(8):
            n = int(g[f"large_{m}_len"][i])
            assert n - P >= 48 - 10 and int(g[f"large_{m}_emit"][i].sum()) == n - P
            assert (g[f"large_{m}_ids"][i][:n] >= 0).all() and (g[f"large_{m}_ids"][i][n:] == -1).all()
