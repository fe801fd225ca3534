"""Features either side of the hot path (SURVEY.md §8f / VERDICT r01 "missing"): prompt_ids conditioning with prompts longer
than one 16-row pass, long-form chunking, language detection, the data-parallel generate_sharded() entry point."""
import os
import sys

import numpy as np
import pytest
import torch

from helpers import MedusaConfig, synth, golden_gen_params, clip_for, check_tokens, ACCEPT_TYPICAL
from oracle.whisper_medusa_oracle import Oracle
from whisper_medusa import WhisperMedusaModel

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("heads", ["base_head", "medusa_block"])
@pytest.mark.parametrize("plen", [5, 16, 23, 40])
def test_long_prompts_match_the_oracle(gpu, heads, plen):
    """prompt_ids conditioning (reference signature model.py:1431, forwarded to HF at :1519-1529): decoder prompt =
    prompt_ids + init tokens.  Prompts longer than 16 tokens go through the layers in 16-row chunks (K/V only) before the
    first iteration; tokens must equal the oracle's, which simply feeds the whole prompt to its first base pass."""
    cfg = MedusaConfig.micro(K=4, heads_type=heads, n_tgt=96)
    sd = synth.synth_state_dict(cfg, seed=41)
    model = WhisperMedusaModel(cfg, sd, device=gpu, max_batch=2)
    orc = Oracle(cfg, sd, sim="bf16")
    feats = model.extract_features([clip_for(cfg, 3), clip_for(cfg, 4)[: cfg.n_mel_frames * 80]])
    pid = torch.tensor([cfg.vocab_size - 5] + [10 + (7 * i) % 900 for i in range(plen - 1)])
    out = model.generate(feats, prompt_ids=pid, max_new_tokens=20, exponential_decay_length_penalty=(6, 1.3))
    gp = model._gen_params(None, None, (6, 1.3), 20, None, None, False, None, None, None, None, pid)
    assert gp.prompt[:plen] == pid.tolist() and out[0, : len(gp.prompt)].tolist() == gp.prompt
    enc = model.engine.encoder_output(2)
    for b in range(2):
        got = out[b].tolist()
        ref = orc.decode(enc[b], gp)
        want = ref.ids[: ref.ids.index(gp.eos_token_id) + 1] if gp.eos_token_id in ref.ids[len(gp.prompt):] else ref.ids
        if got[: len(want)] != want:
            # (round 6: the run ends on an EOS that the exponential length penalty pushes across the arg-max — a decision that sits, by
            #  construction, on a crossing point: tie-aware like every decode-loop comparison, followed ties are counted in the parity report)
            check_tokens(orc, enc[b], gp, model.engine.tokens(b), label=f"long prompt {plen} {heads} b={b}")
        else:
            assert all(t == gp.pad_token_id for t in got[len(want):]), (b, got, want)
    model.engine.close()


def test_exponential_decay_start_may_be_a_float(gpu):
    """eval CLI call shape: --regulation-start is parsed with type=float"""
    cfg = MedusaConfig.micro(K=4)
    model = WhisperMedusaModel(cfg, synth.synth_state_dict(cfg, seed=11), device=gpu)
    feats = model.extract_features(clip_for(cfg, 0))
    a = model.generate(feats, max_new_tokens=16, exponential_decay_length_penalty=(6.0, 1.3))
    b = model.generate(feats, max_new_tokens=16, exponential_decay_length_penalty=(6, 1.3))
    assert torch.equal(a, b)
    model.engine.close()


def test_longform_chunking_equals_per_window_generation(gpu):
    """chunk_longform=True (the reference raises for > 30 s, model.py:1213-1214): every 30 s window decoded as an independent
    stream of one batch; a clip's output = prompt + the windows' generated ids in order."""
    cfg = MedusaConfig.micro(K=4)
    model = WhisperMedusaModel(cfg, synth.synth_state_dict(cfg, seed=11), device=gpu, max_batch=4)
    n = cfg.n_mel_frames * 160
    long_wav = np.concatenate([clip_for(cfg, 5), clip_for(cfg, 6), clip_for(cfg, 7)[: n // 2]])      # 2.5 windows
    F = cfg.n_mel_frames
    wins = [long_wav[i * n: (i + 1) * n] for i in range(3)]
    feats_w = model.extract_features(wins)                                                           # [3, 80, F] (last one padded by the front end)
    feats_long = torch.cat([feats_w[0], feats_w[1], feats_w[2][:, : F // 2]], dim=1)[None]           # [1, 80, 2.5 F]
    with pytest.raises(NotImplementedError, match="Longform"):
        model.generate(feats_long)
    out = model.generate(feats_long, chunk_longform=True, max_new_tokens=14)
    P = len(synth.default_prompt(cfg))
    eos, pad = cfg.eos_token_id, cfg.pad_token_id
    want = list(synth.default_prompt(cfg))
    for j in range(3):
        f = feats_w[j: j + 1].clone()
        if j == 2:
            f[..., F // 2:] = feats_long.amin()                      # the chunker pads the last window in the feature domain
        row = model.generate(f, max_new_tokens=14)[0, P:].tolist()
        for t in row:
            if t in (eos, pad):
                break
            want.append(t)
    got = out[0].tolist()
    assert got[: len(want)] == want and got[len(want)] == eos
    model.engine.close()


def test_language_detection_groups_clips(gpu):
    """language=None on a multilingual checkpoint: detect per clip from the base logits after <|startoftranscript|> (HF
    detect_language via _retrieve_init_tokens), then decode each language group with its own prompt."""
    cfg = MedusaConfig.micro(K=4)
    cfg.is_multilingual = True
    cfg.lang_to_id = {"<|en|>": 20, "<|de|>": 21, "<|fr|>": 22}
    cfg.task_to_id = {"transcribe": 30, "translate": 31}
    sd = synth.synth_state_dict(cfg, seed=43)
    model = WhisperMedusaModel(cfg, sd, device=gpu, max_batch=3)
    orc = Oracle(cfg, sd, sim="bf16")
    feats = model.extract_features([clip_for(cfg, i) for i in range(3)])
    langs = model.detect_language(feats)
    enc = model.engine.encoder_output(3)
    for b in range(3):
        want_id = orc.detect_language(enc[b], list(cfg.lang_to_id.values()))       # pinned to HF detect_language on the CPU
        want = next(k for k, v in cfg.lang_to_id.items() if v == want_id)
        z = orc.decoder_pass(orc.new_state(enc[b]), [cfg.decoder_start_token_id], 0, disable_medusa=True)[0, 0]
        top = sorted((float(z[i]) for i in cfg.lang_to_id.values()), reverse=True)
        assert langs[b] == want or top[0] - top[1] < 1e-3
    out = model.generate(feats, max_new_tokens=10)
    assert model.detected_languages == langs
    for b in range(3):
        assert out[b, 1].item() == cfg.lang_to_id[langs[b]]
        alone = model.generate(feats[b: b + 1], language=langs[b], max_new_tokens=10)
        assert out[b, : alone.shape[1]].tolist() == alone[0].tolist()
    model.engine.close()


def test_language_detection_of_a_single_language_batch_decodes_from_its_own_encoder_pass(gpu):
    """Every clip of one language: generate(language=None) decodes from the encoder pass detect_language() ran (no second wm_encode);
    tokens equal the explicit-language call's, at one stream and at four (one batched context)."""
    cfg = MedusaConfig.micro(K=4)
    cfg.is_multilingual = True
    cfg.lang_to_id = {"<|en|>": 20, "<|de|>": 21, "<|fr|>": 22}
    cfg.task_to_id = {"transcribe": 30, "translate": 31}
    sd = synth.synth_state_dict(cfg, seed=43)
    model = WhisperMedusaModel(cfg, sd, device=gpu, max_batch=4)
    one = model.extract_features([clip_for(cfg, 1)])
    for feats in (one, one.repeat(4, 1, 1)):
        encodes = []
        real = model.engine.encode
        model.engine.encode = lambda f, real=real: (encodes.append(f.shape[0]), real(f))[1]
        try:
            out = model.generate(feats, max_new_tokens=12)
        finally:
            model.engine.encode = real
        assert encodes == [feats.shape[0]] and len(set(model.detected_languages)) == 1
        want = model.generate(feats, language=model.detected_languages[0], max_new_tokens=12)
        assert out.tolist() == want.tolist()
    model.engine.close()


def _shard_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as td
    td.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)                               # both ranks share the one GPU of the test box
    cfg = MedusaConfig.micro(K=4)
    model = WhisperMedusaModel(cfg, synth.synth_state_dict(cfg, seed=11), device=dev, max_batch=5)
    feats = model.extract_features([clip_for(cfg, i) for i in range(5)])
    out = model.generate_sharded(feats, max_new_tokens=12)
    ref = model.generate(feats, max_new_tokens=12)
    q.put((rank, torch.equal(out, ref), out.shape))
    td.destroy_process_group()


def test_generate_sharded_two_ranks_share_the_gpu(gpu):
    """The data-parallel entry point end to end (shard_streams -> generate -> gather_token_lists) with two gloo ranks on the
    one GPU of the test box: every rank ends with all 5 sequences, identical to a single-process generate()."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 1000
    ps = [ctx.Process(target=_shard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=300) for _ in ps]
    for p in ps:
        p.join(timeout=60)
    assert sorted(r[0] for r in res) == [0, 1] and all(r[1] for r in res), res


def test_packed_fp8_export_loads_and_decodes_identically(gpu, tmp_path):
    """save_packed -> from_packed (fp8 decoder weights + fp8 MFMA encoder): the re-loaded engine-ready blob gives the same tokens
    as the model it was exported from."""
    cfg = MedusaConfig.micro(K=4)
    sd = synth.synth_state_dict(cfg, seed=31)
    a = WhisperMedusaModel(cfg, sd, device=gpu, max_batch=2, dec_weight_fp8=True, enc_fp8=True)
    a.save_packed(str(tmp_path / "packed"))
    b = WhisperMedusaModel.from_packed(str(tmp_path / "packed"), device=gpu, max_batch=2)
    n = cfg.n_mel_frames * 160
    feats = a.extract_features([synth.synth_clip(i, n) for i in range(2)])
    ia = a.generate(feats, max_new_tokens=24)
    ib = b.generate(feats, max_new_tokens=24)
    assert torch.equal(ia, ib) and ia.shape[1] > 8
    a.engine.close(); b.engine.close()


def test_forward_with_encoder_outputs_and_cache_handles(gpu):
    """forward(encoder_outputs=, past_key_values=, use_cache=) (reference model.py:1223-1243): a foreign last-hidden-state tensor goes
    through wm_set_encoder_output (stored bf16, cross K/V projected from it) and must give the logits of the input_features call;
    a prompt fed in two cached calls must give the logits of the one-call pass; the oracle agrees on the cached continuation."""
    from oracle.whisper_medusa_oracle import Oracle, log_mel
    cfg = MedusaConfig.tiny_en(K=4)
    sd = synth.synth_state_dict(cfg, seed=0)
    model = WhisperMedusaModel(cfg, sd, device=gpu, max_batch=2)
    n = cfg.n_mel_frames * 160
    feats = torch.from_numpy(np.stack([log_mel(synth.synth_clip(i, n), cfg.num_mel_bins, n) for i in range(2)])).to(gpu)
    ids = torch.tensor([synth.default_prompt(cfg) + [11, 12, 13], synth.default_prompt(cfg) + [21, 22, 23]])
    P = ids.shape[1]
    o_all = model.forward(input_features=feats, decoder_input_ids=ids, use_cache=True)
    assert o_all.logits.shape == (cfg.medusa_num_heads + 1, 2, P, cfg.vocab_size) and o_all.past_key_values.get_seq_length() == P
    hid = o_all.encoder_last_hidden_state.clone()                 # [2, S, d] fp32 of the bf16-stored encoder output
    # (1) a foreign tensor replaces the encoder pass: same stored values -> identical cross K/V -> identical logits
    model.forward(input_features=torch.zeros_like(feats), decoder_input_ids=ids[:, :1])      # clobber the resident encoder pass
    o_ext = model.forward(encoder_outputs=(hid,), decoder_input_ids=ids)
    assert torch.equal(o_ext.logits, o_all.logits)
    assert (o_ext.encoder_last_hidden_state - hid).abs().max() == 0
    # (2) cached continuation: first P - 2 tokens, then the last two behind them; our own encoder handle is free
    o1 = model.forward(encoder_outputs=o_ext.encoder_outputs, decoder_input_ids=ids[:, : P - 2], use_cache=True)
    o2 = model.forward(encoder_outputs=o_ext.encoder_outputs, decoder_input_ids=ids[:, P - 2:], past_key_values=o1.past_key_values)
    assert o2.past_key_values.get_seq_length() == P
    d = (o2.logits - o_all.logits[:, :, P - 2:]).abs().max()
    assert d <= 1e-4 * float(o_all.logits.abs().max()), float(d)   # same rows through a 2-row pass instead of a P-row pass: fp32 order only
    # (3) the oracle on the engine's encoder output agrees with the cached continuation (logit tolerance of test_forward_logits_all_heads)
    orc = Oracle(cfg, sd, sim="bf16")
    st = orc.new_state(hid[1])
    ref = orc.decoder_pass(st, ids[1].tolist(), 0, disable_medusa=False)            # [K+1, P, V]
    scale = float(ref.abs().max())
    assert (o2.logits[:, 1] - ref[:, P - 2:]).abs().max() <= 2e-3 * scale
    # enc_fp8 contexts cannot take a foreign encoder output (their cross-K/V projection reads the fp8 LayerNorm output)
    m8 = WhisperMedusaModel(cfg, sd, device=gpu, max_batch=1, enc_fp8=True)
    with pytest.raises(ValueError, match="enc_fp8"):
        m8.forward(encoder_outputs=(hid[:1],), decoder_input_ids=ids[:1])
    m8.engine.close()
    model.engine.close()


@pytest.mark.parametrize("heads", ["base_head", "medusa_block"])
def test_arbitrary_logits_processors_match_the_fused_loop(gpu, heads):
    """generate(logits_processor=[any Python callable]) (reference model.py:1106-1116 hands any list to HF): the host path — the
    reference's loop structure with every decoder pass on the engine (wm_forward_logits) and processors / candidates / posterior in
    torch — must emit the tokens of the fused device loop: with a processor that changes nothing, and with a token ban that the fused loop
    expresses as its static suppress list.  tiny.en shape, two clips, both acceptance modes."""
    from transformers.generation.logits_process import LogitsProcessor, LogitsProcessorList
    cfg = MedusaConfig.tiny_en(heads, K=4)
    sd = synth.synth_state_dict(cfg, seed=4)
    model = WhisperMedusaModel(cfg, sd, device=gpu, max_batch=2)
    n = cfg.n_mel_frames * 160
    feats = model.extract_features(np.stack([synth.synth_clip(30 + i, n) for i in range(2)]))

    class Identity(LogitsProcessor):
        calls = 0
        def __call__(self, input_ids, scores):
            Identity.calls += 1
            return scores

    class Ban(LogitsProcessor):
        def __init__(self, toks): self.toks = list(toks)
        def __call__(self, input_ids, scores):
            scores = scores.clone(); scores[:, self.toks] = -float("inf"); return scores

    kw = dict(max_new_tokens=28, exponential_decay_length_penalty=(6, 1.3))
    for temperature in (None, 0.0):
        fused = model.generate(feats, temperature=temperature, **kw)
        st_fused = dict(model.last_stats)
        host = model.generate(feats, temperature=temperature, logits_processor=LogitsProcessorList([Identity()]), **kw)
        assert model.last_stats["host_processors"] == 1 and Identity.calls >= 4
        assert host.tolist() == fused.tolist(), (heads, temperature)
        assert model.last_stats["accept_hist"][: cfg.medusa_num_heads + 1] == st_fused["accept_hist"][: cfg.medusa_num_heads + 1]
    # a long decoder prompt (prompt_ids conditioning: 21 + 2 tokens, more than one 16-row pass) and a host-side stopping criterion on top
    from transformers.generation.stopping_criteria import StoppingCriteria
    pid = torch.tensor([cfg.prev_sot_token_id if cfg.prev_sot_token_id < cfg.vocab_size else 9] + list(range(100, 120)))
    fused_p = model.generate(feats[:1], prompt_ids=pid, **kw)
    host_p = model.generate(feats[:1], prompt_ids=pid, logits_processor=[Identity()], **kw)
    assert host_p.tolist() == fused_p.tolist() and len(model._last_prompt) == 21 + len(synth.default_prompt(cfg))

    class StopAfter(StoppingCriteria):
        def __init__(self, n): self.n = n
        def __call__(self, input_ids, scores, **kw_):
            return torch.tensor([input_ids.shape[-1] >= self.n])

    Pp = len(model._last_prompt)
    cut = model.generate(feats[:1], prompt_ids=pid, logits_processor=[Identity()], stopping_criteria=[StopAfter(Pp + 6)], **kw)
    n_cut = cut.shape[1]                                             # one stream: the tensor is exactly its sequence
    assert n_cut <= fused_p.shape[1] and cut[0].tolist() == fused_p[0, :n_cut].tolist()
    assert (Pp + 6 <= n_cut < Pp + 6 + cfg.medusa_num_heads + 1) or cfg.eos_token_id in cut[0, Pp:].tolist()     # stopped at the first iteration that reached the length
    P = len(synth.default_prompt(cfg))
    ban = sorted(set(fused[0, P:].tolist()) - {cfg.eos_token_id, cfg.pad_token_id})[:3]
    want = model.generate(feats, suppress_tokens=sorted(set(cfg.suppress_tokens or []) | set(ban)), **kw)
    got = model.generate(feats, logits_processor=[Ban(ban)], **kw)
    assert got.tolist() == want.tolist() and not set(ban) & set(got[:, P:].flatten().tolist())
    model.engine.close()
