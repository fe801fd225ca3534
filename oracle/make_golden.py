"""Mint golden vectors for the oracle FROM THE REFERENCE'S OWN CODE.  TEST INFRASTRUCTURE ONLY.

Runs only in the build container (needs ``/root/reference``; the GPU box has no copy):

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py

What is executed from the reference (nothing is copied):
  * ``whisper_medusa/models/medusa_utils.py`` imported by path: ``generate_medusa_buffers``,
    ``generate_candidates``, ``evaluate_posterior`` (both branches);
  * ``WhisperMedusaModel.forward()`` (Medusa-Linear AND Medusa-Block) under the import stubs of SURVEY.md
    Appendix B, driven cache-free (full ``decoder_input_ids`` each call) — mathematically the
    cached path because of causal masking.  Medusa-Block needs two more shims for transformers 5.x
    (``build_ref``): ``_forward_medusa_block`` (model.py:1361-1380) has no branch for the ``EncoderDecoderCache``
    HF now always returns, so the name it tests against is swapped for a pass-through class, and
    ``WhisperDecoderLayer`` now returns a tensor where the reference indexes a tuple (model.py:1414);
  * HF's own ``SuppressTokensLogitsProcessor`` / ``SuppressTokensAtBeginLogitsProcessor`` /
    ``ExponentialDecayLengthPenalty`` as the processor list the reference builds
    (model.py:1168-1207, :1106-1116).
Only the ~25 lines of loop glue of model.py:634-793 are restated here (``ref_medusa_loop``).
Outputs: ``tests/golden/*.npz`` (token ids, accept lengths, a few logits) — weights are NOT
stored; tests regenerate them from the seed with ``whisper_medusa.synth``.
"""
import importlib.util
import os
import sys
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "whisper-medusa_amd"))
sys.path.insert(0, ROOT)
REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")

from whisper_medusa.config import MedusaConfig, GenParams, ACCEPT_TYPICAL, ACCEPT_GREEDY  # noqa: E402
from whisper_medusa import synth  # noqa: E402
from oracle.whisper_medusa_oracle import Oracle, log_mel  # noqa: E402


def load_ref_medusa_utils():
    spec = importlib.util.spec_from_file_location(
        "ref_medusa_utils", os.path.join(REF, "whisper_medusa/models/medusa_utils.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _import_reference_package():
    """Import the reference's ``whisper_medusa`` package under an alias so that it does not
    collide with our own drop-in package of the same name."""
    import transformers.generation.utils as gu
    if not hasattr(gu, "NEED_SETUP_CACHE_CLASSES_MAPPING"):
        gu.NEED_SETUP_CACHE_CLASSES_MAPPING = {}
    class _Stub(types.ModuleType):          # any attribute of a missing optional dependency -> inert object
        def __getattr__(self, item):
            if item.startswith("__"):
                raise AttributeError(item)
            return type(item, (), {"__init__": lambda self, *a, **k: None, "__call__": lambda self, *a, **k: None})

    for name in ("wandb", "jiwer", "jiwer.transforms", "torchaudio", "torchaudio.transforms"):
        if name not in sys.modules:
            sys.modules[name] = _Stub(name)
    ours = {k: v for k, v in sys.modules.items() if k == "whisper_medusa" or k.startswith("whisper_medusa.")}
    for k in ours:
        del sys.modules[k]
    sys.path.insert(0, REF)
    try:
        import whisper_medusa.models.model as ref_model            # the reference's model.py
        import whisper_medusa.utils.config_and_args as ref_cfg
    finally:
        sys.path.remove(REF)
        refmods = {k: v for k, v in sys.modules.items() if k == "whisper_medusa" or k.startswith("whisper_medusa.")}
        for k in refmods:
            del sys.modules[k]
        sys.modules.update(ours)
    return ref_model, ref_cfg


def build_ref(cfg: MedusaConfig, sd):
    """The reference's own ``WhisperMedusaModel`` (either head type) with the seeded checkpoint loaded."""
    from transformers import WhisperConfig
    ref_model, ref_cfg = _import_reference_package()
    hf_cfg = WhisperConfig(
        d_model=cfg.d_model, encoder_layers=cfg.encoder_layers, decoder_layers=cfg.decoder_layers,
        encoder_attention_heads=cfg.encoder_attention_heads, decoder_attention_heads=cfg.decoder_attention_heads,
        encoder_ffn_dim=cfg.encoder_ffn_dim, decoder_ffn_dim=cfg.decoder_ffn_dim, vocab_size=cfg.vocab_size,
        num_mel_bins=cfg.num_mel_bins, max_source_positions=cfg.max_source_positions,
        max_target_positions=cfg.max_target_positions, eos_token_id=cfg.eos_token_id,
        pad_token_id=cfg.pad_token_id, bos_token_id=cfg.eos_token_id,
        decoder_start_token_id=cfg.decoder_start_token_id)
    ref_cfg.AutoConfig.from_pretrained = staticmethod(lambda name, **kw: hf_cfg)
    W2M = ref_model.Whisper2MedusaHeadsConditionalGeneration
    W2M.from_pretrained = classmethod(lambda cls, name, **kw: cls(hf_cfg))
    mcfg = ref_cfg.MedusaConfig(
        medusa_num_heads=cfg.medusa_num_heads, medusa_num_layers=1, medusa_hidden_size=cfg.d_model,
        whisper_model_name="dummy", medusa_choices=[1] * (cfg.medusa_num_heads + 1),
        medusa_heads_type=cfg.medusa_heads_type)
    model = ref_model.WhisperMedusaModel(mcfg).eval()
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("proj_out" in m or "embed_positions" in m for m in missing), missing
    if cfg.is_block:
        real_edc = ref_model.EncoderDecoderCache

        class _PassThroughCache:            # `isinstance(hf_cache, _PassThroughCache)` is False -> model.py:1365-1370 runs and
            def __new__(cls, self_cache, cross_cache=None):     # "wraps" the cache HF already built: hand it through unchanged
                return self_cache

            @staticmethod
            def from_legacy_cache(x):
                return real_edc.from_legacy_cache(x)

        ref_model.EncoderDecoderCache = _PassThroughCache
        layer_forward = model.medusa_block.forward

        def tuple_forward(*a, **k):         # transformers 4.49 returned (hidden_states, ...); model.py:1414 takes [0]
            o = layer_forward(*a, **k)
            return o if isinstance(o, tuple) else (o,)

        model.medusa_block.forward = tuple_forward
    return model


def hf_processors(gp: GenParams):
    from transformers.generation.logits_process import (
        LogitsProcessorList, SuppressTokensLogitsProcessor, SuppressTokensAtBeginLogitsProcessor,
        ExponentialDecayLengthPenalty)
    procs = []
    if gp.exp_decay is not None:                     # HF default list first (model.py:1106-1116)
        procs.append(ExponentialDecayLengthPenalty(gp.exp_decay, gp.eos_token_id, len(gp.prompt)))
    if gp.begin_suppress_tokens:                     # model.py:1188-1199 prepends begin-suppress ...
        procs.append(SuppressTokensAtBeginLogitsProcessor(gp.begin_suppress_tokens, begin_index=gp.begin_index))
    if gp.suppress_tokens:                           # ... in front of suppress (model.py:1177-1186)
        procs.append(SuppressTokensLogitsProcessor(gp.suppress_tokens))
    return LogitsProcessorList(procs)


@torch.no_grad()
def ref_medusa_loop(model, mu, enc, gp: GenParams, K: int, use_cache: bool = False):
    """Semi-live end-to-end run: reference forward() + reference candidates/posterior; the glue
    below restates model.py:634-793 (SURVEY.md §8c 'semi-live oracle').  ``use_cache=True`` (Medusa-Block: model.py:1364
    needs a cache object to exist) still is cache-free from the outside: no past is passed in, HF builds a fresh cache per call."""
    procs = hf_processors(gp)
    buffers = mu.generate_medusa_buffers([1] * (K + 1), device="cpu")
    ids = torch.tensor([gp.prompt], dtype=torch.long)
    accepts, first_logits = [], None
    while True:
        L = ids.shape[1]
        out = model(encoder_outputs=(enc[None],), decoder_input_ids=ids, use_cache=use_cache, return_dict=True)
        logits = out.logits[:, :, -1:, :]                                   # [K+1, 1, 1, V] last row only
        if first_logits is None:
            first_logits = logits[:, 0, 0].clone()
        orig = procs(ids, logits[0].squeeze(0)).unsqueeze(0)                # model.py:653-655
        med = procs(ids, logits[1:].reshape(K, -1)).reshape(K, 1, 1, -1)    # model.py:656-665
        cands, tree_cands = mu.generate_candidates(med, orig, [1] * K, buffers["tree_indices"])
        full = torch.cat([ids, tree_cands], dim=1)                          # cache-free verify pass
        vout = model(encoder_outputs=(enc[None],), decoder_input_ids=full, use_cache=use_cache,
                     return_dict=True, disable_medusa=True)
        tree_logits = vout.logits[0][0, L:][buffers["retrieve_indices"]]    # medusa_utils.py:518-521
        proc = procs(ids, tree_logits.reshape(K + 1, -1)).reshape(1, K + 1, -1)
        best, a = mu.evaluate_posterior(proc, cands, gp.temperature, gp.posterior_threshold, gp.posterior_alpha)
        a = int(a)
        nxt = cands[None, best, : a + 1]
        if a == 0:                                                          # medusa_utils.py:636-641
            nxt = torch.cat([nxt, torch.argmax(proc[None, best, 0], dim=-1).unsqueeze(0)], dim=-1)
        ids = torch.cat([ids, nxt], dim=-1)
        accepts.append(a)
        L = ids.shape[1]
        if (nxt == gp.eos_token_id).any() or L >= gp.max_length or L + K >= gp.hard_max_length:
            break
    ids = ids[0].tolist()
    if gp.eos_token_id in ids:                                              # post-EOS overwrite, model.py:798-810
        j = ids.index(gp.eos_token_id)
        ids = ids[: j + 1] + [gp.eos_token_id] * (len(ids) - j - 1)
    return ids, accepts, first_logits


def golden_medusa_utils(mu):
    """Known-answer vectors from the reference's medusa_utils."""
    out = {}
    b = mu.generate_medusa_buffers([1] * 11, device="cpu")
    out["buf11_tree_indices"] = b["tree_indices"].numpy()
    out["buf11_retrieve_indices"] = b["retrieve_indices"].numpy()
    out["buf11_position_ids"] = b["medusa_position_ids"].numpy()
    b = mu.generate_medusa_buffers([1, 3, 2], device="cpu")
    out["buf132_retrieve_indices"] = b["retrieve_indices"].numpy()
    g = torch.Generator().manual_seed(1234)
    V, K = 257, 6
    cases_logits, cases_cand, acc_typ, acc_greedy = [], [], [], []
    for case in range(24):
        sharp = [0.5, 2.0, 6.0, 12.0][case % 4]
        logits = torch.randn(K + 1, V, generator=g) * sharp
        cand = torch.randint(0, V, (K + 1,), generator=g)
        n_agree = case % (K + 1)
        am = torch.argmax(logits[:-1], dim=-1)
        cand[1: 1 + n_agree] = am[:n_agree]                      # first n_agree candidates are the argmax
        for temp, store in ((1.0, acc_typ), (0.0, acc_greedy)):
            best, a = mu.evaluate_posterior(logits[None], cand[None], temp, 0.09, 0.3)
            assert int(best) == 0
            store.append(int(a))
        cases_logits.append(logits.numpy())
        cases_cand.append(cand.numpy())
    out["post_logits"] = np.stack(cases_logits).astype(np.float32)
    out["post_cand"] = np.stack(cases_cand)
    out["post_accept_typical"] = np.array(acc_typ)
    out["post_accept_greedy"] = np.array(acc_greedy)
    # candidates: argmax of base row + top-1 of every head row
    base = torch.randn(1, 1, V, generator=g)
    med = torch.randn(K, 1, 1, V, generator=g)
    b = mu.generate_medusa_buffers([1] * (K + 1), device="cpu")
    c, tc = mu.generate_candidates(med, base, [1] * K, b["tree_indices"])
    out["cand_base"], out["cand_med"] = base.numpy(), med.numpy()
    out["cand_out"], out["cand_tree_out"] = c.numpy(), tc.numpy()
    return out


TREE_CHOICES = ([1, 3, 2], [1, 2, 2, 1], [1, 2, 1, 2, 1], [1, 4, 2], [1, 2, 2, 2], [1, 1, 3, 1])


def golden_tree_utils(mu):
    """Known-answer vectors for candidate trees with top-k > 1: buffers, candidates and the multi-path posterior
    (medusa_utils.py:305-421, :424-458, :526-588), straight from the reference functions."""
    out = {}
    for ch in TREE_CHOICES:
        tag = "buf" + "".join(str(x) for x in ch)
        b = mu.generate_medusa_buffers(ch, device="cpu")
        out[tag + "_tree_indices"] = b["tree_indices"].numpy()
        out[tag + "_position_ids"] = b["medusa_position_ids"].numpy()
        out[tag + "_retrieve_indices"] = b["retrieve_indices"].numpy()
        out[tag + "_attn_mask"] = b["medusa_attn_mask"][0, 0].numpy().astype(np.uint8)
    g = torch.Generator().manual_seed(4321)
    V = 193
    for ci, ch in enumerate(([1, 3, 2], [1, 2, 2, 1], [1, 2, 2, 2])):
        K = len(ch) - 1
        b = mu.generate_medusa_buffers(ch, device="cpu")
        n_paths = b["retrieve_indices"].shape[0]
        tag = f"tree{ci}"
        out[tag + "_choices"] = np.array(ch)
        base = torch.randn(1, 1, V, generator=g) * 3
        med = torch.randn(K, 1, 1, V, generator=g) * 3
        c, tc = mu.generate_candidates(med, base, ch[1:], b["tree_indices"])
        out[tag + "_cand_base"], out[tag + "_cand_med"] = base.numpy(), med.numpy()
        out[tag + "_cand_out"], out[tag + "_cand_tree_out"] = c.numpy(), tc.numpy()
        L, C, BT, AT, BG, AG = [], [], [], [], [], []
        for case in range(32):
            sharp = [1.0, 3.0, 6.0, 12.0][case % 4]
            node_logits = torch.randn(b["tree_indices"].shape[0], V, generator=g) * sharp
            flat = [torch.randint(0, V, (1,), generator=g)] + [torch.randperm(V, generator=g)[: ch[k]] for k in range(1, K + 1)]
            cand = torch.cartesian_prod(*flat)
            # make some nodes' tokens the argmax of their parent's row so that accept lengths vary across paths
            for pth in range(n_paths):
                for i in range(1, K + 1):
                    if float(torch.rand(1, generator=g)) < [0.35, 0.55, 0.75][case % 3]:
                        par, tok = int(b["retrieve_indices"][pth, i - 1]), int(cand[pth, i])
                        node_logits[par, tok] = node_logits[par].max() + (0.5 + 0.25 * ((case + pth) % 4))
            logits = node_logits[b["retrieve_indices"]]                      # [n_paths, K+1, V]
            for temp, bs, as_ in ((1.0, BT, AT), (0.0, BG, AG)):
                best, a = mu.evaluate_posterior(logits, cand, temp, 0.09, 0.3)
                bs.append(int(best)); as_.append(int(a))
            L.append(node_logits.numpy()); C.append(cand.numpy())
        out[tag + "_post_node_logits"] = np.stack(L).astype(np.float32)
        out[tag + "_post_cand"] = np.stack(C)
        out[tag + "_post_best_typical"], out[tag + "_post_accept_typical"] = np.array(BT), np.array(AT)
        out[tag + "_post_best_greedy"], out[tag + "_post_accept_greedy"] = np.array(BG), np.array(AG)
        print(f"  {tag} {ch}: typical accepts {AT[:12]} best {BT[:12]} | greedy accepts {AG[:12]}")
    return out


@torch.no_grad()
def ref_medusa_tree_loop(model, mu, enc, gp: GenParams, choices, use_cache: bool = False):
    """Semi-live run over a candidate tree with top-k > 1: the reference's buffers / generate_candidates / evaluate_posterior,
    and for the verify logits the reference forward() along EVERY path (cache-free: cat(ids, path tokens) under the ordinary
    causal mask) — what tree_decoding (medusa_utils.py:494-521) computes per node when its attention mask is applied: a node
    attends to the history and to its own ancestors, i.e. exactly the tokens in front of it on its path."""
    K = len(choices) - 1
    procs = hf_processors(gp)
    buffers = mu.generate_medusa_buffers(choices, device="cpu")
    ids = torch.tensor([gp.prompt], dtype=torch.long)
    accepts, bests = [], []
    while True:
        L = ids.shape[1]
        out = model(encoder_outputs=(enc[None],), decoder_input_ids=ids, use_cache=use_cache, return_dict=True)
        logits = out.logits[:, :, -1:, :]
        orig = procs(ids, logits[0].squeeze(0)).unsqueeze(0)
        med = procs(ids, logits[1:].reshape(K, -1)).reshape(K, 1, 1, -1)
        cands, tree_cands = mu.generate_candidates(med, orig, choices[1:], buffers["tree_indices"])
        assert torch.equal(tree_cands[0][buffers["retrieve_indices"]], cands)
        rows = []
        for pth in range(cands.shape[0]):
            full = torch.cat([ids, cands[pth][None]], dim=1)
            vout = model(encoder_outputs=(enc[None],), decoder_input_ids=full, use_cache=use_cache, return_dict=True, disable_medusa=True)
            rows.append(procs(ids, vout.logits[0][0, L:]))
        proc = torch.stack(rows)                                                # [n_paths, K+1, V]
        best, a = mu.evaluate_posterior(proc, cands, gp.temperature, gp.posterior_threshold, gp.posterior_alpha)
        best, a = int(best), int(a)
        nxt = cands[None, best, : a + 1]
        if a == 0:
            nxt = torch.cat([nxt, torch.argmax(proc[None, best, 0], dim=-1).unsqueeze(0)], dim=-1)
        ids = torch.cat([ids, nxt], dim=-1)
        accepts.append(a); bests.append(best)
        L = ids.shape[1]
        if (nxt == gp.eos_token_id).any() or L >= gp.max_length or L + K >= gp.hard_max_length:
            break
    ids = ids[0].tolist()
    if gp.eos_token_id in ids:
        j = ids.index(gp.eos_token_id)
        ids = ids[: j + 1] + [gp.eos_token_id] * (len(ids) - j - 1)
    return ids, accepts, bests


def golden_tree_model(tag, cfg, seed, mu, choices, max_new):
    sd = synth.synth_state_dict(cfg, seed=seed)
    model = build_ref(cfg, sd)
    orc = Oracle(cfg, sd, sim="fp32")
    wav = synth.synth_clip(0, n_samples=cfg.n_mel_frames * 160)
    feats = torch.from_numpy(log_mel(wav, cfg.num_mel_bins, cfg.n_mel_frames * 160))
    with torch.no_grad():
        enc = model.whisper_model.model.encoder(feats[None]).last_hidden_state[0]
    out = {f"{tag}_choices": np.array(choices)}
    for mode, mname in ((ACCEPT_TYPICAL, "typical"), (ACCEPT_GREEDY, "greedy")):
        gp = gen_params_for(cfg, mode, max_new, suppress_eos=True)
        ids, accepts, bests = ref_medusa_tree_loop(model, mu, enc, gp, choices, use_cache=cfg.is_block)
        key = f"{tag}_{mname}"
        out[key + "_ids"], out[key + "_accepts"], out[key + "_best"] = np.array(ids), np.array(accepts), np.array(bests)
        r = orc.decode_tree(enc, gp, choices)
        assert r.ids[: len(ids)] == ids and r.accept_lengths == accepts, (key, r.ids, ids, r.accept_lengths, accepts)
        print(f"  {key}: {len(ids) - len(gp.prompt)} tokens, accepts {accepts}, best paths {bests}")
    return out


def gen_params_for(cfg, mode, max_new, suppress_eos=True, exp_decay=(6, 1.3)):
    prompt = synth.default_prompt(cfg)
    sup = [cfg.eos_token_id] if suppress_eos else []
    return GenParams(prompt=prompt, eos_token_id=cfg.eos_token_id, pad_token_id=cfg.pad_token_id,
                     suppress_tokens=sorted(set(sup + [3, 5])), begin_suppress_tokens=list(cfg.begin_suppress_tokens),
                     max_length=min(len(prompt) + max_new, cfg.max_target_positions),
                     hard_max_length=cfg.max_length, exp_decay=exp_decay,
                     accept_mode=mode, temperature=1.0 if mode == ACCEPT_TYPICAL else 0.0)


def golden_model(tag, cfg, seed, mu, max_new):
    """Reference forward()/loop outputs for one seeded checkpoint (Linear or Block heads)."""
    sd = synth.synth_state_dict(cfg, seed=seed)
    model = build_ref(cfg, sd)
    orc = Oracle(cfg, sd, sim="fp32")
    wav = synth.synth_clip(0, n_samples=cfg.n_mel_frames * 160)
    feats = torch.from_numpy(log_mel(wav, cfg.num_mel_bins, cfg.n_mel_frames * 160))
    with torch.no_grad():
        enc = model.whisper_model.model.encoder(feats[None]).last_hidden_state[0]      # HF encoder, reference weights
    out = {f"{tag}_enc_probe": enc[::max(1, enc.shape[0] // 8), :16].numpy()}
    K = cfg.medusa_num_heads
    for mode, mname in ((ACCEPT_TYPICAL, "typical"), (ACCEPT_GREEDY, "greedy")):
        for eos_free, ename in ((True, "noeos"), (False, "eos")):
            gp = gen_params_for(cfg, mode, max_new, suppress_eos=eos_free)
            ids, accepts, first_logits = ref_medusa_loop(model, mu, enc, gp, K, use_cache=cfg.is_block)
            key = f"{tag}_{mname}_{ename}"
            out[key + "_ids"] = np.array(ids)
            out[key + "_accepts"] = np.array(accepts)
            if mode == ACCEPT_TYPICAL and eos_free:
                out[f"{tag}_first_logits_probe"] = first_logits[:, :64].numpy()
            # self-check: oracle must reproduce the reference run token for token
            r = orc.decode(enc, gp)
            assert r.ids[: len(ids)] == ids and r.accept_lengths == accepts, (key, r.ids, ids, r.accept_lengths, accepts)
            print(f"  {key}: {len(ids) - len(gp.prompt)} tokens, accepts {accepts}")
    return out


def main_tree():
    """`python oracle/make_golden.py tree`: the candidate-tree fixtures only (the chain fixtures stay as minted)."""
    os.makedirs(GOLD, exist_ok=True)
    mu = load_ref_medusa_utils()
    np.savez_compressed(os.path.join(GOLD, "medusa_tree_kat.npz"), **golden_tree_utils(mu))
    print("candidate-tree known-answer vectors written")
    out = {}
    out.update(golden_tree_model("micro1221", MedusaConfig.micro(K=3), 21, mu, [1, 2, 2, 1], max_new=36))
    out.update(golden_tree_model("micro132", MedusaConfig.micro(K=2), 22, mu, [1, 3, 2], max_new=36))
    out.update(golden_tree_model("micro1222block", MedusaConfig.micro(K=3, heads_type="medusa_block"), 23, mu, [1, 2, 2, 2], max_new=30))
    # trees of more than 16 nodes (several 16-row query tiles per stream in the engine): 31 nodes / 16 paths, and the K = 10 shape the shipped
    # checkpoints have with top-2 on the first two heads (39 nodes / 4 paths)
    out.update(golden_tree_model("micro12222", MedusaConfig.micro(K=4), 25, mu, [1, 2, 2, 2, 2], max_new=30))
    out.update(golden_tree_model("micro10_122", MedusaConfig.micro(K=10, d_model=128, layers=2), 26, mu, [1, 2, 2] + [1] * 8, max_new=36))
    np.savez_compressed(os.path.join(GOLD, "reference_tree_runs.npz"), **out)
    print("reference tree-loop vectors written")


def main_prompt():
    """`python oracle/make_golden.py prompt`: decode runs conditioned on a long decoder prompt (`prompt_ids`, model.py:1519-1529:
    <|startofprev|> + previous text + the forced init tokens) through the reference forward() + candidates / posterior."""
    os.makedirs(GOLD, exist_ok=True)
    mu = load_ref_medusa_utils()
    out = {}
    for tag, cfg, seed in (("microprompt", MedusaConfig.micro(K=4, n_tgt=96), 41),
                           ("microblockprompt", MedusaConfig.micro(K=4, heads_type="medusa_block", n_tgt=96), 42)):
        sd = synth.synth_state_dict(cfg, seed=seed)
        model = build_ref(cfg, sd)
        orc = Oracle(cfg, sd, sim="fp32")
        wav = synth.synth_clip(2, n_samples=cfg.n_mel_frames * 160)
        feats = torch.from_numpy(log_mel(wav, cfg.num_mel_bins, cfg.n_mel_frames * 160))
        with torch.no_grad():
            enc = model.whisper_model.model.encoder(feats[None]).last_hidden_state[0]
        for plen in (5, 23, 40):
            gp = gen_params_for(cfg, ACCEPT_TYPICAL, 24)
            prev = [cfg.vocab_size - 5] + [10 + (7 * i) % 900 for i in range(plen - 1)]
            # the reference's generate() builds the processors with begin_index = init_tokens.shape[1] (model.py:1537, 1551) — the
            # init tokens WITHOUT the prepended prompt_ids — and set_begin_index() repeats it (:1640-1644): the begin suppression
            # therefore never fires under prompt conditioning.  The processors of this run are built the same way.
            gp.begin_suppress_index = len(gp.prompt)
            gp.prompt = prev + list(gp.prompt)
            gp.max_length = min(len(gp.prompt) + 24, cfg.max_target_positions)
            ids, accepts, _ = ref_medusa_loop(model, mu, enc, gp, cfg.medusa_num_heads, use_cache=cfg.is_block)
            r = orc.decode(enc, gp)
            assert r.ids[: len(ids)] == ids and r.accept_lengths == accepts, (tag, plen)
            out[f"{tag}_{plen}_prompt"] = np.array(gp.prompt)
            out[f"{tag}_{plen}_ids"] = np.array(ids)
            out[f"{tag}_{plen}_accepts"] = np.array(accepts)
            out[f"{tag}_{plen}_begin_index"] = np.array(gp.begin_index)
            print(f"  {tag} prompt {len(gp.prompt)}: {len(ids) - len(gp.prompt)} new tokens, accepts {accepts}")
    np.savez_compressed(os.path.join(GOLD, "reference_prompt_runs.npz"), **out)
    print("reference long-prompt vectors written")


def main():
    os.makedirs(GOLD, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "tree":
        return main_tree()
    if len(sys.argv) > 1 and sys.argv[1] == "prompt":
        return main_prompt()
    mu = load_ref_medusa_utils()
    np.savez_compressed(os.path.join(GOLD, "medusa_utils_kat.npz"), **golden_medusa_utils(mu))
    print("medusa_utils known-answer vectors written")
    out = {}
    out.update(golden_model("micro", MedusaConfig.micro(K=4), 11, mu, max_new=40))
    out.update(golden_model("micro10", MedusaConfig.micro(K=10, d_model=128, layers=2), 12, mu, max_new=40))
    out.update(golden_model("tiny", MedusaConfig.tiny_en(K=4), 0, mu, max_new=24))
    np.savez_compressed(os.path.join(GOLD, "reference_linear_runs.npz"), **out)
    print("reference forward()/loop vectors written (Medusa-Linear)")
    out = {}
    out.update(golden_model("microblock", MedusaConfig.micro(K=4, heads_type="medusa_block"), 13, mu, max_new=40))
    out.update(golden_model("micro10block", MedusaConfig.micro(K=10, heads_type="medusa_block", d_model=128, layers=2), 14, mu, max_new=40))
    out.update(golden_model("tinyblock", MedusaConfig.tiny_en("medusa_block", K=4), 1, mu, max_new=24))
    np.savez_compressed(os.path.join(GOLD, "reference_block_runs.npz"), **out)
    print("reference forward()/loop vectors written (Medusa-Block)")


if __name__ == "__main__":
    main()
