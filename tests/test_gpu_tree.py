"""Candidate trees on the GPU (medusa_choices with top-k > 1): token parity of the engine's tree iteration — top-k candidates,
ONE ancestor-masked verify pass over the tree's nodes, multi-path accept, K/V rows of the chosen path moved into place —
against oracle.decode_tree on the engine's own encoder output.  The oracle's tree decode is pinned to the reference's
functions by tests/test_tree_golden.py.

Token ids must be identical; where they are not, the oracle's first differing iteration must contain a decision whose margin
is below the numerical difference of two correct implementations (Oracle._tree_margins, printed) — the ids up to that
iteration are then what was verified."""
import numpy as np
import pytest
import torch

from helpers import MedusaConfig, synth, golden_gen_params, clip_for, ACCEPT_TYPICAL, ACCEPT_GREEDY
from oracle.whisper_medusa_oracle import Oracle, log_mel
from whisper_medusa import WhisperMedusaModel

pytestmark = pytest.mark.gpu

TREES = {
    "micro1221": (lambda c: MedusaConfig.micro(K=3, medusa_choices=c), 21, [1, 2, 2, 1]),
    "micro132": (lambda c: MedusaConfig.micro(K=2, medusa_choices=c), 22, [1, 3, 2]),
    "micro1222block": (lambda c: MedusaConfig.micro(K=3, heads_type="medusa_block", medusa_choices=c), 23, [1, 2, 2, 2]),
    "micro142": (lambda c: MedusaConfig.micro(K=2, d_model=256, layers=3, medusa_choices=c), 24, [1, 4, 2]),
    "tiny12211": (lambda c: MedusaConfig.from_dict({**MedusaConfig.tiny_en(K=4).to_dict(), "medusa_choices": c}), 2, [1, 2, 2, 1, 1]),
    # more than 16 nodes: several 16-row query tiles per stream (31 nodes / 16 paths; the K = 10 shape with top-2 on two heads: 39 nodes)
    "micro12222": (lambda c: MedusaConfig.micro(K=4, medusa_choices=c), 25, [1, 2, 2, 2, 2]),
    "micro10_122": (lambda c: MedusaConfig.micro(K=10, d_model=128, layers=2, medusa_choices=c), 26, [1, 2, 2] + [1] * 8),
    "tiny1311": (lambda c: MedusaConfig.from_dict({**MedusaConfig.tiny_en(K=4).to_dict(), "medusa_choices": c}), 3, [1, 3, 2, 1, 1]),
    # the limits of wm_config.medusa_choices: 53 nodes (four 16-row query tiles per stream, the last one partly filled), 32 paths, top-4
    "micro1442": (lambda c: MedusaConfig.micro(K=3, medusa_choices=c), 27, [1, 4, 4, 2]),
}


class TreeRig:
    def __init__(self, tag, dev, B=3):
        mk, seed, ch = TREES[tag]
        self.cfg = mk(ch)
        assert self.cfg.is_tree
        self.sd = synth.synth_state_dict(self.cfg, seed=seed)
        self.orc = Oracle(self.cfg, self.sd, sim="bf16")
        self.model = WhisperMedusaModel(self.cfg, self.sd, device=dev, max_batch=B)
        self.eng = self.model.engine
        self.dev, self.B = dev, B
        self.wavs = [clip_for(self.cfg, i) for i in range(B)]
        n = self.cfg.n_mel_frames * 160
        self.feats = np.stack([log_mel(w, self.cfg.num_mel_bins, n) for w in self.wavs])
        self.eng.encode(torch.from_numpy(self.feats).to(dev))
        self.enc = self.eng.encoder_output(B)

    def encode(self, B=None):
        B = B or self.B
        self.eng.encode(torch.from_numpy(self.feats[:B]).to(self.dev))


@pytest.fixture(scope="module", params=list(TREES))
def trig(request, gpu):
    r = TreeRig(request.param, gpu)
    yield r
    r.eng.close()


def check_tree_tokens(orc, enc_b, gp, got, label, groups=None):
    """Strict comparison first.  If it differs and the engine's per-iteration token groups are known (streaming callback), EVERY
    iteration is compared: the oracle runs each iteration from the engine's prefix (teacher forcing, Oracle.decode_tree
    engine_groups); iterations whose outcome differs must hold a decision within the numerical tolerance (margins printed, at most
    two per run) and all others must match exactly — a bug that appears after (or inside) a near-tie iteration no longer passes
    (ADVICE r02).  Without groups: the round-2 check (ids up to the first tie)."""
    r = orc.decode_tree(enc_b, gp, engine_ids=got)
    if r.tie is None:
        assert r.ids == got, (label, r.ids, got)
        return r.accept_lengths, True
    print(f"tree parity[{label}]: first difference in the iteration at L={r.tie['L']}; oracle margins (units of tolerance): {r.tie['margin']}")
    assert r.tie["margin"]["min"] < 1.0, (label, r.tie, got)
    if groups is not None:
        groups = groups() if callable(groups) else groups
        rf = orc.decode_tree(enc_b, gp, engine_groups=groups)
        assert 1 <= len(rf.ties) <= 2 and all(t["margin"]["min"] < 1.0 for t in rf.ties), (label, rf.ties)
        assert rf.n_iters == len(groups), (label, rf.n_iters, len(groups))
        print(f"tree parity[{label}]: {rf.n_iters - len(rf.ties)} of {rf.n_iters} iterations identical from the engine's prefix, {len(rf.ties)} numerical tie(s)")
        return rf.accept_lengths, False
    assert r.verified >= len(gp.prompt) + 4, (label, r.tie)
    return r.accept_lengths, False


def decode_with_groups(eng, gp, B):
    """engine.decode in streaming mode: (final sequences, per-stream list of the tokens every iteration emitted)."""
    groups = [[] for _ in range(B)]

    def on_it(new):
        for b in range(B):
            if new[b]:
                groups[b].append(list(new[b]))
    seqs = eng.decode(gp, B, on_iteration=on_it)
    return seqs, groups


@pytest.mark.parametrize("mode", [ACCEPT_TYPICAL, ACCEPT_GREEDY])
def test_tree_decode_tokens(trig, mode):
    gp = golden_gen_params(trig.cfg, mode, 40)
    trig.encode()
    seqs = trig.eng.decode(gp, trig.B)
    st = trig.eng.stats()
    hist = np.zeros(trig.cfg.medusa_num_heads + 1, dtype=np.int64)
    full = True
    cache = {}

    def groups_of(b):
        # a strict difference somewhere: the batch runs once more in streaming mode (one iteration per call) to learn which tokens every
        # iteration emitted, so that EVERY iteration can be checked from the engine's prefix
        if "g" not in cache:
            trig.encode()
            streamed, cache["g"] = decode_with_groups(trig.eng, gp, trig.B)
            assert streamed == seqs
        return cache["g"][b]
    for b in range(trig.B):
        accepts, complete = check_tree_tokens(trig.orc, trig.enc[b], gp, seqs[b], (b, mode), groups=lambda b=b: groups_of(b))
        full = full and complete
        for a in accepts:
            hist[a] += 1
    if full:
        assert st["accept_hist"][: len(hist)] == hist.tolist()


def test_tree_single_stream_and_eos(trig):
    """B = 1 takes the host-driven hidden-state carry schedule; EOS allowed: the stop rules and the post-EOS overwrite."""
    for eos_free in (True, False):
        gp = golden_gen_params(trig.cfg, ACCEPT_TYPICAL, 40, suppress_eos=eos_free)
        trig.encode(1)
        seqs = trig.eng.decode(gp, 1)
        check_tree_tokens(trig.orc, trig.enc[0], gp, seqs[0], ("B1", eos_free))
    trig.encode()


def test_tree_carry_is_bit_identical_to_two_passes(trig, monkeypatch):
    gp = golden_gen_params(trig.cfg, ACCEPT_TYPICAL, 32)
    runs = {}
    for carry in (True, False):
        if carry:
            monkeypatch.delenv("WM_NO_CARRY", raising=False)
        else:
            monkeypatch.setenv("WM_NO_CARRY", "1")
        trig.encode()
        both = trig.eng.decode(gp, trig.B)
        trig.encode(1)
        one = trig.eng.decode(gp, 1)
        runs[carry] = (both, one, trig.eng.stats()["accept_hist"])
    monkeypatch.delenv("WM_NO_CARRY", raising=False)
    trig.encode()
    assert runs[True][0] == runs[False][0] and runs[True][1] == runs[False][1]
    assert runs[True][1][0] == runs[True][0][0]                       # stream 0 does not depend on its batch


def test_tree_greedy_equals_vanilla(trig):
    """size-independent property: with exact-match verification ANY candidate tree emits the plain greedy tokens."""
    gp = golden_gen_params(trig.cfg, ACCEPT_GREEDY, 36)
    trig.encode()
    med = trig.eng.decode(gp, trig.B)
    gp.vanilla = True
    van = trig.eng.decode(gp, trig.B)
    for b in range(trig.B):
        n = min(len(med[b]), len(van[b]))
        assert n >= len(gp.prompt) + 30 and med[b][:n] == van[b][:n]


def test_chain_choices_unchanged(gpu):
    """medusa_choices = [1]*(K+1) through the same ABI field is the chain engine: identical to the oracle's chain loop."""
    cfg = MedusaConfig.micro(K=3)
    sd = synth.synth_state_dict(cfg, seed=21)
    model = WhisperMedusaModel(cfg, sd, device=gpu, max_batch=1)
    orc = Oracle(cfg, sd, sim="bf16")
    feats = log_mel(clip_for(cfg, 0), cfg.num_mel_bins, cfg.n_mel_frames * 160)[None]
    model.engine.encode(torch.from_numpy(feats).to(gpu))
    enc = model.engine.encoder_output(1)
    gp = golden_gen_params(cfg, ACCEPT_TYPICAL, 36)
    got = model.engine.decode(gp, 1)[0]
    assert got == orc.decode(enc[0], gp).ids == orc.decode_tree(enc[0], gp, [1, 1, 1, 1]).ids
    model.engine.close()
