"""Candidate trees (medusa_choices with top-k > 1): the oracle's restatements and the product's host-side buffers against the
known-answer vectors minted from the reference's own functions (oracle/make_golden.py tree -> tests/golden/medusa_tree_kat.npz,
reference_tree_runs.npz).  CPU only."""
import numpy as np
import pytest
import torch

from helpers import GOLD, golden_gen_params, clip_for, synth, ACCEPT_TYPICAL, ACCEPT_GREEDY, GenParams, MedusaConfig
from oracle.whisper_medusa_oracle import Oracle, log_mel, medusa_buffers, tree_candidates, evaluate_posterior_multi
from whisper_medusa.config import tree_buffers

KAT = np.load(f"{GOLD}/medusa_tree_kat.npz")
RUNS = np.load(f"{GOLD}/reference_tree_runs.npz")
CHOICES = ([1, 3, 2], [1, 2, 2, 1], [1, 2, 1, 2, 1], [1, 4, 2], [1, 2, 2, 2], [1, 1, 3, 1])


@pytest.mark.parametrize("ch", CHOICES, ids=lambda c: "".join(map(str, c)))
def test_tree_buffers_match_reference(ch):
    """generate_medusa_buffers (medusa_utils.py:305-421): oracle restatement == product derivation == reference output."""
    tag = "buf" + "".join(str(x) for x in ch)
    o, p = medusa_buffers(ch), tree_buffers(ch)
    assert o["tree_indices"] == p["tree_indices"] == KAT[tag + "_tree_indices"].tolist()
    assert o["depth"] == p["depth"] == KAT[tag + "_position_ids"].tolist()
    assert o["retrieve"].tolist() == p["retrieve_indices"] == KAT[tag + "_retrieve_indices"].tolist()
    mask = KAT[tag + "_attn_mask"]
    want = [int(sum(1 << j for j in range(mask.shape[0]) if mask[m, j])) for m in range(mask.shape[0])]
    assert o["anc"] == p["anc_mask"] == want
    # structure: a node's ancestors are its parent's ancestors plus itself; every path walks parent links
    for m, par in enumerate(p["parent"]):
        assert p["anc_mask"][m] == (1 << m) | (p["anc_mask"][par] if par >= 0 else 0)
    for row in p["retrieve_indices"]:
        assert all(p["parent"][row[i]] == row[i - 1] for i in range(1, len(row)))


def test_tree_limits_are_checked():
    assert MedusaConfig.micro(K=3, medusa_choices=[1, 3, 3, 2]).is_tree   # 1 + 3 + 9 + 18 = 31 nodes, 18 paths: fits since round 3
    with pytest.raises(ValueError):
        MedusaConfig.micro(K=3, medusa_choices=[1, 4, 4, 4])            # 1 + 4 + 16 + 64 nodes
    assert MedusaConfig.large_v2(K=10, medusa_choices=[1, 2, 2] + [1] * 8).is_tree      # 39 nodes: top-2 on the first two heads of a K = 10 checkpoint
    with pytest.raises(ValueError):
        MedusaConfig.micro(K=2, medusa_choices=[1, 5, 1])               # top-5
    with pytest.raises(ValueError):
        MedusaConfig.micro(K=2, medusa_choices=[2, 1, 1])
    assert MedusaConfig.micro(K=3, medusa_choices=[1, 2, 2, 2]).is_tree and not MedusaConfig.micro(K=3).is_tree


@pytest.mark.parametrize("ci", [0, 1, 2])
def test_tree_candidates_match_reference(ci):
    tag = f"tree{ci}"
    ch = KAT[tag + "_choices"].tolist()
    base, med = torch.from_numpy(KAT[tag + "_cand_base"]), torch.from_numpy(KAT[tag + "_cand_med"])
    z = torch.cat([base[0], med[:, 0, 0]], dim=0)
    cands, flat = tree_candidates(z, ch)
    assert cands.tolist() == KAT[tag + "_cand_out"].tolist()
    assert flat[torch.tensor(tree_buffers(ch)["tree_indices"])].tolist() == KAT[tag + "_cand_tree_out"][0].tolist()


@pytest.mark.parametrize("ci", [0, 1, 2])
@pytest.mark.parametrize("mode,mname", [(ACCEPT_TYPICAL, "typical"), (ACCEPT_GREEDY, "greedy")])
def test_multi_path_posterior_matches_reference(ci, mode, mname):
    """evaluate_posterior (medusa_utils.py:526-588) over several paths: accept length AND the chosen path."""
    tag = f"tree{ci}"
    ch = KAT[tag + "_choices"].tolist()
    retrieve = torch.tensor(tree_buffers(ch)["retrieve_indices"])
    gp = GenParams(prompt=[1, 2], eos_token_id=0, pad_token_id=0, accept_mode=mode, temperature=1.0 if mode == ACCEPT_TYPICAL else 0.0)
    want_a, want_b = KAT[f"{tag}_post_accept_{mname}"], KAT[f"{tag}_post_best_{mname}"]
    assert len(set(want_a.tolist())) >= 2 and len(set(want_b.tolist())) >= 3
    for i in range(len(want_a)):
        node_logits, cand = torch.from_numpy(KAT[tag + "_post_node_logits"][i]), torch.from_numpy(KAT[tag + "_post_cand"][i])
        best, a, _ = evaluate_posterior_multi(node_logits[retrieve], cand, gp)
        assert (best, a) == (int(want_b[i]), int(want_a[i])), (i, best, a, want_b[i], want_a[i])


TREE_MODELS = {
    "micro1221": (lambda: MedusaConfig.micro(K=3), 21, [1, 2, 2, 1], 36),
    "micro132": (lambda: MedusaConfig.micro(K=2), 22, [1, 3, 2], 36),
    "micro1222block": (lambda: MedusaConfig.micro(K=3, heads_type="medusa_block"), 23, [1, 2, 2, 2], 30),
    "micro12222": (lambda: MedusaConfig.micro(K=4), 25, [1, 2, 2, 2, 2], 30),                                      # 31 nodes, 16 paths
    "micro10_122": (lambda: MedusaConfig.micro(K=10, d_model=128, layers=2), 26, [1, 2, 2] + [1] * 8, 36),        # 39 nodes, 4 paths
}


@pytest.mark.parametrize("tag", list(TREE_MODELS))
def test_tree_decode_matches_reference_runs(tag):
    """oracle.decode_tree (ONE masked pass over the tree's nodes) == the reference's forward() along every path + the
    reference's candidates / posterior, token for token, accept lengths included."""
    mk, seed, ch, max_new = TREE_MODELS[tag]
    cfg = mk()
    assert RUNS[f"{tag}_choices"].tolist() == ch
    sd = synth.synth_state_dict(cfg, seed=seed)
    orc = Oracle(cfg, sd, sim="fp32")
    enc = orc.encode(torch.from_numpy(log_mel(clip_for(cfg), cfg.num_mel_bins, cfg.n_mel_frames * 160)))
    for mode, mname in ((ACCEPT_TYPICAL, "typical"), (ACCEPT_GREEDY, "greedy")):
        gp = golden_gen_params(cfg, mode, max_new)
        r = orc.decode_tree(enc, gp, ch)
        assert r.ids == RUNS[f"{tag}_{mname}_ids"].tolist(), (tag, mname)
        assert r.accept_lengths == RUNS[f"{tag}_{mname}_accepts"].tolist()
    assert max(RUNS[f"{tag}_typical_best"].tolist()) > 0            # a path other than the top-1 chain won at least once


def test_chain_tree_equals_chain_loop():
    """decode_tree with medusa_choices = [1]*(K+1) is the chain loop; in greedy mode any tree emits the vanilla greedy tokens."""
    cfg = MedusaConfig.micro(K=3)
    sd = synth.synth_state_dict(cfg, seed=5)
    orc = Oracle(cfg, sd)
    enc = orc.encode(torch.from_numpy(log_mel(clip_for(cfg, 1), 80, cfg.n_mel_frames * 160)))
    gp = golden_gen_params(cfg, ACCEPT_TYPICAL, 30)
    a, b = orc.decode(enc, gp), orc.decode_tree(enc, gp, [1, 1, 1, 1])
    assert a.ids == b.ids and a.accept_lengths == b.accept_lengths
    gpg = golden_gen_params(cfg, ACCEPT_GREEDY, 30)
    t = orc.decode_tree(enc, gpg, [1, 2, 2, 2]).new_tokens
    gpg.vanilla = True
    v = orc.decode(enc, gpg).new_tokens
    n = min(len(t), len(v))
    assert t[:n] == v[:n] and n >= 27
