"""WER / CER of the reference's evaluation (utils/metrics.py) restated without jiwer: transform semantics and measures."""
import numpy as np
import pytest

from helpers import ROOT  # noqa: F401
from whisper_medusa.metrics import cer_standardize, compute_cer, compute_wer, edit_distance, wer_standardize


def test_transform_pipeline_follows_jiwer_order():
    assert wer_standardize("He WON'T go; she can't, let's see.") == "he will not go she can not let us see".split()
    assert wer_standardize("It's Bob's [laugh] <unk> they're   I'd  we'll don't I've I'm") == \
        "it is bob is they are i would we will do not i have i am".split()
    assert wer_standardize("tab\there\nnew\r\nline  —  “quoted” ¿qué?") == ["tab", "here", "new", "line", "quoted", "qué"]
    assert wer_standardize("  ...  ") == [] and cer_standardize(" ?! ") == []
    assert cer_standardize("Ab, c") == list("ab c")


def test_edit_distance_known_answers():
    assert edit_distance("kitten", "sitting") == 3
    assert edit_distance([], list("abc")) == 3 and edit_distance(list("abc"), []) == 3 and edit_distance([], []) == 0
    a, b = "the quick brown fox".split(), "the quack brown box jumps".split()
    assert edit_distance(a, b) == 3 == edit_distance(b, a)
    rng = np.random.default_rng(0)
    for _ in range(50):                                    # metric properties on random strings
        x, y, z = (list(rng.integers(0, 4, rng.integers(0, 9))) for _ in range(3))
        assert edit_distance(x, x) == 0 and edit_distance(x, y) == edit_distance(y, x)
        assert abs(len(x) - len(y)) <= edit_distance(x, y) <= max(len(x), len(y))
        assert edit_distance(x, z) <= edit_distance(x, y) + edit_distance(y, z)


def test_corpus_and_utterance_rates_match_the_reference_accounting():
    preds = ["the cat sat on the mat", "hello word", "", "completely different text here"]
    refs = ["The cat sat on the mat.", "hello world", "", "short"]
    wer, wers = compute_wer(preds, refs)
    # utterances: 0/6, 1/2, EMPTY vs EMPTY 0/1, 4 edits / 1 word
    assert wers == [0.0, 0.5, 0.0, 4.0]
    assert wer == pytest.approx((0 + 1 + 0 + 4) / (6 + 2 + 1 + 1))
    cer, cers = compute_cer(["abc", ""], ["abd", "xy"])
    assert cers == [pytest.approx(1 / 3), pytest.approx(4 / 2)]          # "" -> "EMPTY" -> "empty" vs "xy": 4 edits / 2 chars
    assert cer == pytest.approx(5 / 5)


def test_evaluate_loop_with_stub_model_and_tokenizer(tmp_path):
    import pandas as pd
    import torch
    from whisper_medusa.evaluate import evaluate_model

    class Tok:
        def decode(self, ids, skip_special_tokens=True):
            return " ".join({1: "hello", 2: "world", 3: "again"}[i] for i in ids if i in (1, 2, 3))

    class Model:
        calls = []

        def features_from_file(self, path):
            return path

        def generate(self, feats, language=None, exponential_decay_length_penalty=None):
            self.calls.append((feats, language, exponential_decay_length_penalty))
            return torch.tensor([[50258, 1, 2, 50257]] if feats.startswith("a") else [[50258, 1, 3, 50257]])

    data = pd.DataFrame({"audio": ["a.wav", "b.wav"], "sentence": ["Hello, world!", "hello world"], "language": ["en", None]})
    m = Model()
    res = evaluate_model(m, Tok(), data, language="en", regulation_start=140, regulation_factor=1.01, out_file_path=str(tmp_path / "o" / "r.csv"))
    assert list(res.columns) == ["audio", "label", "prediction", "wer", "cer", "language"]
    assert res.prediction.tolist() == ["hello world", "hello again"] and res.wer.tolist() == [0.0, 0.5]
    assert res.attrs["wer"] == 0.25 and m.calls[0][1:] == ("en", (140, 1.01)) and m.calls[1][1] == "en"
    assert (tmp_path / "o" / "r.csv").exists()
    evaluate_model(m, Tok(), data, regulation_factor=1)
    assert m.calls[-1][2] is None                          # factor 1 -> no length penalty (eval_whisper_medusa.py:52-58)


def test_wer_cer_known_answers_worked_by_hand():
    """Known-answer cases worked by hand from jiwer's documented transform chain (the reference's utils/metrics.py:5-71
    composes: ExpandCommonEnglishContractions -> RemoveKaldiNonWords -> RemoveWhiteSpace(replace_by_space) ->
    RemoveMultipleSpaces -> Strip -> RemovePunctuation -> ToLowerCase -> ReduceToListOfListOfWords / Chars) and jiwer's
    measures (WER = (S + D + I) / N_ref over the whole corpus, Levenshtein alignment)."""
    # 1. "I'm here, aren't I?" vs "i am here are not i": contractions expand on both sides, punctuation drops -> identical: 0 / 6
    w, per = compute_wer(["I'm here, aren't I?"], ["i am here are not i"])
    assert per == [0.0] and w == 0.0
    # 2. ref "the quick brown fox jumps" (5 words), hyp "the quick brown box": 1 substitution (fox->box) + 1 deletion (jumps) = 2 / 5
    w, per = compute_wer(["the quick brown box"], ["The quick brown fox jumps."])
    assert per == [pytest.approx(0.4)] and w == pytest.approx(0.4)
    # 3. insertion-only: ref "a b" hyp "a x y b" -> 2 insertions / 2 reference words = 1.0 (WER can exceed hits)
    w, per = compute_wer(["a x y b"], ["a b"])
    assert per == [pytest.approx(1.0)]
    # 4. corpus WER pools edits and reference lengths: (2 + 2) / (5 + 2), NOT the mean of 0.4 and 1.0
    w, per = compute_wer(["the quick brown box", "a x y b"], ["The quick brown fox jumps.", "a b"])
    assert w == pytest.approx(4 / 7) and per == [pytest.approx(0.4), pytest.approx(1.0)]
    # 5. Kaldi non-words vanish before scoring: "<unk> hello [noise] world" == "hello world"
    w, per = compute_wer(["<unk> hello [noise] world"], ["hello world"])
    assert per == [0.0]
    # 6. CER on characters incl. the single spaces left after whitespace normalisation: ref "ab cd" (5 chars), hyp "ab  d" -> "ab d":
    #    one deletion ('c') / 5
    c, per = compute_cer(["ab  d"], ["ab cd"])
    assert per == [pytest.approx(1 / 5)] and c == pytest.approx(1 / 5)
    # 7. empty hypothesis: the reference maps "" to the token "EMPTY" on both sides before scoring (utils/metrics.py) ->
    #    ref "hello world" (2 words) vs hyp "empty" (1 word): 1 substitution + 1 deletion = 2 / 2
    w, per = compute_wer([""], ["hello world"])
    assert per == [pytest.approx(1.0)]
