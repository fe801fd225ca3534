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


def _byte_level_whisper_tokenizer():
    """A real `transformers.WhisperTokenizer` (byte-level BPE with Whisper's special tokens) built from a vocabulary made here — no
    checkpoint files exist offline.  It encodes / decodes any text; what the evaluation loop needs from the checkpoint's tokenizer."""
    from transformers import WhisperTokenizer
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    cs, n = bs[:], 0
    for b in range(256):
        if b not in bs:
            bs.append(b); cs.append(256 + n); n += 1
    vocab = {chr(c): i for i, c in enumerate(cs)}
    merges = [("t", "h"), ("th", "e"), ("Ġ", "the"), ("i", "n"), ("Ġ", "a"), ("o", "r"), ("Ġ", "w")]
    for a, b in merges:
        vocab[a + b] = len(vocab)
    specials = ["<|endoftext|>", "<|startoftranscript|>", "<|en|>", "<|de|>", "<|transcribe|>", "<|notimestamps|>"]
    for t in specials:
        vocab[t] = len(vocab)
    tok = WhisperTokenizer(vocab=vocab, merges=merges)
    tok.add_special_tokens({"additional_special_tokens": specials[1:]})
    return tok, vocab


def test_evaluate_loop_with_a_real_hf_whisper_tokenizer(tmp_path):
    """eval_whisper_medusa.py:21-97 end to end on the host: token ids as generate() returns them (prompt, text, EOS, padding) ->
    `WhisperTokenizer.decode(skip_special_tokens=True)` -> the jiwer-style normalisation -> WER / CER, with an actual transformers
    tokenizer object instead of a stub (the engine side is a stand-in: there is no GPU in this test)."""
    import pandas as pd
    import torch
    from whisper_medusa.evaluate import evaluate_model
    tok, vocab = _byte_level_whisper_tokenizer()
    prompt = [vocab[t] for t in ("<|startoftranscript|>", "<|en|>", "<|transcribe|>", "<|notimestamps|>")]
    eos = vocab["<|endoftext|>"]
    said = {"a.wav": " The quick brown fox, in a hat.", "b.wav": " the world is wide", "c.wav": " Grüße aus Köln!"}

    class Model:
        def features_from_file(self, path):
            return path

        def generate(self, feats, language=None, exponential_decay_length_penalty=None):
            ids = prompt + tok.encode(said[feats], add_special_tokens=False) + [eos]
            return torch.tensor([ids + [eos] * 3])                      # right-padded with the pad (= EOS) id, as generate() pads

    data = pd.DataFrame({"audio": list(said), "sentence": ["the quick brown fox in a hat", "The world is white.", "Grüße aus Köln"]})
    res = evaluate_model(Model(), tok, data, language="en", out_file_path=str(tmp_path / "r.csv"))
    assert res.prediction.tolist() == [" The quick brown fox, in a hat.", " the world is wide", " Grüße aus Köln!"]
    assert res.wer.tolist() == [0.0, 0.25, 0.0]                          # one substitution in four reference words
    assert abs(res.attrs["wer"] - 1 / 14) < 1e-12 and res.cer.tolist()[0] == 0.0 and 0 < res.cer.tolist()[1] < 0.3
    back = pd.read_csv(tmp_path / "r.csv")
    assert back.prediction.tolist()[2].strip() == "Grüße aus Köln!"      # byte-level round trip of non-ASCII text through the CSV
