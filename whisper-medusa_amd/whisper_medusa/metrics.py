"""WER / CER exactly as the reference's evaluation computes them.

Reference: whisper_medusa/utils/metrics.py:5-71 — jiwer transform pipelines + ``jiwer.compute_measures``.  jiwer==3.0.3
(requirements.txt:1) is not a dependency here: the transforms it applies are restated from its published source
(jiwer/transforms.py) and the measures reduce to a Levenshtein distance: for an optimal alignment
substitutions + deletions + insertions = edit distance and hits + substitutions + deletions = len(reference).
Host-side Python, like the reference."""
from __future__ import annotations

import re
import sys
import unicodedata
from typing import List, Sequence, Tuple

_PUNCT = None


def _punctuation_table():
    """jiwer RemovePunctuation: every code point whose Unicode category starts with 'P'."""
    global _PUNCT
    if _PUNCT is None:
        _PUNCT = {i: None for i in range(sys.maxunicode + 1) if unicodedata.category(chr(i)).startswith("P")}
    return _PUNCT


def _expand_contractions(s: str) -> str:            # jiwer ExpandCommonEnglishContractions, same order
    s = re.sub(r"won't", "will not", s)
    s = re.sub(r"can\'t", "can not", s)
    s = re.sub(r"let\'s", "let us", s)
    s = re.sub(r"n\'t", " not", s)
    s = re.sub(r"\'re", " are", s)
    s = re.sub(r"\'s", " is", s)
    s = re.sub(r"\'d", " would", s)
    s = re.sub(r"\'ll", " will", s)
    s = re.sub(r"\'t", " not", s)
    s = re.sub(r"\'ve", " have", s)
    s = re.sub(r"\'m", " am", s)
    return s


def _common_tail(s: str) -> str:
    for c in " \t\n\r\x0b\x0c":                     # RemoveWhiteSpace(replace_by_space=True)
        s = s.replace(c, " ")
    s = re.sub(r"\s\s+", " ", s)                    # RemoveMultipleSpaces
    s = s.translate(_punctuation_table())           # RemovePunctuation
    return s.strip()                                # Strip


def wer_standardize(s: str) -> List[str]:
    """metrics.py:6-17: lower, expand contractions, drop Kaldi non-words, whitespace, punctuation, strip, split."""
    s = s.lower()
    s = _expand_contractions(s)
    s = re.sub(r"[<\[][^>\]]*[>\]]", "", s)         # RemoveKaldiNonWords
    s = _common_tail(s)
    return [w for w in s.split(" ") if len(w) >= 1]  # ReduceToListOfListOfWords


def cer_standardize(s: str) -> List[str]:
    """metrics.py:43-52: lower, whitespace, punctuation, strip, characters."""
    return list(_common_tail(s.lower()))             # ReduceToListOfListOfChars


def edit_distance(ref: Sequence, hyp: Sequence) -> int:
    """Levenshtein distance with unit costs (what rapidfuzz gives jiwer)."""
    if len(ref) < len(hyp):
        ref, hyp = hyp, ref
    prev = list(range(len(hyp) + 1))
    for i, r in enumerate(ref, 1):
        cur = [i] + [0] * len(hyp)
        for j, h in enumerate(hyp, 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (r != h))
        prev = cur
    return prev[-1]


def _corpus_rate(predictions: Sequence[str], references: Sequence[str], std) -> Tuple[float, List[float]]:
    incorrect = total = 0
    rates = []
    for prediction, reference in zip(predictions, references):
        if not std(reference):                      # metrics.py:23-26 / :58-61
            reference = "EMPTY"
        if not std(prediction):
            prediction = "EMPTY"
        r, h = std(reference), std(prediction)
        d = edit_distance(r, h)
        rates.append(d / len(r))
        incorrect += d
        total += len(r)
    return incorrect / total, rates


def compute_wer(predictions: Sequence[str], references: Sequence[str]) -> Tuple[float, List[float]]:
    """(corpus WER, per-utterance WERs), metrics.py:5-40."""
    return _corpus_rate(predictions, references, wer_standardize)


def compute_cer(predictions: Sequence[str], references: Sequence[str]) -> Tuple[float, List[float]]:
    """(corpus CER, per-utterance CERs), metrics.py:43-77."""
    return _corpus_rate(predictions, references, cer_standardize)
