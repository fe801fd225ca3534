"""Evaluation loop of the reference (whisper_medusa/eval_whisper_medusa.py:21-97) on the MI355X engine (SURVEY.md §8f
row 2): CSV with `audio`, `sentence` (optional `language`) -> transcribe every file -> WER / CER -> results CSV with the
reference's columns.  `python -m whisper_medusa.evaluate --model-name DIR --data-path in.csv --out-file-path out.csv`.

The tokenizer is the checkpoint's own (`transformers.WhisperTokenizer` files next to the weights); any object with
``decode(ids, skip_special_tokens=True)`` can be passed instead (tests use a stub)."""
from __future__ import annotations

import argparse
import logging
from pathlib import Path
from typing import Optional

import pandas as pd

from .metrics import compute_cer, compute_wer


def evaluate_model(model, tokenizer, data: pd.DataFrame, language: str = "en", regulation_start: float = 140,
                   regulation_factor: float = 1.0, out_file_path: Optional[str] = None) -> pd.DataFrame:
    data = data.fillna("")
    preds, gts, langs, audios = [], [], [], []
    for _, row in data.iterrows():
        lang = row.get("language", language) or language
        feats = model.features_from_file(row.audio)                       # decode + downmix + resample + log-mel on the GPU
        decay = (regulation_start, regulation_factor) if regulation_factor != 1 else None      # eval_whisper_medusa.py:52-58
        out = model.generate(feats, language=lang, exponential_decay_length_penalty=decay)
        preds.append(tokenizer.decode(out[0].tolist(), skip_special_tokens=True))
        gts.append(row.sentence)
        langs.append(language)                                            # the reference logs args.language here (:72)
        audios.append(row.audio)
    wer, wers = compute_wer(preds, gts)
    cer, cers = compute_cer(preds, gts)
    logging.info("WER: %s", wer)
    logging.info("CER: %s", cer)
    results = pd.DataFrame({"audio": audios, "label": gts, "prediction": preds, "wer": wers, "cer": cers, "language": langs})
    results.attrs["wer"], results.attrs["cer"] = wer, cer
    if out_file_path:
        p = Path(out_file_path)
        p.parent.mkdir(parents=True, exist_ok=True)
        results.to_csv(p, index=False)
    return results


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--model-name", required=True, help="checkpoint directory (config.json, weights, tokenizer files)")
    ap.add_argument("--data-path", required=True, help="test data csv (audio, sentence[, language])")
    ap.add_argument("--out-file-path", required=True)
    ap.add_argument("--language", default="en")
    ap.add_argument("--regulation-start", type=float, default=140)
    ap.add_argument("--regulation-factor", type=float, default=1)
    args = ap.parse_args(argv)
    logging.basicConfig(level=logging.INFO)
    from transformers import WhisperTokenizer
    from .api import WhisperMedusaModel
    model = WhisperMedusaModel.from_pretrained(args.model_name).to("cuda")
    tok = WhisperTokenizer.from_pretrained(args.model_name)
    res = evaluate_model(model, tok, pd.read_csv(args.data_path), args.language, args.regulation_start, args.regulation_factor,
                         args.out_file_path)
    logging.info("Results saved to %s (WER %.4f, CER %.4f)", args.out_file_path, res.attrs["wer"], res.attrs["cer"])


if __name__ == "__main__":
    main()
