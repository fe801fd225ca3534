"""Audio front door: what the reference's callers do before the feature extractor —
`torchaudio.load` (PCM WAV decode to float in [-1, 1)), stereo -> mono mean, `torchaudio.transforms.Resample(sr, 16000)`
(reference README.md:120-125, eval_whisper_medusa.py:41-45).  torchaudio is not a dependency here: PCM WAV files are
decoded with the standard library, downmix + resampling run on the GPU (`wm_resample`, csrc/wm_encoder.hip)."""
from __future__ import annotations

import wave
from typing import Tuple

import numpy as np

SAMPLING_RATE = 16000


def read_wav(path: str) -> Tuple[np.ndarray, int]:
    """PCM WAV -> (float32 [channels, n] in [-1, 1), sample rate), normalised like ``torchaudio.load(normalize=True)``:
    8-bit unsigned (x - 128) / 128, 16/24/32-bit signed x / 2**(bits-1)."""
    with wave.open(str(path), "rb") as w:
        ch, width, sr, n = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
        raw = w.readframes(n)
    if width == 1:
        x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    elif width == 2:
        x = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif width == 3:
        b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        v = np.where(v >= 1 << 23, v - (1 << 24), v)
        x = v.astype(np.float32) / float(1 << 23)
    elif width == 4:
        x = (np.frombuffer(raw, dtype="<i4").astype(np.float64) / float(1 << 31)).astype(np.float32)
    else:
        raise ValueError(f"unsupported PCM sample width {width}")
    if ch < 1 or len(x) % ch:
        raise ValueError("corrupt WAV: frame count does not match the channel count")
    return np.ascontiguousarray(x.reshape(-1, ch).T), int(sr)


def write_wav(path: str, x: np.ndarray, sr: int) -> None:
    """float [channels, n] or [n] in [-1, 1] -> 16-bit PCM WAV (test fixtures, examples)."""
    x = np.atleast_2d(np.asarray(x, dtype=np.float32))
    q = np.clip(np.round(x.T * 32768.0), -32768, 32767).astype("<i2")
    with wave.open(str(path), "wb") as w:
        w.setnchannels(x.shape[0]); w.setsampwidth(2); w.setframerate(int(sr))
        w.writeframes(q.tobytes())
