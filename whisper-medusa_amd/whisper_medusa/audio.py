"""Audio front door: what the reference's callers do before the feature extractor —
`torchaudio.load` (PCM WAV decode to float in [-1, 1)), stereo -> mono mean, `torchaudio.transforms.Resample(sr, 16000)`
(reference README.md:120-125, eval_whisper_medusa.py:41-45).  torchaudio is not a dependency here: WAV files (PCM and IEEE float,
plain and WAVE_FORMAT_EXTENSIBLE headers) are decoded here, downmix + resampling run on the GPU (`wm_resample`, csrc/wm_encoder.hip).
Compressed containers (FLAC, MP3, ...) are not: decode them to WAV first."""
from __future__ import annotations

import wave
from typing import Tuple

import numpy as np

SAMPLING_RATE = 16000


def _riff_chunks(b: bytes):
    """(fmt chunk, data chunk) of a RIFF / WAVE file.  The standard library's `wave` module only knows PCM (format tag 1): IEEE-float files
    (tag 3) and WAVE_FORMAT_EXTENSIBLE headers (tag 0xFFFE: every multi-channel or > 16-bit file a modern tool writes) need the chunks."""
    if len(b) < 12 or b[:4] != b"RIFF" or b[8:12] != b"WAVE":
        raise ValueError("not a RIFF / WAVE file")
    fmt = data = None
    pos = 12
    while pos + 8 <= len(b):
        cid, size = b[pos: pos + 4], int.from_bytes(b[pos + 4: pos + 8], "little")
        body = b[pos + 8: pos + 8 + size]
        if cid == b"fmt ":
            fmt = body
        elif cid == b"data":
            data = body                                    # (a streamed file may announce more than it holds: take what is there)
            break
        pos += 8 + size + (size & 1)                       # chunks are word aligned
    if fmt is None or data is None or len(fmt) < 16:
        raise ValueError("corrupt WAV: missing fmt / data chunk")
    return fmt, data


def read_wav(path: str) -> Tuple[np.ndarray, int]:
    """WAV -> (float32 [channels, n] in [-1, 1), sample rate), normalised like ``torchaudio.load(normalize=True)``:
    PCM 8-bit unsigned (x - 128) / 128, 16/24/32-bit signed x / 2**(bits-1); IEEE float 32 / 64 as they are; plain and
    WAVE_FORMAT_EXTENSIBLE headers."""
    with open(str(path), "rb") as f:
        fmt, raw = _riff_chunks(f.read())
    tag, ch, sr = int.from_bytes(fmt[0:2], "little"), int.from_bytes(fmt[2:4], "little"), int.from_bytes(fmt[4:8], "little")
    align, bits = int.from_bytes(fmt[12:14], "little"), int.from_bytes(fmt[14:16], "little")
    if tag == 0xFFFE:                                      # extensible: the real format is the first two bytes of the sub-format GUID
        if len(fmt) < 40:
            raise ValueError("corrupt WAV: short WAVE_FORMAT_EXTENSIBLE header")
        tag = int.from_bytes(fmt[24:26], "little")
    if ch < 1 or align < 1 or align % ch:
        raise ValueError("corrupt WAV: block alignment does not match the channel count")
    width = align // ch                                    # container bytes per sample
    raw = raw[: len(raw) // align * align]
    if tag == 3:
        if width not in (4, 8):
            raise ValueError(f"unsupported IEEE-float sample width {width}")
        x = np.frombuffer(raw, dtype="<f4" if width == 4 else "<f8").astype(np.float32)
    elif tag == 1:
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
    else:
        raise ValueError(f"unsupported WAV format tag {tag} (PCM = 1 and IEEE float = 3 are read)")
    _ = bits
    return np.ascontiguousarray(x.reshape(-1, ch).T), int(sr)


def write_wav(path: str, x: np.ndarray, sr: int) -> None:
    """float [channels, n] or [n] in [-1, 1] -> 16-bit PCM WAV (test fixtures, examples)."""
    x = np.atleast_2d(np.asarray(x, dtype=np.float32))
    q = np.clip(np.round(x.T * 32768.0), -32768, 32767).astype("<i2")
    with wave.open(str(path), "wb") as w:
        w.setnchannels(x.shape[0]); w.setsampwidth(2); w.setframerate(int(sr))
        w.writeframes(q.tobytes())
