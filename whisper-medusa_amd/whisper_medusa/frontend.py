"""Constant tables of the log-mel front end (F0) that the engine keeps in HBM next to the weights:
periodic Hann window, 400-point DFT twiddles and the 80 x 201 Slaney mel filter bank that
``WhisperFeatureExtractor`` builds (HF:models/whisper/feature_extraction_whisper.py:95-103)."""
import numpy as np

N_FFT = 400
HOP = 160
SAMPLE_RATE = 16000


def hann_window() -> np.ndarray:
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(N_FFT) / N_FFT)).astype(np.float32)


def dft_twiddles() -> np.ndarray:
    ang = 2.0 * np.pi * np.arange(N_FFT) / N_FFT
    return np.stack([np.cos(ang), np.sin(ang)], axis=1).astype(np.float32)        # [400][2]


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    lin = f / (200.0 / 3.0)
    log = 15.0 + np.log(np.maximum(f, 1e-10) / 1000.0) * (27.0 / np.log(6.4))
    return np.where(f >= 1000.0, log, lin)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    lin = m * (200.0 / 3.0)
    log = 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0))
    return np.where(m >= 15.0, log, lin)


def slaney_mel_bank(n_mels: int = 80) -> np.ndarray:
    """[201][n_mels] float32, Slaney mel scale + Slaney area normalisation, 0..8 kHz."""
    n_freq = N_FFT // 2 + 1
    freqs = np.linspace(0.0, SAMPLE_RATE / 2, n_freq)
    edges = _mel_to_hz(np.linspace(_hz_to_mel(0.0), _hz_to_mel(8000.0), n_mels + 2))
    width = np.diff(edges)
    rel = edges[None, :] - freqs[:, None]
    tri = np.maximum(0.0, np.minimum(-rel[:, :-2] / width[:-1], rel[:, 2:] / width[1:]))
    tri *= (2.0 / (edges[2:] - edges[:-2]))[None, :]
    return tri.astype(np.float32)
