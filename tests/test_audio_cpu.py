"""Audio front door, CPU side: the oracle's restatement of torchaudio's default resampler against independent
references (analytic band-limited signals, scipy's polyphase resampler), WAV decode, and the C-ABI length helper."""
import ctypes
import math
import struct

import numpy as np
import pytest
from scipy.signal import resample_poly

from helpers import ROOT  # noqa: F401  (sys.path)
from oracle.whisper_medusa_oracle import downmix_mono, resample_sinc_hann
from whisper_medusa.audio import read_wav, write_wav

RATES = [44100, 48000, 22050, 8000, 11025, 32000]


def tones(sr, n, top=3000.0):
    """three tones, the highest at ``top`` Hz (kept <= half the narrower Nyquist band: the 6-zero-crossing filter's
    transition band is wide)"""
    freqs = ((440.0, 0.5), (1234.5, 0.25), (top, 0.2))
    t = np.arange(n, dtype=np.float64) / sr
    return sum(a * np.sin(2 * np.pi * f * t + 0.3 * i) for i, (f, a) in enumerate(freqs))


@pytest.mark.parametrize("sr", RATES)
def test_oracle_resampler_reproduces_band_limited_signals(sr):
    n = sr + 137                                           # a bit more than 1 s, not a multiple of anything
    top = min(3000.0, 0.25 * min(sr, 16000))
    x = tones(sr, n, top).astype(np.float32)
    y = resample_sinc_hann(x, sr, 16000)
    assert y.dtype == np.float32 and y.shape == (math.ceil(16000 * n / sr),)
    ref = tones(16000, len(y), top)
    assert np.abs(y[300:-300] - ref[300:-300]).max() <= 3e-3            # interior (edges see the zero padding); passband ripple of a 6-zero-crossing filter
    g = math.gcd(sr, 16000)
    z = resample_poly(x.astype(np.float64), 16000 // g, sr // g)
    assert np.abs(y[300:-300] - z[300:len(y) - 300]).max() <= 5e-3      # different low-pass design, same signal


def test_oracle_resampler_shapes_identity_and_batches():
    x = np.random.default_rng(0).standard_normal((2, 3, 1000)).astype(np.float32)
    assert resample_sinc_hann(x, 16000, 16000).shape == x.shape and np.array_equal(resample_sinc_hann(x, 16000, 16000), x)
    y = resample_sinc_hann(x, 48000, 16000)
    assert y.shape == (2, 3, math.ceil(1000 / 3))
    assert np.allclose(y[1, 2], resample_sinc_hann(x[1, 2], 48000, 16000), atol=1e-7)
    up = resample_sinc_hann(x[0, 0], 8000, 16000)
    assert up.shape == (2000,)
    # a constant keeps its level (DC gain of the interpolation filter is 1 up to the Hann-windowed truncation)
    c = resample_sinc_hann(np.full(4000, 0.5, np.float32), 44100, 16000)
    assert np.abs(c[100:-100] - 0.5).max() <= 2e-3
    st = np.stack([np.ones(8, np.float32), np.zeros(8, np.float32)])
    assert np.array_equal(downmix_mono(st), np.full(8, 0.5, np.float32))


def test_wav_decode_matches_torchaudio_normalisation(tmp_path):
    x = np.stack([np.linspace(-1, 1, 1000, dtype=np.float32), np.zeros(1000, np.float32)])
    p = tmp_path / "a.wav"
    write_wav(p, x, 44100)
    y, sr = read_wav(p)
    assert sr == 44100 and y.shape == (2, 1000) and y.dtype == np.float32
    assert np.abs(y - np.clip(np.round(x * 32768) / 32768, -1, 32767 / 32768)).max() == 0.0
    # hand-built 8-bit and 24-bit files
    import wave
    for width, frames, want in ((1, bytes([0, 128, 255]), [-1.0, 0.0, 127 / 128]),
                                (3, struct.pack("<i", -(1 << 23))[:3] + struct.pack("<i", 1 << 22)[:3], [-1.0, 0.5])):
        q = tmp_path / f"w{width}.wav"
        with wave.open(str(q), "wb") as w:
            w.setnchannels(1); w.setsampwidth(width); w.setframerate(8000); w.writeframes(frames)
        v, sr = read_wav(q)
        assert sr == 8000 and np.allclose(v[0], want, atol=0)


def test_resample_len_helper_of_the_c_abi(built_lib):
    lib = ctypes.CDLL(built_lib)
    lib.wm_resample_len.argtypes = [ctypes.c_int64, ctypes.c_int32, ctypes.c_int32]
    lib.wm_resample_len.restype = ctypes.c_int64
    for n in (1, 440, 441, 442, 44100 * 30, 1323001):
        for sr in RATES + [16000]:
            assert lib.wm_resample_len(n, sr, 16000) == math.ceil(16000 * n / sr)
    assert lib.wm_resample_len(10, 0, 16000) == -1


def test_oracle_resampler_is_pinned_by_an_independent_float64_evaluation_of_the_published_kernel():
    """tests/golden/resample_f64.npz: the torchaudio default-resampler formula y[n] = sum_m x[m] g(m/orig - n/new) evaluated
    directly in float64 with exact rational sample times (oracle/make_resample_golden.py — no torch, no kernel table, no
    conv1d).  The oracle restates torchaudio's implementation (float32 taps, float32 conv1d, float32 phase quotient i/new):
    it must agree to float32 round-off.  Measured: <= 2.2e-7 where new | 2^k·5^j divides evenly (48 k, 32 k, 8 k), <= 2.1e-5
    for the 441:x ratios, where torchaudio's float32 `arange(0, -new, -1) / new` phase offsets are inexact (6e-8 relative,
    amplified by base_freq ~ 158 into the sinc argument) — a property of torchaudio the restatement keeps on purpose."""
    import os
    g = np.load(os.path.join(ROOT, "tests", "golden", "resample_f64.npz"))
    for sr, bound in ((48000, 5e-7), (32000, 5e-7), (8000, 5e-7), (44100, 3e-5), (22050, 3e-5), (11025, 3e-5)):
        x, want = g[f"x_{sr}"], g[f"y_{sr}"]
        got = resample_sinc_hann(x, sr, 16000)
        assert got.shape == want.shape and got.dtype == np.float32
        assert np.abs(got - want).max() <= bound, (sr, float(np.abs(got - want).max()))
    # and the script reproduces its own fixture (the golden file is not hand-edited)
    import importlib.util
    spec = importlib.util.spec_from_file_location("mk", os.path.join(ROOT, "oracle", "make_resample_golden.py"))
    mk = importlib.util.module_from_spec(spec); spec.loader.exec_module(mk)
    assert np.array_equal(mk.resample_direct_f64(g["x_32000"], 32000, 16000), g["y_32000"])
