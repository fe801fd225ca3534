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


def test_wav_decode_agrees_with_scipys_reader_on_every_sample_format(tmp_path):
    """read_wav against an independent decoder (scipy.io.wavfile) on the formats torchaudio.load reads: PCM 8 / 16 / 32 bit, IEEE float 32 /
    64 (format tag 3, which the standard library's `wave` refuses), mono and stereo; plus a hand-built WAVE_FORMAT_EXTENSIBLE header
    (tag 0xFFFE with the PCM sub-format GUID: what multi-channel / high-resolution writers emit) and a file with an odd-sized extra chunk
    in front of the data."""
    from scipy.io import wavfile
    rng = np.random.default_rng(3)
    for dt, scale in ((np.uint8, None), (np.int16, 32768.0), (np.int32, 2147483648.0), (np.float32, 1.0), (np.float64, 1.0)):
        for ch in (1, 2):
            if dt == np.uint8:
                a = rng.integers(0, 256, size=(777, ch), dtype=np.uint8)
                want = (a.astype(np.float32) - 128.0) / 128.0
            elif np.issubdtype(dt, np.integer):
                a = rng.integers(np.iinfo(dt).min, np.iinfo(dt).max, size=(777, ch), dtype=dt)
                want = (a.astype(np.float64) / scale).astype(np.float32)
            else:
                a = rng.uniform(-1, 1, size=(777, ch)).astype(dt)
                want = a.astype(np.float32)
            q = tmp_path / f"s_{np.dtype(dt).name}_{ch}.wav"
            wavfile.write(str(q), 22050, a if ch == 2 else a[:, 0])
            sr_s, back = wavfile.read(str(q))                       # the independent reader sees what we wrote
            assert sr_s == 22050 and np.array_equal(np.atleast_2d(back.T).T.reshape(777, ch), a)
            v, sr = read_wav(q)
            assert sr == 22050 and v.shape == (ch, 777) and v.dtype == np.float32 and np.array_equal(v, want.T), (dt, ch)
    # WAVE_FORMAT_EXTENSIBLE, 24-bit PCM in 3-byte containers, stereo, with a 'LIST' chunk of odd size before 'data'
    samples = [-(1 << 23), 1 << 22, 12345, -54321]                 # L R L R
    data = b"".join(struct.pack("<i", s_)[:3] for s_ in samples)
    guid_pcm = struct.pack("<H", 1) + bytes.fromhex("000000001000800000aa00389b71")
    fmt = struct.pack("<HHIIHH", 0xFFFE, 2, 48000, 48000 * 6, 6, 24) + struct.pack("<HHI", 22, 24, 3) + guid_pcm
    lst = b"LIST" + struct.pack("<I", 5) + b"abcde" + b"\x00"
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + lst + b"data" + struct.pack("<I", len(data)) + data
    q = tmp_path / "ext.wav"
    q.write_bytes(b"RIFF" + struct.pack("<I", len(body)) + body)
    v, sr = read_wav(q)
    assert sr == 48000 and v.shape == (2, 2) and np.allclose(v, [[-1.0, 12345 / (1 << 23)], [0.5, -54321 / (1 << 23)]], atol=0)
    with pytest.raises(ValueError, match="format tag"):
        bad = bytearray(q.read_bytes()); bad[20:22] = struct.pack("<H", 85)      # MPEG layer 3 in a WAV container
        (tmp_path / "mp3.wav").write_bytes(bytes(bad)); read_wav(tmp_path / "mp3.wav")
    with pytest.raises(ValueError, match="RIFF"):
        (tmp_path / "x.wav").write_bytes(b"fLaC" + bytes(40)); read_wav(tmp_path / "x.wav")


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
