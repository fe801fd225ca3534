"""Golden vectors for the audio front door, minted WITHOUT torch and without the oracle's restatement.

torchaudio 2.2.2 (requirements.txt:6 of the reference, not installed here) documents its default resampler
(`torchaudio.functional.resample`, method "sinc_interp_hann", lowpass_filter_width 6, rolloff 0.99) as a windowed-sinc
interpolation.  Written out in continuous time, with orig/new reduced by their gcd:

    base = min(orig, new) * rolloff
    g(tau) = (base / orig) * sinc(base * tau) * cos^2( clamp(base * tau, -lpw, +lpw) * pi / (2 lpw) )      sinc(x) = sin(pi x) / (pi x)
    y[n]   = sum_m x[m] * g(m / orig - n / new)                    n = 0 .. ceil(new * len(x) / orig) - 1

(the polyphase kernel table of `_get_sinc_resample_kernel` is g sampled at (k - width) / orig - i / new, and the strided
conv1d of `_apply_sinc_resample_kernel` is this sum restricted to the taps where the window is non-zero).  This script
evaluates that formula DIRECTLY in float64 with exact rational sample times — an implementation independent of the oracle's
float32 kernel table / conv1d path — and stores inputs and outputs in tests/golden/resample_f64.npz.
The oracle (float32 taps, float32 accumulation) must agree to float32 round-off (tests/test_audio_cpu.py).
"""
import sys as _sys
_sys.dont_write_bytecode = True          # never write .pyc files into /root/reference
import math
import os
from fractions import Fraction

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden", "resample_f64.npz")


def resample_direct_f64(x, orig_freq, new_freq, lpw=6, rolloff=0.99):
    g = math.gcd(orig_freq, new_freq)
    orig, new = orig_freq // g, new_freq // g
    base = min(orig, new) * rolloff
    n_out = -(-new * len(x) // orig)
    x = np.asarray(x, dtype=np.float64)
    m = np.arange(len(x), dtype=np.float64)
    y = np.zeros(n_out, dtype=np.float64)
    span = lpw * orig / base                                  # |m - n*orig/new| beyond this: window is exactly zero
    for n in range(n_out):
        c = Fraction(n * orig, new)                           # centre, in input samples
        lo = max(0, math.floor(float(c) - span) - 1); hi = min(len(x), math.ceil(float(c) + span) + 2)
        tau = (m[lo:hi] - float(c)) / orig                    # m/orig - n/new
        t = np.clip(base * tau, -lpw, lpw)
        w = np.cos(t * math.pi / lpw / 2.0) ** 2
        s = np.sinc(t)                                        # numpy sinc is sin(pi t)/(pi t); the clamp only acts where w == 0
        y[n] = np.sum(x[lo:hi] * s * w) * (base / orig)
    return y


def main():
    rng = np.random.default_rng(20250924)
    out = {}
    for sr in (44100, 48000, 22050, 8000, 11025, 32000):
        n = 1500 + sr % 97
        x = np.clip(0.4 * rng.standard_normal(n), -1, 1).astype(np.float32)
        out[f"x_{sr}"] = x
        out[f"y_{sr}"] = resample_direct_f64(x, sr, 16000)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
