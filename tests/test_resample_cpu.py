"""The oracle's restatement of `torchaudio.functional.resample` (oracle/audio.py; core/io.py:258-262) and
the product's filter bank (pyannote_audio_amd/audio.py:sinc_resample_bank).  torchaudio is not installed
(parity unpinned, see the oracle's header); what is checked here is everything that can be without it:
the bank against a float64 closed form of the published formula, the output-length rule, DC gain, and the
equivalence of "strided conv1d with a filter bank" to scipy's polyphase resampler driven by the SAME
prototype filter (an independent implementation of the same signal-processing operation)."""
import math

import numpy as np
import pytest
import torch

from oracle.audio import resample, sinc_resample_kernel
from pyannote_audio_amd.audio import sinc_resample_bank


@pytest.mark.parametrize("orig,new", [(8000, 16000), (44100, 16000), (48000, 16000), (22050, 16000)])
def test_filter_bank_matches_closed_form(orig, new):
    g = math.gcd(orig, new)
    kern, width = sinc_resample_kernel(orig, new, g)
    taps, L, P, w2 = sinc_resample_bank(orig, new)
    assert (L, P, w2) == (orig // g, new // g, width) and taps.shape == (P, 2 * width + L)
    assert torch.equal(kern[:, 0], taps)                      # product bank == oracle bank, bit for bit
    # float64 closed form: h_p[k] = s * sinc(s'(k - width)/L - p/P ...) * hann^2, clamped at +-6 lobes
    cutoff = min(L, P) * 0.99
    k = np.arange(-width, width + L, dtype=np.float64)[None, :] / L
    p = -np.arange(P, dtype=np.float64)[:, None] / P
    t = np.clip((p + k) * cutoff, -6, 6)
    want = np.sinc(t) * np.cos(t * np.pi / 12) ** 2 * (cutoff / L)
    # (the bank is evaluated in float32, like torchaudio does for float32 waveforms: sin(pi t) at |t| <= 6
    #  carries ~1e-5 of argument rounding for the long 44.1 kHz grids)
    assert np.abs(taps.numpy() - want).max() < 5e-5
    # unity DC gain per phase, to the leakage of a 6-lobe window
    assert np.abs(taps.numpy().sum(axis=1) - 1.0).max() < 5e-3


@pytest.mark.parametrize("orig,new,n", [(8000, 16000, 1234), (44100, 16000, 44100), (48000, 16000, 100001),
                                        (16000, 16000, 77)])
def test_output_length_and_identity(orig, new, n):
    x = torch.randn(2, n, generator=torch.Generator().manual_seed(0))
    y = resample(x, orig, new)
    assert y.shape == (2, math.ceil(new * n / orig)) if orig != new else y is x


def test_equals_direct_correlation_with_same_prototype():
    """out[q P + p] = sum_k h_p[k] x[q L - width + k]: the strided filter-bank conv1d of the restatement
    against numpy's plain full correlation sampled at the same instants (an independent evaluation of
    the same sums), for a decimating (48k), an interpolating (8k) and a rational (44.1k) ratio."""
    for orig, new, n in ((48000, 16000, 5000), (8000, 16000, 3000), (44100, 16000, 4410)):
        g = math.gcd(orig, new)
        L, P = orig // g, new // g
        kern, width = sinc_resample_kernel(orig, new, g, dtype=torch.float64)
        x = torch.randn(1, n, dtype=torch.float64, generator=torch.Generator().manual_seed(1))
        y = resample(x, orig, new)[0].numpy()
        xp = np.concatenate([np.zeros(width), x[0].numpy(), np.zeros(width + L)])
        K = kern.shape[-1]
        for p in range(0, P, max(1, P // 7)):
            full = np.convolve(xp, kern[p, 0].numpy()[::-1])       # full[i] = sum_k h[k] xp[i - (K-1) + k]
            q = np.arange((len(y) - p + P - 1) // P)
            assert np.abs(full[q * L + K - 1] - y[q * P + p]).max() < 1e-12, (orig, p)


def test_sine_survives_resampling():
    sr, new = 44100, 16000
    t = torch.arange(sr, dtype=torch.float32) / sr
    x = torch.sin(2 * math.pi * 1000.0 * t)[None]
    y = resample(x, sr, new)[0]
    tn = torch.arange(y.numel(), dtype=torch.float32) / new
    ref = torch.sin(2 * math.pi * 1000.0 * tn)
    assert (y[200:-200] - ref[200:-200]).abs().max() < 2e-3
