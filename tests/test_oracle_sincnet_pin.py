"""Pins the oracle's ParamSincFB restatement (oracle/models.py) to the PUBLISHED definition of the SincNet
filter bank (Ravanelli & Bengio, "Speaker Recognition from Raw Waveform with SincNet", eq. 4-8) and to the
properties that definition implies -- asteroid_filterbanks 0.4.0 itself (the code the reference imports,
models/blocks/sincnet.py:36-50) is not available offline, so this is a pin against the paper, not
against that package:

  even ("cos") filter  g[n] = ( 2 f2 sinc(2 f2 t) - 2 f1 sinc(2 f1 t) ) w[n] / (2 (f2 - f1)),  t = n / sr
  odd  ("sin") filter  h[n] = ( cos(2 pi f1 t) - cos(2 pi f2 t) ) / (pi t) w[n] / (2 (f2 - f1))   (its quadrature pair)

with a Hamming window w, mel-spaced band edges f1 < f2 (f1 >= 50 Hz, f2 - f1 >= 50 Hz)."""
import numpy as np
import torch

from oracle.models import ParamSincFB


def _closed_form(fb: ParamSincFB):
    sr, K = fb.sample_rate, fb.kernel_size
    low = fb.min_low_hz + np.abs(fb.low_hz_.detach().double().numpy()[:, 0])
    high = np.clip(low + fb.min_band_hz + np.abs(fb.band_hz_.detach().double().numpy()[:, 0]), fb.min_low_hz, sr / 2)
    n = np.arange(K) - K // 2
    t = n / sr
    w = np.hamming(K)
    even, odd = [], []
    for f1, f2 in zip(low, high):
        g = 2 * f2 * np.sinc(2 * f2 * t) - 2 * f1 * np.sinc(2 * f1 * t)          # np.sinc = sin(pi x) / (pi x)
        with np.errstate(divide="ignore", invalid="ignore"):
            h = (np.cos(2 * np.pi * f1 * t) - np.cos(2 * np.pi * f2 * t)) / (np.pi * t)
        h[K // 2] = 0.0
        even.append(g * w / (2 * (f2 - f1)))
        odd.append(h * w / (2 * (f2 - f1)))
    return np.array(even), np.array(odd), low, high


def test_filters_equal_published_closed_form():
    fb = ParamSincFB()
    filt = fb.filters().detach().double().numpy()[:, 0]            # (80, 251): 40 even + 40 odd
    even, odd, _, _ = _closed_form(fb)
    # float32 evaluation (arguments up to ~400 rad) against the float64 closed form: |diff| <= 1e-5 of the peak
    scale = np.abs(even).max()
    np.testing.assert_allclose(filt[:40], even, rtol=0, atol=1e-5 * scale)
    np.testing.assert_allclose(filt[40:], odd, rtol=0, atol=1e-5 * scale)


def test_mel_spaced_band_edges():
    fb = ParamSincFB()
    _, _, low, high = _closed_form(fb)
    mel = lambda hz: 2595 * np.log10(1 + hz / 700)
    edges = np.concatenate([low - fb.min_low_hz, [low[-1] - fb.min_low_hz + (high[-1] - low[-1] - fb.min_band_hz)]])
    steps = np.diff(mel(edges))
    assert np.allclose(steps, steps[0], rtol=1e-4)                  # equally spaced on the mel scale
    assert abs(edges[0] - 30.0) < 1e-3 and np.all(high - low >= fb.min_band_hz) and np.all(low >= fb.min_low_hz)


def test_band_pass_and_quadrature_properties():
    fb = ParamSincFB()
    filt = fb.filters().detach().double().numpy()[:, 0]
    _, _, low, high = _closed_form(fb)
    sr, K = fb.sample_rate, fb.kernel_size
    n = np.arange(K) - K // 2
    for k in (3, 12, 25, 38):                                       # narrow low bands ... wide high bands
        f1, f2 = low[k], high[k]
        band = f2 - f1
        gain = sr / (2 * band)                                      # pass-band gain of the normalised filter
        def response(h, f):
            return np.sum(h * np.exp(-2j * np.pi * f * n / sr))
        mid = response(filt[k], 0.5 * (f1 + f2))
        assert abs(mid.imag) < 1e-6 * gain and 0.7 * gain < mid.real < 1.3 * gain      # even filter: real response
        far = [f for f in (f1 - 6 * band - 300, f2 + 6 * band + 300) if 0 < f < sr / 2]
        for f in far:
            assert abs(response(filt[k], f)) < 0.05 * gain
        odd_mid = response(filt[40 + k], 0.5 * (f1 + f2))
        assert abs(odd_mid.real) < 1e-6 * gain                                           # odd filter: imaginary
        assert abs(abs(odd_mid) - abs(mid)) < 0.15 * gain                                 # quadrature pair
