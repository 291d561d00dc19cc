"""Oracle (CPU) restatement of `torchaudio.functional.resample` as the reference calls it
(core/io.py:258-262: `torchaudio.functional.resample(waveform, sample_rate, self.sample_rate)`, i.e.
lowpass_filter_width=6, rolloff=0.99, resampling_method="sinc_interp_hann").

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  torchaudio 2.10.0 (uv.lock) is a third-party
dependency that is neither under /root/reference nor installed here, and no test of the reference pins
resampled samples: PARITY UNPINNED.  The restatement follows torchaudio's published algorithm
(`_get_sinc_resample_kernel` + `_apply_sinc_resample_kernel`): a bank of `new/gcd` windowed-sinc
filters of 2*width + orig/gcd taps, evaluated in the waveform's dtype, applied as a strided conv1d on the
zero-padded signal, output cropped to ceil(new * length / orig) samples.  tests/test_resample_cpu.py
checks what CAN be checked without torchaudio: the filter bank against a float64 closed form, the output
length rule, and agreement with scipy.signal.resample_poly driven by the same prototype filter."""
from __future__ import annotations

import math

import torch


def sinc_resample_kernel(orig_freq: int, new_freq: int, gcd: int, lowpass_filter_width: int = 6,
                         rolloff: float = 0.99, dtype=torch.float32):
    """-> (kernels (new/gcd, 1, 2*width + orig/gcd), width)"""
    orig = int(orig_freq) // gcd
    new = int(new_freq) // gcd
    base_freq = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base_freq)
    idx = torch.arange(-width, width + orig, dtype=dtype)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=dtype)[:, None, None] / new + idx
    t *= base_freq
    t = t.clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2      # sinc_interp_hann
    t *= math.pi
    scale = base_freq / orig
    kernels = torch.where(t == 0, torch.tensor(1.0).to(t), t.sin() / t)
    kernels *= window * scale
    return kernels, width


def resample(waveform: torch.Tensor, orig_freq: int, new_freq: int) -> torch.Tensor:
    """(..., time) -> (..., ceil(new * time / orig))"""
    if orig_freq == new_freq:
        return waveform
    gcd = math.gcd(int(orig_freq), int(new_freq))
    kernel, width = sinc_resample_kernel(orig_freq, new_freq, gcd, dtype=waveform.dtype)
    orig = int(orig_freq) // gcd
    new = int(new_freq) // gcd
    shape = waveform.size()
    wav = waveform.reshape(-1, shape[-1])
    num_wavs, length = wav.shape
    wav = torch.nn.functional.pad(wav, (width, width + orig))
    out = torch.nn.functional.conv1d(wav[:, None], kernel, stride=orig)
    out = out.transpose(1, 2).reshape(num_wavs, -1)
    target_length = int(math.ceil(new * length / orig))
    out = out[..., :target_length]
    return out.view(shape[:-1] + out.shape[-1:])
