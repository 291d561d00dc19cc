"""In-memory audio front door (core/io.py:153-484, in-memory branch + plain WAV files).

The hot path of SURVEY.md section 8 takes `{"waveform": (channel, time) tensor, "sample_rate": int}`;
WAV files are read with scipy (no ffmpeg/torchcodec in scope).  Down-mixing is the channel mean
(io.py:223-265); resampling restates `torchaudio.functional.resample` (windowed-sinc polyphase bank,
io.py:258-262) with the filter bank evaluated on the host in the waveform's dtype and applied by
`pa_resample_poly` on the GPU (csrc/resample.hip)."""
from __future__ import annotations

import math
from io import IOBase
from pathlib import Path
from typing import Mapping, Optional, Tuple, Union

import numpy as np
import torch

from .core import Segment

AudioFile = Union[str, Path, IOBase, Mapping]


def sinc_resample_bank(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6,
                       rolloff: float = 0.99) -> Tuple[torch.Tensor, int, int, int]:
    """filter bank of torchaudio's "sinc_interp_hann" resampler for orig -> new (reduced by their gcd):
    (taps (P, K) float32, L input samples per block, P phases, width); K = 2 width + L.  Evaluated with
    float32 torch operations in torchaudio's order, so that the taps are the ones the reference uses."""
    g = math.gcd(int(orig_freq), int(new_freq))
    L, P = int(orig_freq) // g, int(new_freq) // g
    cutoff = min(L, P) * rolloff
    width = math.ceil(lowpass_filter_width * L / cutoff)
    grid = torch.arange(-width, width + L, dtype=torch.float32)[None, :] / L
    t = torch.arange(0, -P, -1, dtype=torch.float32)[:, None] / P + grid
    t *= cutoff
    t = t.clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t *= math.pi
    taps = torch.where(t == 0, torch.tensor(1.0), t.sin() / t)
    taps *= window * (cutoff / L)
    return taps.contiguous(), L, P, width


def resample_on_device(waveform: torch.Tensor, orig_freq: int, new_freq: int,
                       device: torch.device) -> torch.Tensor:
    """(channel, time) fp32 -> (channel, ceil(new * time / orig)) on `device` (core/io.py:258-262)"""
    from . import ffi
    taps, L, P, width = sinc_resample_bank(orig_freq, new_freq)
    lib = ffi.load()
    x = waveform.to(device, torch.float32).contiguous()
    n = x.shape[1]
    n_out = int(math.ceil(P * n / L))
    out = torch.empty((x.shape[0], n_out), dtype=torch.float32, device=device)
    tp = taps.to(device)
    with torch.cuda.device(device):
        for c in range(x.shape[0]):
            ffi.check(lib.pa_resample_poly(ffi.ptr(x[c]), n, ffi.ptr(tp), L, P, taps.shape[1], width,
                                           ffi.ptr(out[c]), n_out, ffi.stream()), "pa_resample_poly")
    return out


class Audio:
    def __init__(self, sample_rate: Optional[int] = None, mono: Optional[str] = None,
                 device: Optional[torch.device] = None):
        self.sample_rate = sample_rate
        self.mono = mono
        self.device = device      # where resampling runs (set by SpeakerDiarization.to)

    PRECISION = 0.001

    @staticmethod
    def power_normalize(waveform: torch.Tensor) -> torch.Tensor:
        """unit RMS along time (io.py:134-151)"""
        rms = waveform.square().mean(dim=-1, keepdim=True).sqrt()
        return waveform / (rms + 1e-8)

    @staticmethod
    def validate_file(file: AudioFile) -> Mapping:
        """io.py:153-216"""
        if isinstance(file, Mapping):
            pass
        elif isinstance(file, (str, Path)):
            file = {"audio": str(file), "uri": Path(file).stem}
        elif isinstance(file, IOBase):       # open("audio.wav", "rb"): io.py:180-181
            return {"audio": file, "uri": "stream"}
        else:
            raise ValueError("AudioFile must be a path, a binary file object or a mapping with "
                             "'waveform' or 'audio' keys")
        if "waveform" in file:
            waveform = file["waveform"]
            if len(waveform.shape) != 2 or waveform.shape[0] > waveform.shape[1]:
                raise ValueError("'waveform' must be provided as a (channel, time) torch Tensor.")
            if file.get("sample_rate", None) is None:
                raise ValueError("'waveform' must be provided with their 'sample_rate'.")
            file = dict(file)
            file.setdefault("uri", "waveform")
        elif "audio" in file:
            if isinstance(file["audio"], IOBase):
                return file
            if not Path(file["audio"]).is_file():
                raise ValueError(f"File {file['audio']} does not exist")
            file = dict(file)
            file.setdefault("uri", Path(file["audio"]).stem)
        else:
            raise ValueError("Neither 'waveform' nor 'audio' is available for this file.")
        return file

    def downmix_and_resample(self, waveform: torch.Tensor, sample_rate: int,
                             channel: Optional[int] = None) -> Tuple[torch.Tensor, int]:
        """io.py:223-265: channel selection, then down-mix, then resampling (on the GPU)."""
        if channel is not None:
            waveform = waveform[channel: channel + 1]
        num_channels = waveform.shape[0]
        if num_channels > 1:
            if self.mono == "random":
                waveform = waveform[np.random.randint(num_channels)][None]
            elif self.mono == "downmix":
                waveform = waveform.mean(dim=0, keepdim=True)
        if self.sample_rate is not None and self.sample_rate != sample_rate:
            device = self.device if self.device is not None else \
                (waveform.device if waveform.is_cuda else None)
            if device is None or device.type != "cuda":
                raise RuntimeError(f"resampling {sample_rate} Hz -> {self.sample_rate} Hz runs on the GPU: "
                                   "move the pipeline to a GPU first (there is no CPU path)")
            waveform = resample_on_device(waveform, sample_rate, self.sample_rate, device)
            sample_rate = self.sample_rate
        return waveform, sample_rate

    def get_num_samples(self, duration: float, sample_rate: Optional[int] = None) -> int:
        sample_rate = sample_rate or self.sample_rate
        if sample_rate is None:
            raise ValueError("`sample_rate` must be provided to compute number of samples.")
        return round(duration * sample_rate)

    def get_duration(self, file: AudioFile) -> float:
        file = self.validate_file(file)
        if "waveform" in file:
            return file["waveform"].shape[1] / file["sample_rate"]
        w, sr = self._read(file["audio"])
        return w.shape[1] / sr

    @staticmethod
    def _read_raw(path) -> Tuple[np.ndarray, int]:
        """the samples of a WAV file as stored ((n,) or (n, channels), int16 / int32 / uint8 / float) and its rate"""
        from scipy.io import wavfile
        if isinstance(path, IOBase):
            sr, data = wavfile.read(path)
            path.seek(0)                     # rewind, as the reference does after decoding (io.py:348-349)
        else:
            sr, data = wavfile.read(str(path))
        return data, int(sr)

    @staticmethod
    def _to_float(data: torch.Tensor) -> torch.Tensor:
        """stored samples -> float32 in [-1, 1), (channels, n).  Power-of-two scalings: the same bits on the host and
        on the device (int -> float32 conversion rounds the same way on both)."""
        if data.dtype == torch.int16:
            x = data.to(torch.float32).mul_(1.0 / 32768.0)
        elif data.dtype == torch.int32:
            x = data.to(torch.float32).mul_(1.0 / 2147483648.0)
        elif data.dtype == torch.uint8:
            x = data.to(torch.float32).sub_(128.0).mul_(1.0 / 128.0)
        else:
            x = data.to(torch.float32)
        return x[None] if x.dim() == 1 else x.t().contiguous()

    @staticmethod
    def _as_tensor(data: np.ndarray) -> torch.Tensor:
        arr = np.ascontiguousarray(data)
        if not arr.flags.writeable:          # (samples decoded from a file OBJECT are a read-only view of its buffer)
            arr = arr.copy()
        return torch.from_numpy(arr)

    @staticmethod
    def _read(path) -> Tuple[torch.Tensor, int]:
        data, sr = Audio._read_raw(path)
        return Audio._to_float(Audio._as_tensor(data)), sr

    def load_on_device(self, file: AudioFile, device: torch.device, raw=None) -> Tuple[torch.Tensor, int]:
        """`__call__` for a file on disk with the sample conversion ON `device`: the stored samples (half the bytes of
        their float32 form for 16-bit files) are copied as they are and scaled there -- the host never builds the
        float32 waveform (0.2 s per audio-hour of page faults and two passes over 230 MB).  Same values as `__call__`
        for mono and stereo files; files with more channels take the host path (a device-side mean over > 2 channels
        may associate differently).  `raw`: (samples, rate) already read by `_read_raw` (apply_batch reads one file
        ahead in a worker thread)."""
        file = self.validate_file(file)
        if "waveform" in file:
            return self(file)
        data, sample_rate = raw if raw is not None else self._read_raw(file["audio"])
        if data.ndim == 2 and data.shape[1] > 2:
            waveform = self._to_float(self._as_tensor(data))
        else:
            waveform = self._to_float(self._as_tensor(data).to(device))
        return self.downmix_and_resample(waveform, sample_rate, channel=file.get("channel", None))

    def __call__(self, file: AudioFile) -> Tuple[torch.Tensor, int]:
        """io.py:306-351"""
        file = self.validate_file(file)
        if "waveform" in file:
            waveform, sample_rate = file["waveform"], file["sample_rate"]
        else:
            waveform, sample_rate = self._read(file["audio"])
        return self.downmix_and_resample(waveform, sample_rate, channel=file.get("channel", None))

    def crop(self, file: AudioFile, segment: Segment, mode: str = "raise"
             ) -> Tuple[torch.Tensor, int]:
        """io.py:353-484: the excerpt is cut at the FILE's rate and only then down-mixed / resampled.
        `mode`: "raise" (out-of-bounds segments are an error) or "pad" (zero padding).

        In-memory waveforms (:384-413): samples [round(start sr), round(end sr)); in "raise" mode an end
        AT or beyond the last sample is an error (the reference compares with >=).  Files (:415-484): the
        reference asks torchcodec for the samples played in [start, end) and repairs a +-1 sample difference
        to round(duration sr); this build decodes PCM WAV completely (scipy) and applies the same
        arithmetic, with torchcodec's range rounded to the nearest sample at both ends."""
        file = self.validate_file(file)
        channel = file.get("channel", None)
        start, end = float(segment.start), float(segment.end)
        if "waveform" in file:
            waveform, sample_rate = file["waveform"], file["sample_rate"]
            total = waveform.shape[1]
            first, last = self.get_num_samples(start, sample_rate), self.get_num_samples(end, sample_rate)
            if mode == "raise":
                if first < 0:
                    raise ValueError(f"requested chunk with negative start time (t={start:.3f}s)")
                if last >= total:
                    raise ValueError(f"requested chunk with end time (t={end:.3f}s) greater than "
                                     f"{file.get('uri', 'in-memory')} file duration ({total / sample_rate:.3f}s).")
            lead, trail = max(0, -first), max(last, total) - total
            data = waveform[:, max(first, 0):min(last, total)]
            if lead or trail:
                data = torch.nn.functional.pad(data, (lead, trail))
            return self.downmix_and_resample(data, sample_rate, channel=channel)

        waveform, sample_rate = self._read(file["audio"])
        total = waveform.shape[1]
        duration = total / sample_rate
        lead = max(0, self.get_num_samples(-start, sample_rate))
        if start < 0:
            if mode == "raise":
                raise ValueError(f"requested chunk with negative start time (t={start:.3f}s)")
            start = 0.0
        trail = max(self.get_num_samples(end, sample_rate), total) - total
        if end > duration:
            if mode == "raise":
                raise ValueError(f"requested chunk with end time (t={end:.3f}s) greater than "
                                 f"{file.get('uri', 'in-memory')} file duration ({duration:.3f}s).")
            end = duration
        data = waveform[:, min(self.get_num_samples(start, sample_rate), total):
                        min(self.get_num_samples(end, sample_rate), total)]
        expected = self.get_num_samples(segment.duration, sample_rate)
        difference = lead + data.shape[1] + trail - expected
        if abs(difference) > 1:
            raise ValueError(f"requested chunk {segment} from {file.get('uri', 'in-memory')} file resulted in "
                             f"{data.shape[1]} samples instead of the expected {expected} samples.")
        if difference == 1:
            data = data[:, :-1]
        elif difference == -1:
            trail += 1
        if lead or trail:
            data = torch.nn.functional.pad(data, (lead, trail))
        return self.downmix_and_resample(data, sample_rate, channel=channel)
