"""In-memory audio front door (core/io.py:153-484, in-memory branch + plain WAV files).

The hot path of SURVEY.md section 8 takes `{"waveform": (channel, time) tensor, "sample_rate": int}`;
WAV files are read with scipy (no ffmpeg/torchcodec in scope).  Down-mixing is the channel mean
(io.py:223-265); resampling (torchaudio in the reference) uses scipy's polyphase filter and is a
convenience outside the parity contract."""
from __future__ import annotations

import math
from pathlib import Path
from typing import Mapping, Optional, Tuple, Union

import numpy as np
import torch

from .core import Segment

AudioFile = Union[str, Path, Mapping]


class Audio:
    def __init__(self, sample_rate: Optional[int] = None, mono: Optional[str] = None):
        self.sample_rate = sample_rate
        self.mono = mono

    @staticmethod
    def validate_file(file: AudioFile) -> Mapping:
        """io.py:153-216"""
        if isinstance(file, Mapping):
            pass
        elif isinstance(file, (str, Path)):
            file = {"audio": str(file), "uri": Path(file).stem}
        else:
            raise ValueError("AudioFile must be a path or a mapping with 'waveform' or 'audio' keys")
        if "waveform" in file:
            waveform = file["waveform"]
            if len(waveform.shape) != 2 or waveform.shape[0] > waveform.shape[1]:
                raise ValueError("'waveform' must be provided as a (channel, time) torch Tensor.")
            if file.get("sample_rate", None) is None:
                raise ValueError("'waveform' must be provided with their 'sample_rate'.")
            file = dict(file)
            file.setdefault("uri", "waveform")
        elif "audio" in file:
            if not Path(file["audio"]).is_file():
                raise ValueError(f"File {file['audio']} does not exist")
            file = dict(file)
            file.setdefault("uri", Path(file["audio"]).stem)
        else:
            raise ValueError("Neither 'waveform' nor 'audio' is available for this file.")
        return file

    def downmix_and_resample(self, waveform: torch.Tensor, sample_rate: int
                             ) -> Tuple[torch.Tensor, int]:
        num_channels = waveform.shape[0]
        if num_channels > 1:
            if self.mono == "random":
                waveform = waveform[np.random.randint(num_channels)][None]
            elif self.mono == "downmix":
                waveform = waveform.mean(dim=0, keepdim=True)
        if self.sample_rate is not None and self.sample_rate != sample_rate:
            from scipy.signal import resample_poly
            g = math.gcd(int(self.sample_rate), int(sample_rate))
            y = resample_poly(waveform.cpu().numpy().astype(np.float64), self.sample_rate // g,
                              sample_rate // g, axis=-1)
            waveform = torch.from_numpy(y.astype(np.float32))
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
    def _read(path) -> Tuple[torch.Tensor, int]:
        from scipy.io import wavfile
        sr, data = wavfile.read(str(path))
        if data.dtype == np.int16:
            x = data.astype(np.float32) / 32768.0
        elif data.dtype == np.int32:
            x = data.astype(np.float32) / 2147483648.0
        elif data.dtype == np.uint8:
            x = (data.astype(np.float32) - 128.0) / 128.0
        else:
            x = data.astype(np.float32)
        if x.ndim == 1:
            x = x[None]
        else:
            x = x.T
        return torch.from_numpy(np.ascontiguousarray(x)), int(sr)

    def __call__(self, file: AudioFile) -> Tuple[torch.Tensor, int]:
        """io.py:306-351"""
        file = self.validate_file(file)
        if "waveform" in file:
            waveform, sample_rate = file["waveform"], file["sample_rate"]
        else:
            waveform, sample_rate = self._read(file["audio"])
        channel = file.get("channel", None)
        if channel is not None:
            waveform = waveform[channel: channel + 1]
        return self.downmix_and_resample(waveform, sample_rate)

    def crop(self, file: AudioFile, segment: Segment, mode: str = "raise"
             ) -> Tuple[torch.Tensor, int]:
        """io.py:353-484 (in-memory): fixed-size excerpt, zero padded when mode == 'pad'."""
        waveform, sample_rate = self(file)
        frames = waveform.shape[1]
        start_frame = math.floor(segment.start * sample_rate)
        num_frames = math.floor(segment.duration * sample_rate)
        end_frame = start_frame + num_frames
        if mode == "raise":
            if num_frames > frames:
                raise ValueError("requested fixed duration is longer than file duration")
            if end_frame > frames + math.ceil(0.001 * sample_rate):
                raise ValueError("requested chunk lies outside of file bounds")
            end_frame = min(end_frame, frames)
            start_frame = end_frame - num_frames
            pad_start = pad_end = 0
        else:
            pad_start = -min(0, start_frame)
            pad_end = max(end_frame, frames) - frames
            start_frame = max(0, start_frame)
            end_frame = min(end_frame, frames)
        data = waveform[:, start_frame:end_frame]
        if pad_start or pad_end:
            data = torch.nn.functional.pad(data, (pad_start, pad_end))
        return data, sample_rate
