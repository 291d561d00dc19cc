"""Device-side PyanNet: the b2 interface of SURVEY.md section 8b (`Model.forward`) over
`pa_seg_forward`.  Mirrors models/segmentation/PyanNet.py:211-240 + core/inference.py:182-215."""
from __future__ import annotations

import torch

from . import ffi
from .weights import SegmentationPack


def num_frames(num_samples: int, stride: int = 10) -> int:
    """SincNet.num_frames (models/blocks/sincnet.py:82-107)."""
    n = num_samples
    for k, s in zip([251, 3, 5, 3, 5, 3], [stride, 3, 1, 3, 1, 3]):
        if n < k:
            return 0
        n = 1 + (n - k) // s
    return n


class SegmentationEngine:
    """Runs the segmentation network over strided chunks of a device-resident waveform.

    `max_chunks` bounds one launch group (workspace ~5.9 MB per 10 s chunk); the LSTM recurrence
    wants >= 2048 chunks in flight to fill 256 CUs (one workgroup per 16-chunk tile and direction)."""

    def __init__(self, pack: SegmentationPack, max_chunks: int = 4096):
        self.pack = pack
        self.max_chunks = max_chunks
        self._ws = None

    def _workspace(self, nbytes: int) -> torch.Tensor:
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = None
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.pack.device)
        return self._ws

    def release_workspace(self):
        self._ws = None

    @ffi.on_device(lambda self, *a, **k: self.pack.device)
    def forward_strided(self, wav: torch.Tensor, chunk_stride: int, num_chunks: int, num_samples: int,
                        want_logp: bool = True, want_multilabel: bool = True):
        """wav: 1-D fp32 device tensor; chunk c = wav[c*stride : c*stride + num_samples] (zero padded
        past the end).  Returns (logp (C,F,K) fp32 | None, multilabel (C,F,S) uint8 | None); for a
        multi-label (non-powerset) checkpoint the first item holds the sigmoid SCORES and there is no hard
        output here (hysteresis thresholding is a pipeline stage: frames.binarize)."""
        lib = ffi.load()
        w = self.pack.struct
        assert wav.dim() == 1 and wav.dtype == torch.float32
        F = self.frames_of(num_samples)
        if F <= 0:
            raise ValueError(f"chunks of {num_samples} samples are too short for this model")
        dev = self.pack.device
        if not self.pack.powerset:
            want_logp, want_multilabel = True, False
        logp = torch.empty((num_chunks, F, w.num_classes), dtype=torch.float32, device=dev) \
            if want_logp else None
        ml = torch.empty((num_chunks, F, w.num_speakers), dtype=torch.uint8, device=dev) \
            if want_multilabel else None
        c0 = 0
        while c0 < num_chunks:
            nb = min(self.max_chunks, num_chunks - c0)
            need = self._workspace_bytes(lib, w, nb, num_samples, chunk_stride)
            ws = self._workspace(need)
            off = c0 * chunk_stride
            sub = wav[off:] if off < wav.numel() else wav[:0]
            rc = self._launch(
                lib, w, ffi.c_fp(sub.data_ptr()) if sub.numel() else ffi.c_fp(wav.data_ptr()),
                sub.numel(), chunk_stride, nb, num_samples,
                ffi.ptr(logp[c0:c0 + nb]) if logp is not None else None,
                ffi.ptr(ml[c0:c0 + nb]) if ml is not None else None, ws, F)
            ffi.check(rc, "segmentation forward")
            c0 += nb
        return logp, ml

    # -- model-specific entry points (PyanNet here; SSeRiouSSEngine overrides them)
    def frames_of(self, num_samples: int) -> int:
        return num_frames(num_samples, int(self.pack.struct.sinc_stride) or 10)

    def _workspace_bytes(self, lib, w, nb, num_samples, chunk_stride):
        return lib.pa_seg_workspace_bytes_strided(w, nb, num_samples, chunk_stride)

    def _launch(self, lib, w, wav_ptr, wav_len, chunk_stride, nb, num_samples, logp_ptr, ml_ptr, ws, F):
        return lib.pa_seg_forward(w, wav_ptr, wav_len, chunk_stride, nb, num_samples, logp_ptr, ml_ptr,
                                  ffi.ptr(ws), ws.numel(), ffi.stream())

    def forward(self, waveforms: torch.Tensor) -> torch.Tensor:
        """(B, 1, N) -> (B, F, K) log-probabilities (the reference `Model.forward` contract)."""
        B, ch, N = waveforms.shape
        assert ch == 1
        x = waveforms.to(self.pack.device, torch.float32).contiguous().view(-1)
        logp, _ = self.forward_strided(x, N, B, N, want_logp=True, want_multilabel=False)
        return logp


class SSeRiouSSEngine(SegmentationEngine):
    """SSeRiouSS (models/segmentation/SSeRiouSS.py:289-328) over `pa_sser_forward`: wav2vec 2.0 / WavLM encoder
    -> layer mix -> bi-LSTM stack -> head, on strided chunks of a device-resident waveform.  The feature
    extractor's first stages are large (65 MB per 10 s chunk at 512 channels): 16 chunks per launch group."""

    def __init__(self, pack, max_chunks: int = 16):
        super().__init__(pack, max_chunks=max_chunks)

    def frames_of(self, num_samples: int) -> int:
        return ffi.load().pa_sser_num_frames(self.pack.struct, num_samples)

    def _workspace_bytes(self, lib, w, nb, num_samples, chunk_stride):
        return lib.pa_sser_workspace_bytes(w, nb, num_samples)

    def _launch(self, lib, w, wav_ptr, wav_len, chunk_stride, nb, num_samples, logp_ptr, ml_ptr, ws, F):
        bias = self.pack.relative_bias(F)
        return lib.pa_sser_forward(w, wav_ptr, wav_len, chunk_stride, nb, num_samples,
                                   ffi.ptr(bias) if bias is not None else None, logp_ptr, ml_ptr,
                                   ffi.ptr(ws), ws.numel(), ffi.stream())
