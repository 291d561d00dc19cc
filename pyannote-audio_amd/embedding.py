"""Device-side WeSpeaker ResNet34 embedding extractor over `pa_emb_forward`
(mirrors models/embedding/wespeaker/__init__.py:324-343 and the b3 interface of SURVEY.md 8b)."""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.nn.functional as F

from . import ffi
from .weights import EmbeddingPack, XVectorPack


class EmbeddingEngine:
    """fbank -> ResNet34 -> weighted statistics pooling -> Linear over strided chunks of a
    device-resident waveform.  The backbone runs once per chunk; all S masks are pooled from it."""

    def __init__(self, pack: EmbeddingPack, max_chunks: Optional[int] = None):
        self.pack = pack
        # chunks per launch group: ~31 MB of activations per 10 s chunk -> 62 GB at 2 048 of the 288 GB.  Measured
        # per audio-hour (profiles/r3_emb_batch_sweep.txt): 64 -> 1 131 ms, 128 -> 1 068, 256 -> 1 039,
        # 512 -> 1 023 / 1 013 (split evenly), 1 024 -> 1 007, 2 048 -> 1 001, the whole file in one group (111 GB)
        # -> 1 000: every launch pays a pipeline fill and a tail, so fewer and longer launches win (Winograd
        # kernel 0.55 -> 0.62 of peak); groups sized to the 256-MiB Infinity Cache (8 chunks, so that a convolution
        # would read its predecessor's output on-die) are far on the wrong side of that trade.  The chunks of a
        # file are split EVENLY over the groups (3 591 = 2 x 1 796).
        # Round 6: the limit is a WORKSPACE SIZE, not a chunk count -- all 10 000 segments of 3 s in one group:
        # BASELINE.json configs[2] 16 001 -> 16 529 segments/s (2 048 per group: 115 launches of the F(4x4) kernel; 3 334:
        # 69, 16 195; 5 000: 46, 16 419; 10 000: 23 launches).  First 96 GB (a one-hour file = 2 x 1 796 chunks), then 128 GB:
        # the 3 591 chunks of a one-hour file in ONE group (111 GB) measured 757.1 / 757.6 ms per file against 760.6 / 765.6
        # as two groups and 766.6 / 769.9 as three (A/B pairs in one call, profiles/r6_emb_group_ab.txt).
        # `max_chunks` / PA_EMB_BATCH (a chunk count) overrides it.
        env = os.environ.get("PA_EMB_BATCH")
        self.max_chunks = max_chunks or (int(env) if env else None)
        self._ws = None
        self._idx_cache: dict = {}

    # share of the device's FREE memory one launch group's workspace may take (several pipelines / processes on one
    # GPU, or a smaller device, must not run out where the 2 048-chunk default asks for ~62 GB)
    MAX_FREE_FRACTION = 0.5
    #: workspace of one launch group when no chunk count is given (bytes)
    GROUP_WORKSPACE_BYTES = 128 << 30

    def _group_size(self, num_chunks: int, num_samples: int, S: int) -> int:
        """chunks per launch group: `max_chunks` (or as many as GROUP_WORKSPACE_BYTES hold), the file split EVENLY over
        the groups, and halved until the workspace fits MAX_FREE_FRACTION of what is free right now (a cached workspace
        counts as free)."""
        lib = ffi.load()
        limit = self.max_chunks
        if limit is None:
            # (the workspace is linear in the chunk count up to alignment: measured on 64 chunks)
            per_chunk = lib.pa_emb_workspace_bytes(self.pack.struct, 64, num_samples, S) / 64.0
            limit = max(8, int(self.GROUP_WORKSPACE_BYTES / max(per_chunk, 1.0)))
        groups = -(-num_chunks // limit)
        per_group = -(-num_chunks // groups)
        free, _ = torch.cuda.mem_get_info(self.pack.device)
        # + what torch's caching allocator holds without using it, + this engine's own cached workspace
        free += torch.cuda.memory_reserved(self.pack.device) - torch.cuda.memory_allocated(self.pack.device)
        free += self._ws.numel() if self._ws is not None else 0
        while per_group > 8 and lib.pa_emb_workspace_bytes(self.pack.struct, per_group, num_samples, S) > \
                self.MAX_FREE_FRACTION * free:
            groups *= 2
            per_group = -(-num_chunks // groups)
        return per_group

    def _workspace(self, nbytes: int) -> torch.Tensor:
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = None
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.pack.device)
        return self._ws

    def release_workspace(self):
        self._ws = None

    def num_pool_frames(self, num_samples: int) -> int:
        return ffi.load().pa_emb_num_pool_frames(self.pack.struct, num_samples)

    def nearest_index(self, mask_frames: int, pool_frames: int) -> torch.Tensor:
        """source index of F.interpolate(mode="nearest") (blocks/pooling.py:113-117): obtained by
        running torch's own op on an index ramp, so it is exactly what the reference does."""
        key = (mask_frames, pool_frames)
        if key not in self._idx_cache:
            ramp = torch.arange(mask_frames, dtype=torch.float32).view(1, 1, -1)
            idx = F.interpolate(ramp, size=pool_frames, mode="nearest").view(-1).to(torch.int32)
            self._idx_cache[key] = idx.to(self.pack.device)
        return self._idx_cache[key]

    @ffi.on_device(lambda self, *a, **k: self.pack.device)
    def forward_strided(self, wav: torch.Tensor, chunk_stride: int, num_chunks: int, num_samples: int,
                        masks: torch.Tensor | None = None) -> torch.Tensor:
        """wav: 1-D fp32 device tensor; masks: (C, S, Fm) fp32 device or None -> (C, S, D) fp32."""
        lib = ffi.load()
        w = self.pack.struct
        dev = self.pack.device
        if num_samples < 400:
            raise ValueError("chunk shorter than one fbank frame (400 samples)")
        S = 1 if masks is None else masks.shape[1]
        Fm = 0 if masks is None else masks.shape[2]
        Tp = self.num_pool_frames(num_samples)
        idx = self.nearest_index(Fm, Tp) if masks is not None else None
        if masks is not None:
            masks = masks.to(dev, torch.float32).contiguous()
        emb = torch.empty((num_chunks, S, w.embed_dim), dtype=torch.float32, device=dev)
        c0 = 0
        per_group = self._group_size(num_chunks, num_samples, S)
        while c0 < num_chunks:
            nb = min(per_group, num_chunks - c0)
            ws = self._workspace(lib.pa_emb_workspace_bytes(w, nb, num_samples, S))
            off = c0 * chunk_stride
            sub = wav[off:]
            rc = lib.pa_emb_forward(
                w, ffi.c_fp(sub.data_ptr()), sub.numel(), chunk_stride, nb, num_samples,
                ffi.ptr(masks[c0:c0 + nb]) if masks is not None else None, S, Fm,
                ffi.ptr(idx) if idx is not None else None, ffi.ptr(emb[c0:c0 + nb]),
                ffi.ptr(ws), ws.numel(), ffi.stream())
            ffi.check(rc, "pa_emb_forward")
            c0 += nb
        return emb

    def forward(self, waveforms: torch.Tensor, weights: torch.Tensor | None = None) -> torch.Tensor:
        """(B,1,N) [, (B,Fm) or (B,S,Fm)] -> (B,D) or (B,S,D): the reference forward contract."""
        B, ch, N = waveforms.shape
        assert ch == 1
        x = waveforms.to(self.pack.device, torch.float32).contiguous().view(-1)
        squeeze = weights is None or weights.dim() == 2
        m = None
        if weights is not None:
            m = weights.unsqueeze(1) if weights.dim() == 2 else weights
        emb = self.forward_strided(x, N, B, N, m)
        return emb[:, 0] if squeeze else emb


class XVectorEngine(EmbeddingEngine):
    """XVectorSincNet (models/embedding/xvector.py:205-349) over `pa_xvec_forward`: SincNet -> 5 TDNN layers
    -> weighted statistics pooling -> Linear; same strided-chunk / all-masks-at-once interface."""

    def __init__(self, pack: XVectorPack, max_chunks: Optional[int] = None):
        self.pack = pack
        self.max_chunks = max_chunks or 512      # ~5.3 MB of activations per 10 s chunk
        self._ws = None
        self._idx_cache = {}

    def num_pool_frames(self, num_samples: int) -> int:
        return ffi.load().pa_xvec_num_frames(self.pack.struct, num_samples)

    @ffi.on_device(lambda self, *a, **k: self.pack.device)
    def forward_strided(self, wav: torch.Tensor, chunk_stride: int, num_chunks: int, num_samples: int,
                        masks: torch.Tensor | None = None) -> torch.Tensor:
        lib = ffi.load()
        w = self.pack.struct
        dev = self.pack.device
        Tp = self.num_pool_frames(num_samples)
        if Tp < 1:
            raise ValueError(f"chunks of {num_samples} samples are too short for SincNet + the TDNN stack")
        S = 1 if masks is None else masks.shape[1]
        Fm = 0 if masks is None else masks.shape[2]
        idx = self.nearest_index(Fm, Tp) if masks is not None else None
        if masks is not None:
            masks = masks.to(dev, torch.float32).contiguous()
        emb = torch.empty((num_chunks, S, w.dimension), dtype=torch.float32, device=dev)
        c0 = 0
        while c0 < num_chunks:
            nb = min(self.max_chunks, num_chunks - c0)
            ws = self._workspace(lib.pa_xvec_workspace_bytes(w, nb, num_samples, S))
            sub = wav[c0 * chunk_stride:]
            rc = lib.pa_xvec_forward(
                w, ffi.c_fp(sub.data_ptr()), sub.numel(), chunk_stride, nb, num_samples,
                ffi.ptr(masks[c0:c0 + nb]) if masks is not None else None, S, Fm,
                ffi.ptr(idx) if idx is not None else None, ffi.ptr(emb[c0:c0 + nb]),
                ffi.ptr(ws), ws.numel(), ffi.stream())
            ffi.check(rc, "pa_xvec_forward")
            c0 += nb
        return emb
