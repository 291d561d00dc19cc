"""Multi-GPU sharding of the sliding-window front-end.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI; "gloo" on CPU for tests).
Every chunk is independent until clustering, so rank r processes the contiguous chunk range
[r*C/G, (r+1)*C/G) and the per-chunk results -- hard segmentations (uint8) and embeddings (fp32) --
are exchanged with ONE all-gather of a fused byte buffer (payload ~17 MB per audio-hour: latency bound
on the 7 x 153 GB/s xGMI mesh).  Counting, clustering and reconstruction then run redundantly on every
rank (no broadcast needed); the reference has no multi-GPU inference path at all."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist


@dataclass(frozen=True)
class Shard:
    rank: int = 0
    world_size: int = 1
    group: Optional[object] = None


_CURRENT = Shard()


def set_shard(shard: Optional[Shard]):
    """Select how `SpeakerDiarization.apply` splits one file across ranks (None = no sharding)."""
    global _CURRENT
    _CURRENT = shard or Shard()


def current_shard() -> Shard:
    return _CURRENT


def shard_from_env(group=None) -> Shard:
    if dist.is_available() and dist.is_initialized():
        return Shard(dist.get_rank(group), dist.get_world_size(group), group)
    return Shard()


def chunk_range(total_chunks: int, shard: Shard) -> Optional[Tuple[int, int]]:
    """contiguous range of rank `shard.rank` (None when not sharded): sizes differ by at most 1."""
    if shard.world_size == 1:
        return None
    base, rem = divmod(total_chunks, shard.world_size)
    begin = shard.rank * base + min(shard.rank, rem)
    return begin, begin + base + (1 if shard.rank < rem else 0)


def all_gather_chunks(seg_local: np.ndarray, emb_local: np.ndarray, total_chunks: int, shard: Shard,
                      device: torch.device):
    """(c_r, F, S) float32 {0,1} + (c_r, S, D) float32 per rank -> full (C, F, S), (C, S, D) on every
    rank, via a single all-gather of fixed-size byte records (ranks are padded to the largest share)."""
    F, S = seg_local.shape[1], seg_local.shape[2]
    D = emb_local.shape[2]
    base, rem = divmod(total_chunks, shard.world_size)
    cmax = base + (1 if rem else 0)
    rec = F * S + S * D * 4  # bytes per chunk: uint8 segmentation + fp32 embeddings
    buf = np.zeros((cmax, rec), dtype=np.uint8)
    c = seg_local.shape[0]
    buf[:c, :F * S] = seg_local.astype(np.uint8).reshape(c, -1)
    buf[:c, F * S:] = np.ascontiguousarray(emb_local, dtype=np.float32).reshape(c, -1).view(np.uint8)
    backend = dist.get_backend(shard.group)
    dev = device if (backend == "nccl" and device.type == "cuda") else torch.device("cpu")
    send = torch.from_numpy(buf).to(dev)
    recv = torch.empty((shard.world_size, cmax, rec), dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(recv.view(-1), send.view(-1), group=shard.group)
    recv = recv.cpu().numpy()
    segs, embs = [], []
    for r in range(shard.world_size):
        cr = base + (1 if r < rem else 0)
        segs.append(recv[r, :cr, :F * S].reshape(cr, F, S).astype(np.float32))
        embs.append(np.ascontiguousarray(recv[r, :cr, F * S:]).view(np.float32).reshape(cr, S, D))
    return np.concatenate(segs, 0), np.concatenate(embs, 0)
