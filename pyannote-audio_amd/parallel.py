"""Multi-GPU sharding of the sliding-window front-end.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI; "gloo" on CPU for tests).
Every chunk is independent until clustering, so the per-chunk results -- hard segmentations (uint8)
and embeddings (fp32) -- travel as fixed-size byte RECORDS in ONE all-gather (17 MB per audio-hour:
latency bound on the 7 x 153 GB/s xGMI mesh).  Two partitions share the record format:

  * one long file: rank r processes the contiguous chunk range [r*C/G, (r+1)*C/G) (`all_gather_chunks`);
  * many files (BASELINE.json configs[4]): every rank diarizes its own files and all ranks exchange the
    records of ALL files for one joint clustering (`all_gather_files`).

The send buffer is assembled on the device from the tensors the kernels produced and, with RCCL, never
leaves HBM; with gloo (CPU tests) it is staged through host memory because gloo cannot read device
pointers.  Counting, clustering and reconstruction then run redundantly on every rank (no broadcast);
the reference has no multi-GPU inference path at all."""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

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


# ---------------------------------------------------------------------------------------------------
# record format: per chunk, F*S bytes of {0,1} segmentation followed by S*D fp32 embeddings (bytes)
# ---------------------------------------------------------------------------------------------------
def pack_records(seg: torch.Tensor, emb: torch.Tensor) -> torch.Tensor:
    """(C, F, S) uint8/float {0,1} + (C, S, D) fp32, same device -> (C, F*S + 4*S*D) uint8."""
    C = seg.shape[0]
    seg_b = seg.to(torch.uint8).reshape(C, -1)
    emb_b = emb.to(torch.float32).contiguous().view(torch.uint8).reshape(C, -1)
    return torch.cat([seg_b, emb_b], dim=1).contiguous()


def unpack_records(rec: torch.Tensor, F: int, S: int, D: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """inverse of `pack_records` (views where possible)."""
    C = rec.shape[0]
    seg = rec[:, :F * S].reshape(C, F, S)
    emb = rec[:, F * S:].contiguous().view(torch.float32).reshape(C, S, D)
    return seg, emb


def _wire_device(shard: Shard, device: torch.device) -> torch.device:
    backend = dist.get_backend(shard.group)
    return device if (backend == "nccl" and device.type == "cuda") else torch.device("cpu")


def all_gather_records(send: torch.Tensor, shard: Shard) -> torch.Tensor:
    """(c, R) uint8 on `send.device` -> (world, c, R) on the same device; c equal on all ranks."""
    wire = _wire_device(shard, send.device)
    buf = send if send.device == wire else send.to(wire)
    recv = torch.empty((shard.world_size,) + tuple(buf.shape), dtype=torch.uint8, device=wire)
    dist.all_gather_into_tensor(recv.view(-1), buf.contiguous().view(-1), group=shard.group)
    return recv if wire == send.device else recv.to(send.device)


def all_gather_chunks(seg_local, emb_local, total_chunks: int, shard: Shard, device: torch.device):
    """One file sharded by chunk range.  (c_r, F, S) {0,1} + (c_r, S, D) fp32 per rank (device tensors
    or host arrays) -> the full (C, F, S) uint8 and (C, S, D) fp32 DEVICE tensors on every rank, via a
    single all-gather of fixed-size records (ranks are padded to the largest share)."""
    seg_t = torch.as_tensor(seg_local).to(device)
    emb_t = torch.as_tensor(emb_local).to(device)
    F, S, D = seg_t.shape[1], seg_t.shape[2], emb_t.shape[2]
    base, rem = divmod(total_chunks, shard.world_size)
    cmax = base + (1 if rem else 0)
    rec = pack_records(seg_t, emb_t)
    if rec.shape[0] < cmax:
        rec = torch.cat([rec, rec.new_zeros((cmax - rec.shape[0], rec.shape[1]))], dim=0)
    recv = all_gather_records(rec, shard)
    parts = [recv[r, :base + (1 if r < rem else 0)] for r in range(shard.world_size)]
    return unpack_records(torch.cat(parts, dim=0), F, S, D)


def all_gather_files(records: Sequence[torch.Tensor], shard: Shard, device: torch.device
                     ) -> List[List[torch.Tensor]]:
    """Many files.  `records`: this rank's per-file (C_f, R) uint8 record tensors.  Returns, on every
    rank, result[r][j] = records of the j-th file of rank r.  Two collectives: a tiny all-gather of the
    chunk counts, then ONE all-gather of the concatenated records padded to the largest rank."""
    counts = torch.tensor([r.shape[0] for r in records], dtype=torch.int64)
    nfiles = torch.tensor([len(records)], dtype=torch.int64)
    wire = _wire_device(shard, device)
    all_n = torch.empty(shard.world_size, dtype=torch.int64, device=wire)
    dist.all_gather_into_tensor(all_n, nfiles.to(wire), group=shard.group)
    fmax = int(all_n.max().item())
    cnt = torch.zeros(fmax, dtype=torch.int64)
    cnt[:len(records)] = counts
    all_cnt = torch.empty((shard.world_size, fmax), dtype=torch.int64, device=wire)
    dist.all_gather_into_tensor(all_cnt.view(-1), cnt.to(wire), group=shard.group)
    all_cnt = all_cnt.cpu()
    cmax = int(all_cnt.sum(dim=1).max().item())
    R = records[0].shape[1] if len(records) else 0
    Rt = torch.tensor([R], dtype=torch.int64)
    all_R = torch.empty(shard.world_size, dtype=torch.int64, device=wire)
    dist.all_gather_into_tensor(all_R, Rt.to(wire), group=shard.group)
    R = int(all_R.max().item())
    mine = torch.cat(list(records), dim=0) if len(records) else torch.zeros((0, R), dtype=torch.uint8,
                                                                           device=device)
    if mine.shape[0] < cmax:
        mine = torch.cat([mine, mine.new_zeros((cmax - mine.shape[0], R))], dim=0)
    recv = all_gather_records(mine.to(device), shard)
    out: List[List[torch.Tensor]] = []
    for r in range(shard.world_size):
        files, pos = [], 0
        for j in range(int(all_n[r].item())):
            c = int(all_cnt[r, j].item())
            files.append(recv[r, pos:pos + c])
            pos += c
        out.append(files)
    return out
