"""Multi-GPU sharding of the sliding-window front-end.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI; "gloo" on CPU for tests).
Every chunk is independent until clustering, so the per-chunk results -- hard segmentations (uint8)
and embeddings (fp32) -- travel as fixed-size byte RECORDS in ONE all-gather (17 MB per audio-hour:
latency bound on the 7 x 153 GB/s xGMI mesh).  Two partitions share the record format:

  * one long file: rank r processes the contiguous chunk range [r*C/G, (r+1)*C/G) (`all_gather_chunks`);
  * many files (BASELINE.json configs[4]): every rank diarizes its own files and all ranks exchange the
    records of ALL files for one joint clustering (`all_gather_files`: ONE all-gather per job in steady state --
    the per-file chunk counts travel in a fixed header inside the same buffer).

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
    # (a copy in canonical strides: the slice of a ONE-chunk file is "contiguous" as it stands, at an unaligned offset)
    emb_b = rec[:, F * S:].clone(memory_format=torch.contiguous_format)
    return seg, emb_b.view(torch.float32).reshape(C, S, D)


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


def agree_all(flag: bool, shard: Shard, device: torch.device) -> bool:
    """True iff `flag` is true on EVERY rank of `shard` (one all-reduce(MIN) on the shard's own group): how ranks choose
    between code paths that issue different collectives, so that a rank with another environment or a local failure
    cannot take one schedule while the others wait for it inside the other."""
    if shard.world_size == 1:
        return bool(flag)
    wire = _wire_device(shard, device)
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=wire)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=shard.group)
    return bool(int(t.item()))


def broadcast_object(obj, src: int, shard: Shard, group, device: torch.device):
    """a small picklable object (the cluster labels and centroids of a joint job: numpy arrays) from rank `src` to
    every rank of `group`: its size, then its bytes (two broadcasts; through device memory when `group` is an RCCL
    group, through host memory -- where the labels live anyway -- when it is a gloo group).  The ranks trust each
    other: this is the process group of ONE job."""
    import io
    import pickle
    wire = device if (dist.get_backend(group) == "nccl" and device.type == "cuda") else torch.device("cpu")
    if shard.rank == src:
        buf = io.BytesIO()
        pickle.dump(obj, buf, protocol=pickle.HIGHEST_PROTOCOL)
        payload = torch.frombuffer(bytearray(buf.getvalue()), dtype=torch.uint8)
        size = torch.tensor([payload.numel()], dtype=torch.int64, device=wire)
    else:
        payload = None
        size = torch.zeros(1, dtype=torch.int64, device=wire)
    dist.broadcast(size, src=src, group=group)
    n = int(size.item())
    data = payload.to(wire) if shard.rank == src else torch.empty(n, dtype=torch.uint8, device=wire)
    dist.broadcast(data, src=src, group=group)
    if shard.rank == src:
        return obj
    return pickle.loads(data.cpu().numpy().tobytes())


# ---------------------------------------------------------------------------------------------------
# many files per rank: ONE all-gather of {header, records} buffers whose size every rank derives from what it
# has SEEN in earlier headers -- never from local information, so no rank can leave the collective alone
# ---------------------------------------------------------------------------------------------------
_HEADER_FIXED = 4                 # int64 words in front of the per-file counts: flag, files, chunks, record bytes
_FILES_STEP = 64                  # the header grows in steps of 64 words (the first: 4 + 60 counts = 512 bytes)
collectives_issued = 0            # (tests) all-gathers issued by this module


@dataclass
class _Agreed:
    """what all ranks of a process group agree on before an exchange (identical on every rank: only ever updated
    from gathered headers)"""
    group: object                 # strong reference: keeps id(group) from being reused while the entry lives
    record_bytes: int = 0
    chunks: int = 0               # record capacity of a rank's buffer
    files: int = _FILES_STEP - _HEADER_FIXED


_agreed: dict = {}                # id(process group) -> _Agreed


def _agreed_capacity(chunks: int) -> int:
    """capacity every rank derives from the same number: the next multiple of 512 chunks above `chunks` + 12 %
    (files of a stream are about the same length: one 1-hour file is 3 591 chunks -> 4 096)"""
    return -(-int(chunks * 1.125 + 1) // 512) * 512


def _agreed_files(files: int) -> int:
    return -(-(files + _HEADER_FIXED) // _FILES_STEP) * _FILES_STEP - _HEADER_FIXED


def _state_of(shard: Shard) -> _Agreed:
    """the agreement of `shard.group` (None = the default group).  Entries of destroyed process groups are dropped:
    a group created later starts from scratch on every rank, whatever `id()` it gets."""
    group = shard.group if shard.group is not None else dist.group.WORLD
    try:
        alive = set(map(id, dist.distributed_c10d._world.pg_map))
        for k in [k for k in _agreed if k not in alive]:
            del _agreed[k]
    except Exception:   # (private registry moved: entries then live as long as the process, which is harmless)
        pass
    st = _agreed.get(id(group))
    if st is None or st.group is not group:
        st = _agreed[id(group)] = _Agreed(group)
    return st


def all_gather_files(records: Sequence[torch.Tensor], shard: Shard, device: torch.device,
                     record_bytes: Optional[int] = None) -> List[List[torch.Tensor]]:
    """Many files.  `records`: this rank's per-file (C_f, R) uint8 record tensors.  Returns, on every rank,
    result[r][j] = records of the j-th file of rank r.

    ONE all-gather per call in steady state: a rank's buffer is a header {flag, number of files, chunks, record
    bytes R, chunks of file 0, 1, ...} followed by its records, padded to a capacity (files and chunks per rank,
    R) that all ranks already agree on -- it only ever changes through numbers every rank has read from the
    gathered headers, so every rank computes the same buffer size without talking.  The very first call of a
    process group, and a call in which some rank's files outgrow the agreement (more chunks, more files, another
    record size: that rank sends its header with the flag set and no records; every rank sees it, moves to the
    same new agreement and repeats), take TWO.  Nothing is checked BEFORE the collective: a rank that cannot take
    part as agreed says so inside it, and errors (ranks announcing different record sizes) are raised by every
    rank together after it -- no rank is ever left waiting in an all-gather the others never enter.
    `record_bytes`: R, for a rank that has no file of its own (all ranks run the same models); a rank that knows
    neither simply contributes no file."""
    global collectives_issued
    R_local = int(records[0].shape[1]) if len(records) else int(record_bytes or 0)
    counts = [int(r.shape[0]) for r in records]
    need = sum(counts)
    st = _state_of(shard)
    wire = _wire_device(shard, device)
    while True:
        words = _HEADER_FIXED + st.files
        as_agreed = (need == 0 or R_local == st.record_bytes) and need <= st.chunks and len(counts) <= st.files \
            and (R_local in (0, st.record_bytes))
        header = torch.zeros(words, dtype=torch.int64)
        header[0] = 0 if as_agreed else 1
        header[1] = len(counts)
        header[2] = need
        header[3] = R_local
        if len(counts) <= st.files:
            header[_HEADER_FIXED:_HEADER_FIXED + len(counts)] = torch.tensor(counts, dtype=torch.int64)
        hbytes = 8 * words
        buf = torch.zeros(hbytes + st.chunks * st.record_bytes, dtype=torch.uint8, device=device)
        buf[:hbytes] = header.view(torch.uint8).to(device)
        if as_agreed and need:
            buf[hbytes:hbytes + need * R_local] = torch.cat([r.to(device) for r in records], dim=0).reshape(-1)
        send = buf if buf.device == wire else buf.to(wire)
        recv = torch.empty((shard.world_size, send.numel()), dtype=torch.uint8, device=wire)
        dist.all_gather_into_tensor(recv.view(-1), send, group=shard.group)
        collectives_issued += 1
        heads = recv[:, :hbytes].cpu().contiguous().view(torch.int64).reshape(shard.world_size, -1)
        sizes = sorted({int(v) for v in heads[:, 3].tolist() if int(v) > 0})
        if len(sizes) > 1:     # (every rank sees the same headers: every rank raises)
            raise ValueError(f"all_gather_files: ranks announce different record sizes {sizes} "
                             "(all ranks must run the same models)")
        if int(heads[:, 0].max().item()) == 0:
            break
        # the new agreement, from the headers alone
        if sizes and sizes[0] != st.record_bytes:
            st.record_bytes, st.chunks = sizes[0], 0
        most_chunks, most_files = int(heads[:, 2].max().item()), int(heads[:, 1].max().item())
        if most_chunks > st.chunks:
            st.chunks = _agreed_capacity(most_chunks)
        if most_files > st.files:
            st.files = _agreed_files(most_files)
    if recv.device != device:
        recv = recv.to(device)
    R = st.record_bytes
    out: List[List[torch.Tensor]] = []
    for r in range(shard.world_size):
        files, pos = [], hbytes
        for j in range(int(heads[r, 1].item())):
            c = int(heads[r, _HEADER_FIXED + j].item())
            files.append(recv[r, pos:pos + c * R].reshape(c, R))
            pos += c * R
        out.append(files)
    return out
