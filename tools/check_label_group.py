"""Functional check on the GPU box (one rank): the process-group combination of the multi-GPU joint path -- default
group RCCL (`init_process_group("nccl", device_id=...)` as bench.py does), a second GLOO group for the label
broadcasts (SpeakerDiarization._label_group), parallel.broadcast_object through it from a worker thread while the
main thread runs an RCCL all-gather (parallel.all_gather_files).  usage: python tools/check_label_group.py"""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29519")
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from pyannote_audio_amd import parallel
from pyannote_audio_amd.speaker_diarization import SpeakerDiarization

shard = parallel.shard_from_env()
group = SpeakerDiarization._label_group(SpeakerDiarization.__new__(SpeakerDiarization), shard)
print("default backend", dist.get_backend(), "| label group backend", dist.get_backend(group))
out = {}

def tail():
    obj = (np.arange(12, dtype=np.int64).reshape(4, 3), np.ones((2, 256)))
    out["labels"] = parallel.broadcast_object(obj, 0, shard, group, dev)

t = threading.Thread(target=tail)
t.start()
rec = [torch.arange(5 * 4839, dtype=torch.int64, device=dev).remainder(251).to(torch.uint8).reshape(5, 4839)]
got = parallel.all_gather_files(rec, shard, dev, record_bytes=4839)
t.join()
assert torch.equal(got[0][0], rec[0]) and out["labels"][0].shape == (4, 3)
dist.barrier()
dist.destroy_process_group()
print("label group + RCCL exchange from two threads: ok")
