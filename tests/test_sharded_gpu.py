"""SURVEY.md section 8e on hardware: ONE file sharded by chunk range over 2 ranks (one process per rank;
both ranks share the single GPU of the test box, the exchange goes through gloo -- on a multi-GPU node the
same code runs one rank per GPU over RCCL).  Every rank must end with exactly the unsharded result."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, pipeline_dir, seconds, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pyannote_audio_amd as pa
    from pyannote_audio_amd import parallel
    from oracle.synthetic import synth_conversation
    wav, _ = synth_conversation(seconds, seed=31)
    pipeline = pa.Pipeline.from_pretrained(pipeline_dir).to(torch.device("cuda:0"))
    parallel.set_shard(parallel.shard_from_env())
    out = pipeline({"waveform": wav, "sample_rate": 16000, "uri": "sharded"})
    turns = [(s.start, s.end, l) for s, _, l in out.speaker_diarization.itertracks(yield_label=True)]
    q.put((rank, turns, out.speaker_embeddings.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_file_equals_unsharded(pipeline_dir, gpu_device):
    import torch.multiprocessing as mp
    import pyannote_audio_amd as pa
    from oracle.synthetic import synth_conversation
    seconds = 47.0   # 38 full chunks + a ragged last one -> uneven shares
    wav, _ = synth_conversation(seconds, seed=31)
    ref = pa.Pipeline.from_pretrained(pipeline_dir).to(gpu_device)(
        {"waveform": wav, "sample_rate": 16000, "uri": "sharded"})
    want = [(s.start, s.end, l) for s, _, l in ref.speaker_diarization.itertracks(yield_label=True)]
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, pipeline_dir, seconds, q))
             for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert len(want) > 0
    for rank, turns, cent in res:
        assert turns == want, f"rank {rank}"
        assert np.array_equal(np.asarray(cent, dtype=ref.speaker_embeddings.dtype), ref.speaker_embeddings)
