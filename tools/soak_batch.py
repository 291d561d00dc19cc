"""Soak check of the batch path (development aid): repeated `pipeline(files)` calls -- threads and device memory must not
grow from call to call (the run-ahead producer, the tail executor and the read-ahead worker all end with their call).
usage: python tools/soak_batch.py [calls]"""
import os, sys, tempfile, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import pyannote_audio_amd as pa
from conftest import write_pipeline_dir
from oracle.models import seeded_pyannet, seeded_wespeaker
from oracle.synthetic import synth_conversation
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 8
d = tempfile.mkdtemp()
write_pipeline_dir(d, seeded_pyannet(), seeded_wespeaker())
dev = torch.device("cuda:0")
pipeline = pa.Pipeline.from_pretrained(d).to(dev)
files = [{"waveform": synth_conversation(20.0 + 3 * i, seed=i)[0], "sample_rate": 16000, "uri": f"f{i}"} for i in range(5)]
base = None
for c in range(calls):
    outs = list(pipeline(files))
    if c % 3 == 2:                      # a consumer that walks away after two results
        it = iter(pipeline(files)); next(it); next(it); it.close()
    torch.cuda.synchronize()
    th, mem = threading.active_count(), torch.cuda.memory_allocated(dev)
    print(f"call {c}: {len(outs)} outputs, {th} threads, {mem / 1e6:.1f} MB allocated", flush=True)
    if c == 2:
        base = (th, mem)
    if c > 2:
        assert th <= base[0] + 1, (th, base)
        assert mem <= base[1] * 1.05 + 1e6, (mem, base)
print("soak ok")
