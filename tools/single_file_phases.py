"""Where the wall time of ONE `pipeline(file)` call goes beyond its kernels (the reference's usual call:
one file at a time, nothing overlaps).  Prints apply marks, clustering timings and -- with PA_LINKAGE_TIMING=1 /
PA_LINKAGE_EVENTS=1 -- the phases of `distance.linkage_centroid`.
usage (GPU box): python tools/single_file_phases.py [hours] [reps]"""
import cProfile
import io
import json
import os
import pstats
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
import pyannote_audio_amd as pa
from pyannote_audio_amd import distance

hours = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
device = torch.device("cuda", 0)
torch.cuda.set_device(0)
workdir = tempfile.mkdtemp(prefix="pa_phases_")
bench.build_checkpoints(workdir)
pipeline = pa.Pipeline.from_pretrained(workdir)
pipeline.to(device)
wav = bench.synth_hour(hours, seed=0, device=device)
file = {"waveform": wav, "sample_rate": 16000, "uri": "phases"}
pipeline(file)     # warm-up (workspaces, allocator cache)
for i in range(reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = pipeline(file)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    marks = {k: round(v, 4) for k, v in getattr(pipeline, "timings", {}).items()}
    print(json.dumps({"rep": i, "ms": round(1e3 * dt, 1), "marks": marks,
                      "clustering": {k: (round(v, 4) if isinstance(v, float) else v)
                                     for k, v in pipeline.clustering.timings.items()},
                      "linkage_phases": [(n, round(1e3 * s, 2)) for n, s in (distance.last_linkage_phases or [])]}),
          flush=True)
if os.environ.get("PA_PHASES_PROFILE") == "1":
    pr = cProfile.Profile()
    pr.enable()
    pipeline(file)
    torch.cuda.synchronize()
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
    print(s.getvalue())
