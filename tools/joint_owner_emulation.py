"""BASELINE.json configs[4] at N = 8, the load of ONE rank measured on ONE GPU (no 8-GPU node is reachable).

A rank of an 8-GPU job stream does, per job: the front end of its own one-hour file (main stream), the exchange
(~1 ms over xGMI, not emulated), and the back end of its file; the ONE joint clustering of the job's 8 files
(57 k training embeddings) is done
  * redundantly by every rank (round 4: `pipelined`), or
  * by the job's owner only, rank j % 8, on a third stream, labels broadcast (round 5: `pipelined_owned`).
This script runs rank 0's share of both schedules with the REAL kernels: real front ends of a one-hour file, the real
clustering of the gathered records of 8 one-hour files (front ends computed once, untimed), real back ends; the labels
of the jobs rank 0 does not own come from a cached solution (what the broadcast would deliver).
usage: python tools/joint_owner_emulation.py [jobs]"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pyannote_audio_amd as pa
from pyannote_audio_amd.audio import Audio
from pyannote_audio_amd.pipelining import pipelined, pipelined_owned
from bench import build_checkpoints, synth_hour

jobs = int(sys.argv[1]) if len(sys.argv) > 1 else 25
world = 8
dev = torch.device("cuda:0")
workdir = tempfile.mkdtemp(prefix="pa_joint_")
build_checkpoints(workdir)
pipeline = pa.Pipeline.from_pretrained(workdir).to(dev)
sd = pipeline   # SpeakerDiarization
bounds = sd._speaker_bounds(None, None, None, {})
files = [{"waveform": synth_hour(1.0, seed=100 + i, device=dev), "sample_rate": 16000, "uri": f"h{i}"} for i in range(world)]
job8 = sd._joint_gather([Audio.validate_file(f) for f in files], None, dev)     # records of all 8 files
print(f"gathered {sum(job8['sizes'])} chunks of {world} files; {len(job8['all_emb_dev'])} x {tuple(job8['all_emb_dev'].shape[1:])} embeddings")
cached = sd._joint_cluster(job8, bounds)          # (also warms the allocator: the first call pays ~1 s of hipMalloc)
torch.cuda.synchronize()
side, solving = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
own = dict(job8, fronts=job8["fronts"][:1], hooks=job8["hooks"][:1], mine=job8["mine"][:1])   # rank 0's own file

front_done = []

def front(item, release):
    fr = sd._front_end(Audio.validate_file(dict(files[0], uri=f"job{item}")), sd.setup_hook(files[0]), release)
    torch.cuda.current_stream(dev).synchronize()
    front_done.append(time.perf_counter())
    return fr

def solve(state):
    with torch.cuda.device(dev), torch.cuda.stream(solving):
        out = sd._joint_cluster(job8, bounds)
        solving.synchronize()
    return out

def back_end(state, solution):
    with torch.cuda.device(dev), torch.cuda.stream(side):
        out = list(sd._joint_back_ends(own, solution[0], solution[1], bounds))
        side.synchronize()
    return out

def timed(gen):
    """-> (seconds for the whole stream, cadence of the front ends from job `world` on: the steady state of a long
    stream -- the results of the owned schedule come back in bursts, up to world + 1 jobs are in flight)"""
    front_done.clear()
    t = time.perf_counter()
    for _ in gen:
        pass
    total = time.perf_counter() - t
    return total, (front_done[-1] - front_done[world]) / (len(front_done) - 1 - world)

# round 4: every rank clusters every job
redundant = timed(pipelined(range(jobs), front, lambda st, alone: back_end(st, solve(st)), 5.0))
# round 5: rank 0 clusters jobs 0, 8, 16, ...; the other jobs' labels arrive by broadcast
owned = timed(pipelined_owned(range(jobs), front, solve, lambda j, owner, st, sol: sol if sol is not None else cached,
                              back_end, rank=0, world=world))
for name, (total, steady) in (("every rank clusters every job", redundant), ("owner of job j = rank j % 8", owned)):
    print(f"{name}: {jobs} jobs in {total:.2f} s ({total / jobs:.3f} s per job incl. fill and drain); front-end cadence "
          f"from job {world} on {steady:.3f} s per job -> {world / steady:.2f} audio-h/s projected at N = {world} "
          f"(8 one-hour files per job)")
t = sd.clustering.timings
print("clustering of one job:", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in t.items()})
