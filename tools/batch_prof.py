"""Per-kernel HIP-event times of the SAME files, one at a time and as a pipelined batch (apply_batch: the
clustering / back end of file i runs on a second stream beside the front end of file i + 1): shows which kernels
of the front end are slowed by the overlapped tail.  usage (GPU box): python tools/batch_prof.py [files]"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import pyannote_audio_amd as pa
import pyannote_audio_amd.ffi as ffi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
work = tempfile.mkdtemp(prefix="pa_bp_")
bench.build_checkpoints(work)
pipeline = pa.Pipeline.from_pretrained(work).to(dev)
wav = bench.synth_hour(1.0, seed=0, device=dev)
files = [{"waveform": wav, "sample_rate": 16000, "uri": f"f{i}"} for i in range(n)]
pipeline(dict(files[0], uri="warm"))
list(pipeline([dict(f, uri="w" + f["uri"]) for f in files[:2]]))


def measure(tag, fn):
    torch.cuda.synchronize()
    ffi.prof_enable(True)
    ffi.prof_report()
    t = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    rep = ffi.prof_report()
    ffi.prof_enable(False)
    print(f"## {tag}: {1e3 * dt / n:.1f} ms per file (HIP-event profiling on)")
    return rep


seq = measure("sequential", lambda: [pipeline(dict(f, uri="s" + f["uri"])) for f in files])
bat = measure("pipelined batch", lambda: list(pipeline([dict(f, uri="b" + f["uri"]) for f in files])))
print(f"{'kernel':26s} {'launches':>8s} {'seq ms/file':>12s} {'batch ms/file':>14s} {'delta':>8s}")
tot = [0.0, 0.0]
for k in sorted(seq, key=lambda k: -seq[k]["ms"]):
    a, b = seq[k]["ms"] / n, bat.get(k, {"ms": 0.0})["ms"] / n
    if k != "k_linkage_centroid":
        tot[0] += a
        tot[1] += b
    print(f"{k:26s} {seq[k]['launches'] // n:8d} {a:12.2f} {b:14.2f} {b - a:8.2f}")
print(f"{'sum without the merge':26s} {'':8s} {tot[0]:12.2f} {tot[1]:14.2f} {tot[1] - tot[0]:8.2f}")
if getattr(pipeline, "batch_timeline", None):
    for f in pipeline.batch_timeline:
        print({k: round(v, 4) for k, v in f.items()})
