"""BASELINE.json configs[4] on ONE GPU: N one-hour files, ONE joint clustering over all of them
(`apply_batch(files, joint_clustering=True)`) -- measures what the serial dendrogram merge costs at the
size class of 8 hours of audio.  usage: python tools/joint_scale.py [num_files ...]"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pyannote_audio_amd as pa
import pyannote_audio_amd.distance
from bench import build_checkpoints, synth_hour

dev = torch.device("cuda:0")
workdir = tempfile.mkdtemp(prefix="pa_joint_")
build_checkpoints(workdir)
pipeline = pa.Pipeline.from_pretrained(workdir).to(dev)
waves = {}
for n in [int(a) for a in sys.argv[1:]] or [2, 4, 8]:
    for i in range(n):
        if i not in waves:
            waves[i] = synth_hour(1.0, seed=100 + i, device=dev)
    files = [{"waveform": waves[i], "sample_rate": 16000, "uri": f"h{i}"} for i in range(n)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = list(pipeline.apply_batch(files, joint_clustering=True))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t = pipeline.clustering.timings
    print(f"{n} files x 1 h: total {dt:.2f} s ({n / dt:.3f} audio-h/s); joint clustering on "
          f"{t.get('num_embeddings')} training embeddings: cluster {t.get('cluster', 0):.2f} s "
          f"(linkage {t.get('linkage', 0):.2f} s), assign {t.get('assign', 0):.2f} s; "
          f"speakers {len(outs[0][1].speaker_diarization.labels())}", flush=True)
    st = pa.distance.last_linkage_stats
    if st is not None:
        m = max(int(t.get("num_embeddings") or 1), 1)
        print(f"    merge kernel: heap-free status {st[8]} workgroups {st[13]} repairs/merge {st[9] / m:.2f} "
              f"cycles(pop, pass)/merge {[int(c / m) for c in st[10:12]]}; heap kernel merges {st[7]}", flush=True)
    if pa.distance.last_linkage_phases:   # PA_LINKAGE_TIMING=1
        print("    linkage call: " + ", ".join(f"{k} {1e3 * v:.0f} ms" for k, v in pa.distance.last_linkage_phases),
              flush=True)
