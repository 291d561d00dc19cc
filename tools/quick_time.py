"""Rough stage timing on one GPU (development aid, not the bench)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import seeded_pyannet, seeded_wespeaker
from pyannote_audio_amd.weights import SegmentationPack, EmbeddingPack
from pyannote_audio_amd.segmentation import SegmentationEngine
from pyannote_audio_amd.embedding import EmbeddingEngine

dev = torch.device("cuda:0")
hours = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
n = int(hours * 3600 * 16000)
wav = (0.1 * torch.randn(n, device=dev)).clamp(-1, 1)
N, step = 160000, 16000
C = (n - N) // step + 1
seg = SegmentationEngine(SegmentationPack(seeded_pyannet().state_dict(), {"lstm": {"num_layers": 4}}, 7, 3, 2, dev))
emb = EmbeddingEngine(EmbeddingPack(seeded_wespeaker().state_dict(), dev), max_chunks=int(os.environ.get("EMB_BATCH", 64)))

def timeit(f, reps=2):
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps

ts = timeit(lambda: seg.forward_strided(wav, step, C, N))
print(f"segmentation: {C} chunks in {ts*1e3:.1f} ms -> {C*2.636e-3/ts:.1f} TFLOP/s ({2.636*C/1e3/ts/157.3*100:.1f}% of fp32 MFMA peak)")
seg.release_workspace()
Ce = int(os.environ.get("EMB_CHUNKS", 512))
masks = (torch.rand(Ce, 3, 589, device=dev) < 0.7).float()
te = timeit(lambda: emb.forward_strided(wav, step, Ce, N, masks), reps=1)
print(f"embedding: {Ce} chunks in {te*1e3:.1f} ms -> {Ce*45.228e-3/te:.1f} TFLOP/s ({45.228*Ce/1e3/te/157.3*100:.1f}% of peak); per audio-hour {te/Ce*3591:.2f} s")
