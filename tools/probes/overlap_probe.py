"""Can the HBM-bound front of the embedding stage (fbank + stem of the next launch group) run UNDER the
segmentation stage, whose LSTM recurrence is latency-bound (k_lstm_rec: one 359-register wave per SIMD, 0.63 of
peak, ~22 ms of a 96 ms stage)?  Existing entry points only, nothing of the product changes:

    A  segmentation forward of one audio-hour alone              (main stream)
    B  pa_fbank + pa_resnet_stem of 1 798 chunks alone           (side stream)
    C  both at once                                              (two streams)

If C is close to max(A, B) -- and A's own duration inside C close to A -- splitting pa_emb_forward into a
"front" (fbank, stem) and a "rest" half and issuing the front on a second stream when the segmentation stage starts
is worth ~B ms per launch group (2 groups per audio-hour).  usage (GPU box): python tools/probes/overlap_probe.py"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
import pyannote_audio_amd as pa
import pyannote_audio_amd.ffi as ffi

dev = torch.device("cuda:0")
work = tempfile.mkdtemp(prefix="pa_ov_")
bench.build_checkpoints(work)
pipeline = pa.Pipeline.from_pretrained(work).to(dev)
lib = ffi.load()
wav = bench.synth_hour(1.0, seed=0, device=dev).reshape(-1).contiguous()
seg = pipeline._segmentation.model.engine          # SegmentationEngine (Inference.model -> Model.engine)
emb_pack = pipeline._embedding.model_.engine.pack   # EmbeddingPack of the WeSpeaker engine
w = emb_pack.struct

N, STEP = 160000, 16000
C = (wav.numel() - N) // STEP + 1
B = 1798
T = lib.pa_emb_num_fbank_frames(N)
fb = torch.empty((B, T, 80), device=dev)
stem = torch.empty((B, 80, T, 32), device=dev)
side = torch.cuda.Stream(device=dev)


def run_seg():
    seg.forward_strided(wav, STEP, C, N, want_logp=False, want_multilabel=True)


def run_front():
    with torch.cuda.stream(side):
        ffi.check(lib.pa_fbank(ffi.ptr(wav), wav.numel(), STEP, B, N, w.fb_window, w.fb_tw256, w.fb_tw512, w.fb_mel_w,
                               w.fb_mel_lo, w.fb_mel_hi, 80, ffi.ptr(fb), 1, ffi.stream()), "fbank")
        ffi.check(lib.pa_resnet_stem(ffi.ptr(fb), B, T, 80, w.stem_w, w.stem_shift, ffi.ptr(stem), ffi.stream()), "stem")


def timed(*fns, reps=5):
    for f in fns:
        f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t = time.perf_counter()
        for f in fns:
            f()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t)
    return 1e3 * best


a, b = timed(run_seg), timed(run_front)
c = timed(run_front, run_seg)        # the front is queued first, on its own stream
print(f"A segmentation alone          {a:7.2f} ms")
print(f"B fbank + stem ({B} chunks)  {b:7.2f} ms")
print(f"C both, two streams           {c:7.2f} ms   (A + B = {a + b:.2f}; hidden: {a + b - c:.2f} ms)")
