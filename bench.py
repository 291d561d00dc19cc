#!/usr/bin/env python
"""bench.py -- audio-hours/s of the speaker-diarization-3.1 hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: under torch.distributed.run, or plain -- the script then
                                                          starts its N ranks itself, `self_launch`)

One "step" = one pass of the whole pipeline (`Pipeline.__call__` -> sliding-window PyanNet
segmentation -> speaker counting -> WeSpeaker ResNet34 embeddings -> agglomerative clustering ->
reconstruction -> Annotation) over one synthetic 1-hour 16 kHz mono recording PER GPU, the waveform
already resident in HBM when the timed region starts (BASELINE.json configs[3]).  N > 1 is configs[4] AS
WRITTEN: one 1-hour file per GPU and step, ONE RCCL all-gather of the per-chunk hard segmentations +
embeddings of all files, ONE joint clustering over all of them, reconstruction per file (weak scaling);
the rate of N independent per-file pipelines (no exchange, per-file clustering) is measured right after
it and reported under "per_file_clustering" -- it is never `value` at N > 1.

Checkpoints are synthetic (no network): seeded weights in the reference's
state-dict layout with an extreme-learning-machine read-out so that the segmentation actually tracks
the synthetic speakers.  They are produced by the CPU oracle's torch modules (oracle/synthetic.py),
saved in the reference checkpoint format and loaded by the product through `Pipeline.from_pretrained`;
the oracle takes no part in the measured path.  The same oracle models run the `cpu_baseline` leg.

Output: ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel,
HIP events on the launch stream via the library's built-in profiler) and `cpu_baseline`; since round 4 the default
N = 1 line also carries `roofline_others` (the other matrix-pipe kernels above 2 % of a step), `configs` (BASELINE.json
configs[1] and configs[2] -- `--config seg5s` / `emb3s` -- timed by the same run, a few steps each) and `ingest` (one
file that starts in HOST memory through `pipeline(file)`, one call at a time: what the reference's own speed metric,
__main__.py:684-744, measures and `value` leaves out).  None of these extra legs is inside the timed region of `value`.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# multi-process GPU work on this pool needs dmabuf IPC (RCCL otherwise fails with `hipIpcGetMemHandle: invalid
# argument`); the variable is exported on the boxes already -- kept here for an environment built by hand
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch
import torch.distributed as dist

PEAK_MFMA_F32_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_HBM_GBS = 8000.0          # HBM3E spec
# kernels whose roofline is the f32 MFMA rate; everything else is priced against HBM bandwidth
MFMA_KERNELS = {"k_conv3x3", "k_conv3x3_wino", "k_conv3x3_wino4", "k_gemm_tn", "k_lstm_rec", "k_sinc_fir_pool",
                "k_conv5_pool", "k_sinc_fir_span"}
# Winograd F(2x2,3x3) executes 16 multiplies per 2x2 output tile and (cin, cout) pair instead of 36; F(4x4,3x3) 36
# per 4x4 tile instead of 144
EXECUTED_FLOP_FRACTION = {"k_conv3x3_wino": 16.0 / 36.0, "k_conv3x3_wino4": 36.0 / 144.0}
# bare v_mfma_f32_16x16x4_f32 stream measured on MI355X with random operands over 0.7 s: 151.8 TFLOP/s at
# a 2.37 GHz shader clock (profiles/r2_clock_trace.txt; bursts of a few ms run at ~2.1 GHz while the
# clock ramps: profiles/r2_mfma_probe_box2.txt)
SUSTAINED_MFMA_F32_TFLOPS = 151.8


def build_checkpoints(workdir: str):
    """synthetic speaker-diarization-3.1 directory (config.yaml + 2 reference-format checkpoints)."""
    from oracle.synthetic import calibrated_pyannet, calibrated_wespeaker
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import write_pipeline_dir
    seg_o = calibrated_pyannet(calib_seconds=60.0)
    emb_o = calibrated_wespeaker(calib_seconds=24.0)
    write_pipeline_dir(workdir, seg_o, emb_o)
    return seg_o, emb_o


def synth_hour(hours: float, seed: int, device=None) -> torch.Tensor:
    """(1, n) float32 synthetic 3-speaker conversation (harmonic "voices" with distinct pitch and
    formant, random turn-taking with 25 % overlapped turns, light noise), synthesised on `device`.
    Same recipe as oracle.synthetic.synth_conversation, vectorised so that an hour takes seconds."""
    sr, S = 16000, 3
    n = int(hours * 3600 * sr)
    rng = np.random.default_rng(seed)
    dev = device or torch.device("cpu")
    t = torch.arange(n, dtype=torch.float64, device=dev) / sr
    d2 = torch.zeros((S, n + 1), dtype=torch.float64, device=dev)
    events = [[] for _ in range(S)]

    def add(s, a, b):
        a, b = max(0, a), min(n, b)
        if b - a < 200:
            return
        k = min(400, (b - a) // 2)
        events[s].append((a, b, k))

    pos, dur = 0.0, n / sr
    while pos < dur:
        pos += rng.uniform(0.1, 1.2)
        s = int(rng.integers(S))
        d = rng.uniform(0.8, 4.0)
        a, b = int(pos * sr), int((pos + d) * sr)
        if a >= n:
            break
        add(s, a, b)
        if rng.uniform() < 0.25:
            s2 = (s + 1 + int(rng.integers(S - 1))) % S
            a2 = a + (b - a) // 2
            b2 = b + int(rng.uniform(0.3, 1.5) * sr)
            add(s2, a2, b2)
            pos = min(b2, n) / sr
        else:
            pos = min(b, n) / sr
    wav = torch.zeros(n, dtype=torch.float64, device=dev)
    for s in range(S):
        ev = np.array(events[s], dtype=np.int64).reshape(-1, 3)
        a, b, k = (torch.from_numpy(ev[:, i]).to(dev) for i in range(3))
        inv = 1.0 / k.to(torch.float64)
        # trapezoid envelopes as the double prefix sum of 4 impulses per turn
        d2[s].index_add_(0, a, inv)
        d2[s].index_add_(0, a + k, -inv)
        d2[s].index_add_(0, b - k, -inv)
        d2[s].index_add_(0, b, inv)
        env = torch.cumsum(torch.cumsum(d2[s], 0), 0)[:n].clamp_(0.0, 1.0)
        f0 = 95 + 55 * s + rng.uniform(-5, 5)
        vib = 1 + 0.01 * torch.sin(2 * np.pi * 0.7 * t)
        sig = torch.zeros_like(t)
        for h in range(1, 14):
            amp = (1.0 / h) * np.exp(-((f0 * h - (450 + 500 * s)) / 1000.0) ** 2)
            sig += amp * torch.sin(2 * np.pi * f0 * h * t * vib + rng.uniform(0, 6.28))
        sig *= 0.6 + 0.4 * torch.sin(2 * np.pi * (3 + s) * t)
        sig /= sig.abs().max()
        wav += 0.3 * sig * env
    g = torch.Generator(device=dev).manual_seed(seed)
    wav += 0.003 * torch.randn(n, dtype=torch.float64, device=dev, generator=g)
    return wav.clamp_(-1, 1).to(torch.float32)[None]


class StageTimer:
    """wall-clock per pipeline stage from the reference's hook protocol (TimingHook semantics)."""

    def __init__(self):
        self.t = {}
        self._last = None

    def start(self):
        self._last = time.perf_counter()
        self.t = {}

    def __call__(self, step, artifact, file=None, total=None, completed=None):
        if total is not None:   # progress callbacks
            return
        torch.cuda.synchronize()
        now = time.perf_counter()
        self.t[step] = self.t.get(step, 0.0) + now - self._last
        self._last = now


def cpu_baseline(seg_o, emb_o, seconds: float, hour_artifacts=None, hours: float = 1.0):
    """the CPU oracle (reference algorithm, reference batching: 32/32, 3 backbone passes per chunk)
    on a bounded sample, on the host cores of this box.  The neural stages scale linearly with the
    audio, the reference's host clustering (SciPy pdist + linkage, O(N^2)) does not: it is timed
    separately at the FULL size of the benchmarked file, on the embeddings the GPU run produced, and
    `value` combines the linearly scaled stages with that measured clustering time."""
    from oracle import pipeline as op
    from oracle.synthetic import synth_conversation
    wav, _ = synth_conversation(seconds, seed=77)
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    t0 = time.perf_counter()
    out = op.diarize(seg_o, emb_o, wav, exclude_overlap=True)
    dt = time.perf_counter() - t0
    res = {"value": (seconds / 3600.0) / dt, "unit": "audio-hours/s", "cores": torch.get_num_threads(),
           "kind": "port", "sample": f"{seconds:.0f} s synthetic conversation, full pipeline, "
           f"{dt:.1f} s wall", "stages_s": {k: round(v, 3) for k, v in out.timings.items()}}
    if hour_artifacts is not None:
        emb, seg = hour_artifacts
        t0 = time.perf_counter()
        op.clustering(np.array(emb), seg, min_clusters=1, max_clusters=np.inf, method="centroid",
                      threshold=0.7045654963945799, min_cluster_size=12)
        tc = time.perf_counter() - t0
        linear = (dt - out.timings.get("clustering", 0.0)) * (hours * 3600.0 / seconds)
        res.update({"value": hours / (linear + tc), "clustering_full_size_s": round(tc, 2),
                    "linear_stages_scaled_s": round(linear, 1),
                    "sample": res["sample"] + f"; neural / frame stages scaled x{hours * 3600.0 / seconds:.0f}, "
                    f"host clustering (SciPy) timed at full size ({emb.shape[0] * emb.shape[1]} embeddings): "
                    f"{tc:.1f} s"})
    return res


def load_traffic(kernel: str, config: str = "pipeline"):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (separate FETCH_SIZE /
    WRITE_SIZE runs, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950; summarised by
    tools/pmc_traffic.py).  PMC counters cannot be read from inside this process, so the figure is a
    STATIC one and says which file / commit it comes from.  A stage configuration (`--config seg5s|emb3s`) launches
    the kernels on other shapes than the pipeline does: it only ever reads a capture of ITS OWN command
    (profiles/r5_traffic_<config>.json, tools/capture_config_traffic.sh) -- or reports null."""
    names = ("r6_traffic.json", "r5_traffic.json", "r4_traffic.json", "r3_traffic.json", "r3a_traffic.json",
             "r2_traffic.json", "r1_traffic.json") if config == "pipeline" else \
        (f"r6_traffic_{config}.json", f"r5_traffic_{config}.json")
    for name in names:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as fp:
                d = json.load(fp)
        except OSError:
            continue
        if kernel in d:
            src = f"profiles/{name}" + (f"@{d['_commit']}" if "_commit" in d else "")
            return d[kernel].get("hbm_bytes_per_launch"), src
    return None, None


def roofline_entry(name: str, r: dict, config: str = "pipeline") -> dict:
    """`roofline` object for one kernel from the library profiler's record (HIP events on the launch
    stream): `achieved` is what the bounding unit really executed -- for the Winograd kernel the
    matrix pipe issues 16/36 of the direct convolution's multiplies, so `achieved` counts those and
    `frac` <= 1; the reference operation's (direct-convolution) rate is reported separately."""
    launches = max(r["launches"], 1)
    avg_ms = r["ms"] / launches
    traffic, src = load_traffic(name, config)
    if name in MFMA_KERNELS:
        algorithmic = r["flops"] / r["ms"] / 1e9
        executed = algorithmic * EXECUTED_FLOP_FRACTION.get(name, 1.0)
        roof = {"kernel": name, "bound": "mfma", "achieved": round(executed, 2),
                "peak": PEAK_MFMA_F32_TFLOPS, "unit": "TFLOP/s",
                "frac": round(executed / PEAK_MFMA_F32_TFLOPS, 4),
                "algorithmic": {"tflops": round(algorithmic, 2),
                                "gflop_per_launch": round(r["flops"] / launches / 1e9, 3),
                                "note": "flops of the reference operation (for the Winograd kernel: the direct 3x3 convolution)"},
                "executed_gflop_per_launch": round(
                    r["flops"] * EXECUTED_FLOP_FRACTION.get(name, 1.0) / launches / 1e9, 3),
                "sustained_mfma_tflops": SUSTAINED_MFMA_F32_TFLOPS,
                "frac_of_sustained": round(executed / SUSTAINED_MFMA_F32_TFLOPS, 4)}
        if name == "k_conv3x3_wino4":
            # F(4x4) issues 36/144 of the direct convolution's multiplies, F(2x2) 16/36: the same wall-clock rate
            # expressed in the currency of the F(2x2) kernel it replaced (rounds 1-3 quoted that kernel's `frac`)
            roof["f2x2_equivalent_frac"] = round(algorithmic * EXECUTED_FLOP_FRACTION["k_conv3x3_wino"]
                                                 / PEAK_MFMA_F32_TFLOPS, 4)
    else:
        ach = r["bytes"] / r["ms"] / 1e6
        roof = {"kernel": name, "bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS,
                "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBS, 4)}
    roof.update({"traffic": traffic, "traffic_source": src,
                 "algorithmic_bytes_per_launch": round(r["bytes"] / launches),
                 "launches": r["launches"], "avg_launch_ms": round(avg_ms, 4)})
    return roof


def bench_stage(args, pipeline, device, rank, config=None, steps=None, warmup=None):
    """BASELINE.json configs[1] / configs[2]: one model stage alone, inputs resident in HBM.
    A step = one pass over the whole synthetic input.  Returns the JSON line as a dict."""
    import pyannote_audio_amd.ffi as ffi
    config = config or args.config
    steps = steps or args.steps
    warmup = args.warmup if warmup is None else warmup
    g = torch.Generator(device=device).manual_seed(1000 + rank)
    if config == "seg5s":
        wav = synth_hour(args.hours, seed=rank, device=device).view(-1)
        window, stride = 80000, 8000
        num = (wav.numel() - window) // stride + 1
        engine = pipeline._segmentation.model.engine
        per_unit_gflop = 1.313      # PyanNet (L = 4) on one 5 s chunk
        workload = (f"PyanNet sliding window 5 s / 0.5 s over {args.hours:g} h of 16 kHz audio "
                    f"({num} chunks, one launch group)")
        metric, unit, units_per_step = "segmentation chunks/s (5 s chunks)", "chunks/s", num

        def step():
            engine.forward_strided(wav, stride, num, window, want_logp=True, want_multilabel=True)
    else:
        num, window = 10000, 48000
        wav = (0.1 * torch.randn(num * window, device=device, generator=g)).clamp_(-1, 1)
        Fm = 173                     # segmentation frames of a 3 s chunk
        masks = (torch.rand((num, 1, Fm), device=device, generator=g) < 0.7).float()
        engine = pipeline._embedding.model_.engine
        per_unit_gflop = 13.575      # ResNet34 on one 3 s item
        workload = "WeSpeaker ResNet34 embeddings of 10 000 x 3 s segments, Bernoulli(0.7) masks"
        metric, unit, units_per_step = "embedding segments/s (3 s segments)", "segments/s", num

        def step():
            engine.forward_strided(wav, window, num, window, masks)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ffi.prof_enable(True)
    step()
    torch.cuda.synchronize()
    prof = ffi.prof_report()
    ffi.prof_enable(False)
    dom = max(prof, key=lambda k: prof[k]["ms"])
    rate = units_per_step * steps / elapsed
    stage_tflops = rate * per_unit_gflop / 1e3
    line = {"metric": metric, "value": round(rate, 1), "unit": unit, "n_gpus": 1, "steps": steps,
            "warmup": warmup, "ms_per_step": round(1e3 * elapsed / steps, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic", "config": {"workload": workload},
            "audio_hours_per_s": round(rate * (window / 16000.0) / 3600.0 *
                                       (0.1 if config == "seg5s" else 1.0), 4),
            "stage_algorithmic": {"tflops": round(stage_tflops, 2),
                                  "frac_of_f32_mfma_peak": round(stage_tflops / PEAK_MFMA_F32_TFLOPS, 4),
                                  "gflop_per_unit": per_unit_gflop},
            "roofline": roofline_entry(dom, prof[dom], config),
            "kernels": {k: {"launches": r["launches"], "ms": round(r["ms"], 3),
                            "tflops": round(r["flops"] / r["ms"] / 1e9, 2) if r["ms"] > 0 else None,
                            "gbs": round(r["bytes"] / r["ms"] / 1e6, 1) if r["ms"] > 0 else None}
                        for k, r in prof.items()},
            "cpu_baseline": None}
    return line


def bench_ingest(pipeline, wav: torch.Tensor, device, hours: float, reps: int = 2):
    """What the headline leaves out (the reference's own speed metric, __main__.py:684-744, wall-clocks the file
    loop INCLUDING loading): ONE file that starts on the HOST (pageable memory, as a decoder leaves it) through
    `pipeline(file)`, one call at a time -- host-to-device copy of the waveform, front end, clustering and back end
    all exposed -- beside the same call on the HBM-resident waveform."""
    wav_host = wav.cpu()
    t = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        w = wav_host.to(device)
        torch.cuda.synchronize()
        t.append(time.perf_counter() - t0)
        del w

    def one(waveform, tag):
        # one untimed call first: apply() runs on the default stream, apply_batch's tail on a side stream, and torch's
        # caching allocator keeps a pool per stream -- the first plain calls behind a batch pay ~25 ms of hipMalloc
        # for the 206-MB condensed distance matrix (tools/single_file_phases.py, profiles/r6_single_file_phases.txt)
        pipeline({"waveform": waveform, "sample_rate": 16000, "uri": f"ingest_{tag}_warm"})
        best = None
        for i in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pipeline({"waveform": waveform, "sample_rate": 16000, "uri": f"ingest_{tag}{i}"})
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        return best

    host_s, resident_s = one(wav_host, "h"), one(wav, "r")
    return {"waveform_bytes": wav_host.numel() * 4, "h2d_ms": round(1e3 * min(t), 2),
            "single_file_from_host": {"value": round(hours / host_s, 5), "unit": "audio-hours/s",
                                      "ms_per_file": round(1e3 * host_s, 1)},
            "single_file_resident": {"value": round(hours / resident_s, 5), "unit": "audio-hours/s",
                                     "ms_per_file": round(1e3 * resident_s, 1)},
            "note": "pipeline(file) one call at a time (nothing overlaps), best of %d; `value` of the line is the "
                    "pipelined stream of HBM-resident files" % reps}


def bench_sparse_speech(pipeline, wav: torch.Tensor, hours: float, period_s: int = 120, pause_s: int = 60):
    """A recording that is half pauses (every `period_s` seconds the last `pause_s` are digital silence): the chunks in
    which no speaker is active skip the embedding backbone (SpeakerDiarization._embed_speech_chunks; their embeddings
    cannot change any output) -- `pipeline(file)` one call at a time with the skip (the default) and without it (what
    the reference computes).  Never `value`: the headline file has no pauses and embeds every chunk."""
    sparse = wav.clone()
    n = sparse.shape[1]
    for start in range(period_s - pause_s, n // 16000, period_s):
        sparse[:, start * 16000:(start + pause_s) * 16000] = 0.0
    file = {"waveform": sparse, "sample_rate": 16000, "uri": "sparse"}
    out = {"pattern": f"{period_s - pause_s} s of conversation, {pause_s} s of silence, repeated",
           "note": "the seeded segmentation read-out was calibrated on conversation only and reports speakers in digital "
                   "silence; the chunks that lie wholly inside a pause are forced to 'nobody' here, which is what a "
                   "trained model outputs by itself"}
    keep = pipeline.skip_inactive_chunks
    inference = pipeline._segmentation
    plain_slide = inference.slide
    window = round(inference.duration)
    inside = [c for c in range(max(0, n // 16000 - window + 1))
              if c % period_s >= period_s - pause_s and (c + window - 1) // period_s == c // period_s]

    def slide(*args, **kwargs):
        result = plain_slide(*args, **kwargs)
        idx = [c for c in inside if c < result.data.shape[0]]
        result.data[idx] = 0
        inference.last_device_output[torch.as_tensor(idx, device=inference.last_device_output.device)] = 0
        return result
    inference.slide = slide
    try:
        for tag, skip in (("with_skip", True), ("without_skip", False)):
            pipeline.skip_inactive_chunks = skip
            pipeline(file)
            best = None
            for _ in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                pipeline(file)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            total, done = pipeline.last_embedded_chunks
            out[tag] = {"ms_per_file": round(1e3 * best, 1), "audio_hours_per_s": round(hours / best, 4),
                        "chunks": total, "chunks_through_backbone": done}
    finally:
        pipeline.skip_inactive_chunks = keep
        del inference.slide          # (the instance attribute: the class's method is back)
    return out


def bench_reference_metric(pipeline, wav: torch.Tensor, hours: float, num_files: int = 4):
    """The reference's OWN speed metric (`pyannote-audio benchmark`, src/pyannote/audio/__main__.py:684-744): wall
    clock around the loop `for file, prediction in pipeline(files)` over audio files ON DISK -- decoding included --
    that writes every prediction as RTTM (`write_rttm`) and keeps its `serialize()` for the JSON dump, reported
    as `seconds_per_hour` / `times_faster_than_realtime` beside the README's 31 s per hour (community-1 on one
    H100, README.md:107).  `num_files` 16-bit PCM WAV files of `hours` each are written first (untimed); the JSON
    dump happens after `tac`, as in the reference.  Never `value`."""
    import shutil
    import tempfile
    from scipy.io import wavfile
    root = tempfile.mkdtemp(prefix="pa_benchmark_")
    try:
        pcm = (wav[0].clamp(-1, 1) * 32767.0).round().to(torch.int16).cpu().numpy()
        files = []
        for i in range(num_files + 1):
            path = os.path.join(root, f"file{i}.wav")
            wavfile.write(path, 16000, pcm if i % 2 == 0 else pcm[::-1].copy())
            files.append({"audio": path, "uri": f"file{i}"})
        disk_bytes = os.path.getsize(files[0]["audio"])

        def loop(batch, tag):
            rttm_file = os.path.join(root, f"{tag}.rttm")
            serialized = {}
            tic = time.time()
            for file, prediction in pipeline(batch):
                serialized[file["uri"]] = prediction.serialize()
                with open(rttm_file, "a") as rttm:
                    prediction.speaker_diarization.write_rttm(rttm)
            tac = time.time()
            with open(os.path.join(root, f"{tag}.json"), "w") as f:
                json.dump(serialized, f, indent=2)
            return tac - tic, os.path.getsize(rttm_file)

        loop(files[:1], "warmup")
        processing, rttm_bytes = loop(files[1:], "benchmark")
        playing = hours * 3600.0 * num_files
        return {"files": num_files, "hours_per_file": hours, "wav_bytes_per_file": disk_bytes,
                "rttm_bytes": rttm_bytes,
                "total_processing_time": round(processing, 3),
                "seconds_per_hour": round(processing / (playing / 3600.0), 3),
                "times_faster_than_realtime": round(playing / processing, 1),
                "audio_hours_per_s": round(playing / 3600.0 / processing, 5),
                "published_reference": {"seconds_per_hour": 31.0, "what": "community-1 (same segmentation and "
                                        "embedding models, VBx clustering) on one H100, AMI-IHM, README.md:107",
                                        "comparable": "other hardware, other clustering, real speech: context only"},
                "note": "WAV on disk -> pipeline(files) -> RTTM + serialize(), wall clock around the file loop as in "
                        "src/pyannote/audio/__main__.py:684-744 (decoding, host-to-device copies and result "
                        "writing inside the clock)"}
    finally:
        shutil.rmtree(root, ignore_errors=True)


def self_launch(n_gpus: int, argv=None) -> int:
    """`python bench.py --gpus N` without a launcher: re-executes this command under
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` on a free local port (one process per GPU, RCCL
    over xGMI; rank 0 prints the JSON line on the inherited stdout) and returns its exit status -- non-zero as soon
    as any rank dies (torch.distributed.run tears the others down)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)]
    cmd += list(sys.argv[1:] if argv is None else argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n_gpus)))
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10,
                    help="timed files per GPU (the stream is pipelined: the last file's back end is exposed once)")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--hours", type=float, default=1.0, help="audio hours per GPU and step")
    ap.add_argument("--cpu-seconds", type=float, default=60.0, help="audio seconds of the CPU baseline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--config", choices=["pipeline", "seg5s", "emb3s"], default="pipeline",
                    help="pipeline = BASELINE.json configs[3] (the headline metric; configs[4] at N > 1); "
                         "seg5s = configs[1] (PyanNet 5 s / 0.5 s over 1 h); emb3s = configs[2] (ResNet34 on "
                         "10 000 x 3 s segments)")
    ap.add_argument("--joint", action="store_true",
                    help="(default at N > 1) ONE joint clustering over the files of all ranks = configs[4]; "
                         "at N = 1 (under torch.distributed.run) the same code path with one rank")
    ap.add_argument("--per-file", action="store_true",
                    help="N > 1: headline the per-file-clustering rate (N independent pipelines) instead")
    ap.add_argument("--sequential", action="store_true",
                    help="one pipeline(file) call per step instead of the pipelined apply_batch")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the (untimed for `value`) extra legs of the default line: BASELINE.json configs[1] / "
                         "configs[2] (`configs`) and the host-resident single file (`ingest`)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N` (no launcher): start the N ranks ourselves, one process per GPU
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ   # under torch.distributed.run
    # PA_BENCH_SHARED_GPU=1 (functional check of the N > 1 code path on a ONE-GPU box, never a measurement): all ranks
    # use cuda:0 and the process group is gloo -- RCCL refuses two ranks on one device
    shared_gpu = os.environ.get("PA_BENCH_SHARED_GPU") == "1"
    if shared_gpu:
        local_rank = 0
    if world > 1 or launched:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    exchange = dist.is_initialized()
    joint = exchange and (args.joint or (world > 1 and not args.per_file))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    import pyannote_audio_amd as pa
    import pyannote_audio_amd.ffi as ffi
    ffi.require_gpu()

    # host threads for the (untimed) checkpoint synthesis: do not oversubscribe the node at N > 1
    torch.set_num_threads(max(1, min(16, (os.cpu_count() or 8) // max(world, 1))))
    workdir = tempfile.mkdtemp(prefix=f"pa_bench_r{rank}_")
    seg_o, emb_o = build_checkpoints(workdir)
    pipeline = pa.Pipeline.from_pretrained(workdir)
    pipeline.to(device)

    if args.config != "pipeline":
        line = bench_stage(args, pipeline, device, rank)
        if rank == 0:
            print(json.dumps(line), flush=True)
        return

    wav = synth_hour(args.hours, seed=rank, device=device)      # resident in HBM before timing
    file = {"waveform": wav, "sample_rate": 16000, "uri": f"synthetic_{rank}"}
    timer = StageTimer()

    def run(num_files: int, tag: str, joint_mode: bool):
        """`num_files` steps = `num_files` one-hour files through the pipeline.  Per-file clustering:
        apply_batch, which overlaps clustering + reconstruction of file i with the front end of file i+1.
        Joint: one joint job per step = front end of this rank's file, all-gather of every rank's records
        over RCCL, one clustering of all of them, back end of this rank's file; jobs pipelined through
        `apply_joint_batches`."""
        files = [dict(file, uri=f"synthetic_{rank}_{tag}{i}") for i in range(num_files)]
        last = None
        if joint_mode:
            # one joint job per step; consecutive jobs are pipelined (front ends + exchange of job i+1 beside
            # the clustering / back end of job i), as apply_batch pipelines files
            for outs in pipeline.apply_joint_batches([[f] for f in files]):
                last = outs[-1][1]
        elif args.sequential:
            for f in files:
                last = pipeline(f)
        else:
            for _, last in pipeline(files):
                pass
        return last

    def barrier():
        if exchange:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(joint_mode: bool, tag: str):
        run(args.warmup, "w" + tag, joint_mode)
        barrier()
        t0 = time.perf_counter()
        result = run(args.steps, "s" + tag, joint_mode)
        barrier()
        dt = time.perf_counter() - t0
        if exchange:
            t = torch.tensor([dt], dtype=torch.float64, device="cpu" if shared_gpu else device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return result, dt

    out, elapsed = timed(joint, "")
    joint_info = None
    if joint:
        joint_info = {"points": int(getattr(pipeline.clustering, "timings", {}).get("num_embeddings", 0)),
                      "clustering_s": {k: (round(v, 4) if isinstance(v, float) else v)
                                       for k, v in pipeline.clustering.timings.items()}}
    other = None
    if world > 1:            # the other N > 1 mode, measured in the same run (second key, never `value`)
        _, other_elapsed = timed(not joint, "o")
        other = {"value": round(args.hours * world * args.steps / other_elapsed, 5),
                 "ms_per_step": round(1e3 * other_elapsed / args.steps, 2)}

    # ---- stage split of ONE sequential (un-pipelined, untimed) pass: host clock + device sync per stage
    artifacts = {}

    def collecting(step_name, artifact, file=None, total=None, completed=None):
        timer(step_name, artifact, file=file, total=total, completed=completed)
        if artifact is not None and total is None and step_name in ("segmentation", "embeddings"):
            artifacts[step_name] = artifact

    timer.start()
    pipeline(file, hook=collecting)
    stage_sum = dict(timer.t)

    def step():
        return pipeline(file)

    # ---- per-kernel timing of one extra (untimed) step: HIP events on the launch stream
    ffi.prof_enable(True)
    step()
    torch.cuda.synchronize()
    prof = ffi.prof_report()
    ffi.prof_enable(False)

    if rank == 0 and os.environ.get("PA_BENCH_DUMP_EMB"):
        hooked = {}
        pipeline(file, hook=lambda step, art, **kw: hooked.__setitem__(step, art) if kw.get("total") is None else None)
        emb, seg = hooked["embeddings"], hooked["segmentation"]
        tr, _, _ = pipeline.clustering.filter_embeddings(emb, seg)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        np.save(os.path.join(ROOT, "gpurun_out", "bench_train_emb.npy"), tr.astype(np.float32))
    if rank == 0:
        kernels = {}
        for name, r in prof.items():
            ms = r["ms"]
            kernels[name] = {"launches": r["launches"], "ms": round(ms, 3),
                             "tflops": round(r["flops"] / ms / 1e9, 2) if ms > 0 else None,
                             "gbs": round(r["bytes"] / ms / 1e6, 1) if ms > 0 else None}
        dom = max(prof, key=lambda k: prof[k]["ms"])
        roof = roofline_entry(dom, prof[dom])
        # the other matrix-pipe kernels that take more than 2 % of the step, same accounting (never `roofline`)
        step_ms = sum(r["ms"] for r in prof.values())
        others = [roofline_entry(n, r) for n, r in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])
                  if n != dom and n in MFMA_KERNELS and r["ms"] > 0.02 * step_ms]
        total_hours = args.hours * world * args.steps
        line = {
            "metric": "audio-hours/sec (real-time factor) for speaker-diarization-3.1 pipeline",
            "value": round(total_hours / elapsed, 5), "unit": "audio-hours/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 2), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "full speaker-diarization-3.1 pipeline (PyanNet 10 s / 1 s sliding "
                                   "window, WeSpeaker ResNet34 embeddings, centroid AHC) on "
                                   f"{args.hours:g} h of 16 kHz mono audio per GPU",
                       "chunks_per_file": int((wav.shape[1] - 160000) // 16000 + 1),
                       # (chunks of the last file, chunks of it that went through the embedding backbone: the skip
                       #  of chunks without an active speaker never fires on the headline file unless these differ)
                       "chunks_embedded": list(getattr(pipeline, "last_embedded_chunks", (0, 0))),
                       "files": world,
                       "parallelism": f"file-per-gpu x{world}" + (", joint clustering" if joint else "")},
            "real_time_factor": round(total_hours * 3600.0 / elapsed, 1),
            "mode": ("configs[4]: one file per GPU and step, RCCL all-gather of the records, ONE joint "
                     "clustering over all files per step -- computed by the step's owner rank (step % N), labels "
                     "broadcast (apply_joint_batches / pipelining.pipelined_owned)") if joint else
                    ("sequential pipeline(file) calls" if args.sequential else
                     "pipeline([files]) = apply_batch: clustering/back end of file i overlap the front end "
                     "of file i+1" + (" (N independent per-file pipelines, no exchange)" if world > 1 else "")),
            "sequential_stages_ms": {k: round(1e3 * v, 1) for k, v in stage_sum.items()},
            "speakers": len(out.speaker_diarization.labels()),
            "apply_marks_s": {k: round(v, 4) for k, v in getattr(pipeline, "timings", {}).items()},
            "clustering_s": {k: (round(v, 4) if isinstance(v, float) else v)
                             for k, v in pipeline.clustering.timings.items()},
            "roofline": roof,
            "roofline_others": others,
            "kernels": kernels,
        }
        # the numerical guard's decisions (weights.EmbeddingPack): a convolution demoted from F(4x4) to a slower
        # kernel changes the numbers above -- it has to be visible next to them
        guard = pipeline._embedding.model_.engine.pack.winograd_guard
        line["winograd_guard"] = {"convolutions_measured": len(guard),
                                  "demoted": [f"layer{g['layer']}.{g['block']}.conv{g['conv']} -> {g['path']}"
                                              for g in guard if g["path"] != ("f4" if g["f4"] is not None else "f2")]}
        if shared_gpu:
            line["functional_check_only"] = f"{world} ranks share cuda:0 over gloo (PA_BENCH_SHARED_GPU=1): not a measurement"
        if joint_info is not None:
            line["joint_clustering"] = joint_info
        if other is not None:
            line["joint_clustering_rate" if not joint else "per_file_clustering"] = other
        if getattr(pipeline, "batch_timeline", None) and not joint:   # host-clock stage boundaries of the timed files
            line["batch_timeline_s"] = [{k: round(v, 4) for k, v in f.items()} for f in pipeline.batch_timeline]
        if world == 1 and not args.no_extras:
            # the other single-GPU configurations of BASELINE.json, timed by the same run (never `value`), and
            # what a file that starts on the host costs
            line["ingest"] = bench_ingest(pipeline, wav, device, args.hours)
            line["reference_metric"] = bench_reference_metric(pipeline, wav, args.hours)
            line["sparse_speech"] = bench_sparse_speech(pipeline, wav, args.hours)
            line["configs"] = {}
            for name, (st, wu) in (("seg5s", (4, 1)), ("emb3s", (2, 1))):
                d = bench_stage(args, pipeline, device, rank, config=name, steps=st, warmup=wu)
                line["configs"][name] = {k: d[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step",
                                                           "config", "audio_hours_per_s", "stage_algorithmic",
                                                           "roofline")}
        if not args.no_cpu_baseline and world == 1:   # rank 0 at N = 1 only
            hour = None
            if "embeddings" in artifacts and "segmentation" in artifacts:
                hour = (artifacts["embeddings"], artifacts["segmentation"].data)
            line["cpu_baseline"] = cpu_baseline(seg_o, emb_o, args.cpu_seconds, hour, args.hours)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if exchange:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
