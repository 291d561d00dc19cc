"""Speaker diarization pipeline (`pyannote/speaker-diarization-3.1` configuration), MI355X native.

Mirrors pipelines/speaker_diarization.py (class :127-279, get_segmentations :305-330, get_embeddings
:332-478, reconstruct :480-528, apply :530-784).  Same stages, same outputs; different mechanics:
  * the waveform is uploaded once; segmentation and embedding kernels gather their chunks in HBM;
  * the ResNet backbone runs once per chunk and is pooled for the 3 local speakers (the reference runs
    it once per (chunk, speaker) on identical audio, :417-425);
  * overlap-add / top-k / hysteresis steps are array operations instead of per-frame Python loops;
  * `shard=(rank, world_size)` splits the chunk range across GPUs; per-chunk results are exchanged
    with one RCCL all-gather (parallel.py)."""
from __future__ import annotations

import math
import textwrap
import time
import warnings
from dataclasses import dataclass
from typing import Any, Callable, Optional

import numpy as np
import torch

from . import frames as frame_ops
from . import parallel
from .agglomerative import Clustering
from .audio import Audio, AudioFile
from .core import Annotation, SlidingWindow, SlidingWindowFeature
from .diarization import set_num_speakers, to_annotation
from .inference import Inference
from .model import Model
from .pipeline import ParamDict, Pipeline, Uniform
from .speaker_verification import PipelineModel, PretrainedSpeakerEmbedding, get_model


@dataclass
class DiarizeOutput:
    """speaker_diarization.py:63-124"""
    speaker_diarization: Annotation
    exclusive_speaker_diarization: Annotation
    speaker_embeddings: Optional[np.ndarray] = None

    def serialize(self) -> dict:
        def turns(annotation):
            return [{"start": round(s.start, 3), "end": round(s.end, 3), "speaker": l}
                    for s, _, l in annotation.itertracks(yield_label=True)]
        return {"diarization": turns(self.speaker_diarization),
                "exclusive_diarization": turns(self.exclusive_speaker_diarization)}


class SpeakerDiarization(Pipeline):
    def __init__(self, legacy: bool = False, segmentation: PipelineModel = None,
                 segmentation_step: float = 0.1, embedding: PipelineModel = None,
                 embedding_exclude_overlap: bool = False, plda: Any = None,
                 clustering: str = "VBxClustering", embedding_batch_size: int = 1,
                 segmentation_batch_size: int = 1, der_variant: Optional[dict] = None, token=None,
                 cache_dir=None):
        super().__init__()
        if segmentation is None or embedding is None:
            raise ValueError("`segmentation` and `embedding` must be local checkpoints (or Model "
                             "instances): Hugging Face defaults cannot be downloaded in this build.")
        self.legacy = legacy
        self.segmentation_model = segmentation
        model: Model = get_model(segmentation, token=token, cache_dir=cache_dir)
        self.segmentation_step = segmentation_step
        self.embedding = embedding
        self.embedding_batch_size = embedding_batch_size
        self.embedding_exclude_overlap = embedding_exclude_overlap
        self.plda = plda  # only consumed by VBxClustering (next row)
        self.klustering = clustering
        self.der_variant = der_variant or {"collar": 0.0, "skip_overlap": False}

        segmentation_duration = model.specifications.duration
        self._segmentation = Inference(model, duration=segmentation_duration,
                                       step=self.segmentation_step * segmentation_duration,
                                       skip_aggregation=True, batch_size=segmentation_batch_size)
        if self._segmentation.model.specifications.powerset:
            self.segmentation = ParamDict(min_duration_off=Uniform(0.0, 1.0))
        else:
            self.segmentation = ParamDict(threshold=Uniform(0.1, 0.9),
                                          min_duration_off=Uniform(0.0, 1.0))
        if self.klustering == "OracleClustering":
            metric = "not_applicable"
        else:
            self._embedding = PretrainedSpeakerEmbedding(self.embedding, token=token,
                                                         cache_dir=cache_dir)
            self._audio = Audio(sample_rate=self._embedding.sample_rate, mono="downmix")
            metric = self._embedding.metric
        try:
            Klustering = Clustering[clustering]
        except KeyError:
            raise ValueError(
                f"clustering must be one of [{', '.join(list(Clustering.__members__))}]")
        self.clustering = Klustering.value(metric=metric)
        self._expects_num_speakers = self.clustering.expects_num_clusters

    @property
    def segmentation_batch_size(self) -> int:
        return self._segmentation.batch_size

    @segmentation_batch_size.setter
    def segmentation_batch_size(self, batch_size: int):
        self._segmentation.batch_size = batch_size

    def default_parameters(self):
        if self.klustering == "AgglomerativeClustering":
            # the instantiated parameters of pyannote/speaker-diarization-3.1
            return {"segmentation": {"min_duration_off": 0.0},
                    "clustering": {"method": "centroid", "min_cluster_size": 12,
                                   "threshold": 0.7045654963945799}}
        return {"segmentation": {"min_duration_off": 0.0},
                "clustering": {"threshold": 0.6, "Fa": 0.07, "Fb": 0.8}}

    def classes(self):
        speaker = 0
        while True:
            yield f"SPEAKER_{speaker:02d}"
            speaker += 1

    # -----------------------------------------------------------------------------------------
    def _load(self, file) -> torch.Tensor:
        """whole file as a (1, n) fp32 tensor at the model's rate (core/inference.py:403)."""
        waveform, sample_rate = self._audio(file)
        return waveform

    def get_segmentations(self, file, hook=None, waveform: Optional[torch.Tensor] = None,
                          chunk_range=None) -> SlidingWindowFeature:
        """speaker_diarization.py:305-330"""
        if hook is not None:
            import functools
            hook = functools.partial(hook, "segmentation", None)
        if waveform is None:
            waveform = self._load(file)
        return self._segmentation.slide(waveform, self._audio.sample_rate, hook=hook,
                                        chunk_range=chunk_range)

    def _device_segmentation(self, binary_segmentations: SlidingWindowFeature) -> torch.Tensor:
        """(C, F, S) uint8 device copy of the hard segmentations: the tensor the segmentation kernels
        just produced when it is still the same data, an upload otherwise (sharded / user-supplied)."""
        dev = self._segmentation.last_device_output
        if dev is None or tuple(dev.shape) != tuple(binary_segmentations.data.shape) \
                or self._segmentation.last_host_output is not binary_segmentations.data:
            dev = frame_ops.as_device_segmentation(binary_segmentations.data, self._segmentation.device)
        cache = getattr(self, "_stats_cache", None)
        if cache is None or cache[0] is not dev:
            self._stats_cache = (dev, None)
        return dev

    def _chunk_stats(self, dev_bin: torch.Tensor):
        cache = getattr(self, "_stats_cache", None)
        if cache is not None and cache[0] is dev_bin and cache[1] is not None:
            return cache[1]
        stats = frame_ops.chunk_stats(dev_bin)
        self._stats_cache = (dev_bin, stats)
        return stats

    def get_embeddings(self, file, binary_segmentations: SlidingWindowFeature,
                       exclude_overlap: bool = False, hook: Optional[Callable] = None,
                       waveform: Optional[torch.Tensor] = None, chunk_range=None):
        """speaker_diarization.py:332-478 -> (num_chunks, num_speakers, dimension) float32."""
        device = self._embedding.device
        duration = binary_segmentations.sliding_window.duration
        num_chunks, num_frames, num_speakers = binary_segmentations.data.shape
        if waveform is None:
            waveform = self._load(file)
        sr = self._embedding.sample_rate
        window = self._audio.get_num_samples(duration, sr)
        step = round(binary_segmentations.sliding_window.step * sr)
        begin = 0 if chunk_range is None else chunk_range[0]
        dev_bin = self._device_segmentation(binary_segmentations)
        _, clean = self._chunk_stats(dev_bin)
        if exclude_overlap:
            # speaker_diarization.py:375-382
            min_num_samples = self._embedding.min_num_samples
            num_samples = duration * self._embedding.sample_rate
            min_num_frames = math.ceil(num_frames * min_num_samples / num_samples)
        else:
            min_num_frames = -1
        masks = frame_ops.embedding_masks(dev_bin, clean, exclude_overlap, min_num_frames)
        batch_count = math.ceil(num_chunks * num_speakers / self.embedding_batch_size)
        if hook is not None:
            hook("embeddings", None, total=batch_count, completed=0)
        wav = waveform.to(device, torch.float32).contiguous().view(-1)
        engine = self._embedding.model_.engine
        emb = engine.forward_strided(wav[begin * step:], step, num_chunks, window, masks)
        self._last_exchange = (dev_bin, emb)
        embeddings = emb.cpu().numpy()
        if hook is not None:
            hook("embeddings", embeddings, total=batch_count, completed=batch_count)
        return embeddings

    def last_exchange_payload(self, device: torch.device) -> torch.Tensor:
        """Device-resident fused byte records of the last file's per-chunk results -- uint8 hard
        segmentation (F*S bytes) followed by the fp32 embeddings (S*D*4 bytes) per chunk -- i.e. the
        send buffer of the multi-file all-gather (parallel.all_gather_chunks uses the same record)."""
        seg, emb = self._last_exchange
        C = seg.shape[0]
        seg_b = seg.to(torch.uint8).reshape(C, -1)
        emb_b = emb.contiguous().view(torch.uint8).reshape(C, -1)
        return torch.cat([seg_b, emb_b], dim=1).contiguous()

    def reconstruct(self, segmentations: SlidingWindowFeature, hard_clusters: np.ndarray,
                    count: SlidingWindowFeature) -> SlidingWindowFeature:
        """speaker_diarization.py:480-528 (per chunk, activation of cluster k = max over the local
        speakers assigned to k) followed by to_diarization (diarization.py:221-268), on the GPU."""
        rec = frame_ops.Reconstructor(self._device_segmentation(segmentations),
                                      segmentations.sliding_window,
                                      self._segmentation.model.receptive_field, hard_clusters,
                                      count.data)
        return rec.discretize()

    # -----------------------------------------------------------------------------------------
    def apply(self, file: AudioFile, num_speakers: Optional[int] = None,
              min_speakers: Optional[int] = None, max_speakers: Optional[int] = None,
              hook: Optional[Callable] = None, **kwargs):
        """speaker_diarization.py:530-784"""
        if len(kwargs) > 0:
            warnings.warn(f"Ignoring unexpected keyword arguments: {', '.join(list(kwargs.keys()))}")
        hook = self.setup_hook(file, hook=hook)
        num_speakers, min_speakers, max_speakers = set_num_speakers(
            num_speakers=num_speakers, min_speakers=min_speakers, max_speakers=max_speakers)
        if self._expects_num_speakers and num_speakers is None:
            raise ValueError(f"num_speakers must be provided when using {self.klustering} clustering")

        # no CPU path: every stage below runs through libpyannote_amd.so on a gfx950 device
        from . import ffi
        ffi.require_gpu()
        device = getattr(self.clustering, "device", None)
        if device is None or getattr(device, "type", None) != "cuda":
            raise RuntimeError("SpeakerDiarization must be moved to the GPU first: "
                               "pipeline.to(torch.device('cuda')) -- there is no CPU fallback")
        marks = [("start", time.perf_counter())]

        def mark(name):
            marks.append((name, time.perf_counter()))

        waveform = self._load(file)
        mark("load")
        shard = parallel.current_shard()
        sr = self._audio.sample_rate
        window = self._segmentation.model.audio.get_num_samples(self._segmentation.duration)
        step = round(self._segmentation.step * sr)
        n_full, has_last = Inference.num_chunks(waveform.shape[1], window, step)
        total_chunks = n_full + has_last
        chunk_range = parallel.chunk_range(total_chunks, shard)

        segmentations = self.get_segmentations(file, hook=hook, waveform=waveform,
                                               chunk_range=chunk_range)
        mark("segmentation")
        if shard.world_size == 1:
            hook("segmentation", segmentations)
        num_chunks, num_frames, local_num_speakers = segmentations.data.shape
        binarized_segmentations = segmentations  # powerset models are already hard (:598-600)

        embeddings = None
        if shard.world_size > 1:
            # every rank embeds its own chunk range, then ONE all-gather of (segmentation, embedding)
            embeddings = self.get_embeddings(file, binarized_segmentations,
                                             exclude_overlap=self.embedding_exclude_overlap,
                                             waveform=waveform, chunk_range=chunk_range)
            seg_all, embeddings = parallel.all_gather_chunks(segmentations.data, embeddings,
                                                             total_chunks, shard,
                                                             self._segmentation.device)
            segmentations = SlidingWindowFeature(
                seg_all, SlidingWindow(start=0.0, duration=self._segmentation.duration,
                                       step=self._segmentation.step))
            binarized_segmentations = segmentations
            hook("segmentation", segmentations)
            num_chunks = total_chunks

        dev_seg = self._device_segmentation(binarized_segmentations)
        count = frame_ops.speaker_count(dev_seg, binarized_segmentations.sliding_window,
                                        self._segmentation.model.receptive_field)
        mark("speaker_counting")
        hook("speaker_counting", count)

        if np.nanmax(count.data) == 0.0:
            output = DiarizeOutput(
                speaker_diarization=Annotation(uri=file["uri"]),
                exclusive_speaker_diarization=Annotation(uri=file["uri"]),
                speaker_embeddings=np.zeros((0, self._embedding.dimension)))
            return output.speaker_diarization if self.legacy else output

        if embeddings is None:
            embeddings = self.get_embeddings(file, binarized_segmentations,
                                             exclude_overlap=self.embedding_exclude_overlap, hook=hook,
                                             waveform=waveform)
        mark("embeddings")
        hook("embeddings", embeddings)

        active_frames, clean_frames = (t.cpu().numpy() for t in self._chunk_stats(dev_seg))
        hard_clusters, _, centroids = self.clustering(
            embeddings=embeddings, segmentations=binarized_segmentations, num_clusters=num_speakers,
            min_clusters=min_speakers, max_clusters=max_speakers, file=file,
            frames=self._segmentation.model.receptive_field, num_clean_frames=clean_frames)
        mark("clustering")
        num_different_speakers = np.max(hard_clusters) + 1
        if num_different_speakers < min_speakers or num_different_speakers > max_speakers:
            warnings.warn(textwrap.dedent(f"""
                The detected number of speakers ({num_different_speakers}) for {file["uri"]} is outside
                the given bounds [{min_speakers}, {max_speakers}]. This can happen if the
                given audio file is too short to contain {min_speakers} or more speakers.
                Try to lower the desired minimal number of speakers.
                """))
        count.data = np.minimum(count.data, max_speakers).astype(np.int8)

        inactive_speakers = active_frames == 0
        hard_clusters[inactive_speakers] = -2

        reconstructor = frame_ops.Reconstructor(dev_seg, segmentations.sliding_window,
                                                self._segmentation.model.receptive_field,
                                                hard_clusters, count.data)
        discrete_diarization = reconstructor.discretize()
        mark("reconstruction")
        hook("discrete_diarization", discrete_diarization)
        diarization = to_annotation(discrete_diarization, min_duration_on=0.0,
                                    min_duration_off=self.segmentation.min_duration_off)
        diarization.uri = file["uri"]

        count.data = np.minimum(count.data, 1).astype(np.int8)
        exclusive_discrete_diarization = reconstructor.discretize(cap=1)
        exclusive_diarization = to_annotation(exclusive_discrete_diarization, min_duration_on=0.0,
                                              min_duration_off=self.segmentation.min_duration_off)
        exclusive_diarization.uri = file["uri"]

        # hypothesised speakers -> SPEAKER_00, SPEAKER_01, ... in labels() order (:730-737).
        # (mapping onto a reference annotation, :718-729, needs pyannote.metrics: out of scope)
        mapping = {label: expected_label
                   for label, expected_label in zip(diarization.labels(), self.classes())}
        diarization = diarization.rename_labels(mapping=mapping)
        exclusive_diarization = exclusive_diarization.rename_labels(mapping=mapping)
        mark("annotation")
        # wall-clock per stage of the last call (host clock, no device synchronisation)
        self.timings = {b[0]: b[1] - a[1] for a, b in zip(marks[:-1], marks[1:])}

        if centroids is None:
            output = DiarizeOutput(speaker_diarization=diarization,
                                   exclusive_speaker_diarization=exclusive_diarization,
                                   speaker_embeddings=centroids)
            return output.speaker_diarization if self.legacy else output

        if len(diarization.labels()) > centroids.shape[0]:
            centroids = np.pad(centroids,
                               ((0, len(diarization.labels()) - centroids.shape[0]), (0, 0)))
        inverse_mapping = {label: index for index, label in mapping.items()}
        centroids = centroids[[inverse_mapping[label] for label in diarization.labels()]]
        output = DiarizeOutput(speaker_diarization=diarization,
                               exclusive_speaker_diarization=exclusive_diarization,
                               speaker_embeddings=centroids)
        return output.speaker_diarization if self.legacy else output
