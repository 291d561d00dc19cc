"""Speaker diarization pipeline (`pyannote/speaker-diarization-3.1` configuration), MI355X native.

Public surface = the reference's (pipelines/speaker_diarization.py: constructor :127-279, `apply`
:530-784, hooks, `DiarizeOutput` :63-124, `get_segmentations` :305-330, `get_embeddings` :332-478,
`reconstruct` :480-528; `apply_batch` is the hook `core/pipeline.py:489-508` looks for).  Inside, the
work is organised as three device stages instead of the reference's one long method:

  front end   waveform -> hard segmentations (C,F,S) + speaker count + embeddings (C,S,D), all produced
              by the HIP library from ONE upload of the waveform (`_front_end`, stream 0)
  clustering  embeddings of one file -- or of MANY files, possibly gathered from all ranks
              (`apply_batch(..., joint_clustering=True)`, BASELINE.json configs[4]) -> cluster ids
  back end    cluster ids -> overlap-add reconstruction, top-count discretisation, annotations
              (`_back_end`)

`apply_batch` pipelines files: clustering + back end of file i run on a second HIP stream in a worker
thread while the front end of file i+1 occupies the rest of the chip (the dendrogram merge is one
persistent workgroup on one CU: 15 % of a sequential step with 255 CUs idle).  `shard=(rank, world)`
(parallel.set_shard) splits the chunk range of ONE file across GPUs instead."""
from __future__ import annotations

import contextlib
import math
import os
import textwrap
import time
import warnings
from dataclasses import dataclass, field
from typing import Any, Callable, Iterable, Iterator, List, Mapping, Optional, Tuple

import numpy as np
import torch

from . import distance
from . import ffi
from . import frames as frame_ops
from . import parallel
from .clustering import Clustering
from .audio import Audio, AudioFile
from .core import Annotation, SlidingWindow, SlidingWindowFeature
from .diarization import optimal_mapping, set_num_speakers, to_annotation
from .inference import Inference
from .model import Model
from .pipeline import ParamDict, Pipeline, Uniform
from .pipelining import ReadAhead, pipelined, pipelined_owned, run_ahead
from .speaker_verification import PipelineModel, PretrainedSpeakerEmbedding, get_model


@dataclass
class DiarizeOutput:
    """speaker_diarization.py:63-124"""
    speaker_diarization: Annotation
    exclusive_speaker_diarization: Annotation
    speaker_embeddings: Optional[np.ndarray] = None

    def serialize(self) -> dict:
        def turns(annotation):
            return [{"start": round(start, 3), "end": round(end, 3), "speaker": label}
                    for start, end, _, label in annotation.flat_rows()]
        return {"diarization": turns(self.speaker_diarization),
                "exclusive_diarization": turns(self.exclusive_speaker_diarization)}


@dataclass
class _FrontEnd:
    """Everything the later stages need about one file.  Device tensors stay referenced here so that
    the back end (possibly on another stream / thread) never sees recycled memory."""
    file: dict
    chunks: SlidingWindow                          # chunk grid of the WHOLE file
    segmentations: SlidingWindowFeature            # host, float32 {0,1}, (C, F, S)
    dev_seg: torch.Tensor                          # device, uint8, (C, F, S): hard (binarized) decisions
    count: SlidingWindowFeature                    # host, uint8, (T, 1)
    dev_scores: Optional[torch.Tensor] = None      # device, float32 (C, F, S): soft scores (non-powerset)
    silent: bool = False                           # nobody speaks anywhere in the file
    active: Optional[np.ndarray] = None            # (C, S) frames a local speaker is on
    clean: Optional[np.ndarray] = None             # (C, S) frames it speaks alone
    embeddings: Optional[np.ndarray] = None        # host, float32, (C, S, D)
    dev_emb: Optional[torch.Tensor] = None
    marks: list = field(default_factory=list)
    enqueued: dict = field(default_factory=dict)   # host clock when a stage's launches were all queued


class SpeakerDiarization(Pipeline):
    # pipelined batches: longest wait (s) of a file's tail for the next file's segmentation stage (a safety net:
    # the gate is opened in a `finally`)
    TAIL_GATE_TIMEOUT = 5.0

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
        self.segmentation_step = segmentation_step
        self.embedding = embedding
        self.embedding_batch_size = embedding_batch_size
        self.embedding_exclude_overlap = embedding_exclude_overlap
        self.plda = plda
        self.klustering = clustering
        self.der_variant = der_variant or {"collar": 0.0, "skip_overlap": False}
        if clustering not in Clustering.__members__:
            raise ValueError(f"clustering must be one of [{', '.join(list(Clustering.__members__))}]")

        seg_model: Model = get_model(segmentation, token=token, cache_dir=cache_dir)
        chunk_duration = seg_model.specifications.duration
        self._segmentation = Inference(seg_model, duration=chunk_duration,
                                       step=self.segmentation_step * chunk_duration,
                                       skip_aggregation=True, batch_size=segmentation_batch_size)
        tunables = {"min_duration_off": Uniform(0.0, 1.0)}
        if not seg_model.specifications.powerset:
            tunables = {"threshold": Uniform(0.1, 0.9), **tunables}
        self.segmentation = ParamDict(**tunables)

        metric = "not_applicable"
        self._embedding = None
        self._audio = Audio(sample_rate=seg_model.audio.sample_rate, mono="downmix")
        if clustering != "OracleClustering":
            self._embedding = PretrainedSpeakerEmbedding(self.embedding, token=token,
                                                         cache_dir=cache_dir)
            self._audio = Audio(sample_rate=self._embedding.sample_rate, mono="downmix")
            metric = self._embedding.metric
        self.clustering = Clustering[clustering].value(**self._clustering_kwargs(clustering, metric))
        self._expects_num_speakers = self.clustering.expects_num_clusters
        self.timings: dict = {}
        self.batch_timeline: list = []       # apply_batch: host-clock stage boundaries per file

    def _clustering_kwargs(self, name: str, metric: str) -> dict:
        kwargs = {"metric": metric}
        if name == "VBxClustering":
            kwargs["plda"] = self.plda
        return kwargs

    def to(self, device: torch.device):
        super().to(device)
        self._audio.device = device          # the front door resamples on the pipeline's GPU
        return self

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

    # ------------------------------------------------------------------------------ geometry helpers
    @property
    def _frames(self) -> SlidingWindow:
        return self._segmentation.model.receptive_field

    def _chunk_grid(self) -> SlidingWindow:
        return SlidingWindow(start=0.0, duration=self._segmentation.duration, step=self._segmentation.step)

    def _require_device(self) -> torch.device:
        """no CPU path: every stage runs through libpyannote_amd.so on a gfx950 device"""
        ffi.require_gpu()
        device = getattr(self.clustering, "device", None)
        if device is None or getattr(device, "type", None) != "cuda":
            raise RuntimeError("SpeakerDiarization must be moved to the GPU first: "
                               "pipeline.to(torch.device('cuda')) -- there is no CPU fallback")
        return device

    def _load(self, file, raw=None) -> torch.Tensor:
        """whole file as a (1, n) fp32 tensor at the model's rate (core/inference.py:403).  A file on disk goes to the
        device as stored (16-bit samples: half the bytes) and is scaled there (`Audio.load_on_device`: same values);
        `raw` = its samples, already read ahead by `apply_batch`."""
        device = self._segmentation.device if self._segmentation is not None else None
        if "waveform" not in file and device is not None and device.type == "cuda":
            waveform, _ = self._audio.load_on_device(file, device, raw=raw)
        else:
            waveform, _ = self._audio(file)
        return waveform

    @staticmethod
    def _read_ahead(file):
        """what `apply_batch` reads one file ahead (worker thread, host only): the stored samples of a file on disk"""
        audio = file.get("audio") if "waveform" not in file else None
        return Audio._read_raw(audio) if isinstance(audio, (str, os.PathLike)) else None

    # --------------------------------------------------------------------- reference-named stage API
    def get_segmentations(self, file, hook=None, waveform: Optional[torch.Tensor] = None,
                          chunk_range=None) -> SlidingWindowFeature:
        """speaker_diarization.py:305-330"""
        progress = None
        if hook is not None:
            def progress(**kw):
                return hook("segmentation", None, **kw)
        if waveform is None:
            waveform = self._load(file)
        return self._segmentation.slide(waveform, self._audio.sample_rate, hook=progress,
                                        chunk_range=chunk_range)

    def _min_num_frames(self, num_frames: int, duration: float, exclude_overlap: bool) -> int:
        """speaker_diarization.py:375-382: shortest clean mask that still yields an embedding"""
        if not exclude_overlap:
            return -1
        sr = self._embedding.sample_rate
        return math.ceil(num_frames * self._embedding.min_num_samples / (duration * sr))

    def _embed(self, waveform: torch.Tensor, dev_seg: torch.Tensor, chunks: SlidingWindow,
               first_chunk: int, exclude_overlap: bool, hook: Optional[Callable]):
        """masks on the device -> ONE backbone pass per chunk, pooled for every local speaker
        -> ((C, S, D) device tensor, clean-frame counts).  Replaces speaker_diarization.py:384-476."""
        C, F, S = dev_seg.shape
        sr = self._embedding.sample_rate
        window = self._audio.get_num_samples(chunks.duration, sr)
        step = round(chunks.step * sr)
        active, clean = frame_ops.chunk_stats(dev_seg)
        masks = frame_ops.embedding_masks(dev_seg, clean, exclude_overlap,
                                          self._min_num_frames(F, chunks.duration, exclude_overlap))
        batches = math.ceil(C * S / self.embedding_batch_size)
        if hook is not None:
            hook("embeddings", None, total=batches, completed=0)
        wav = waveform.to(self._embedding.device, torch.float32).contiguous().view(-1)
        engine = self._embedding.model_.engine
        emb = self._embed_speech_chunks(engine, wav[first_chunk * step:], step, C, window, masks)
        return emb, active, clean, batches

    #: chunks in which NO local speaker is active skip the embedding backbone (see `_embed_speech_chunks`);
    #: PA_EMB_SKIP_INACTIVE=0 or this attribute set to False runs every chunk, as the reference does
    skip_inactive_chunks: bool = os.environ.get("PA_EMB_SKIP_INACTIVE", "1") != "0"
    #: what the last `_embed` call did: (chunks of the file, chunks that went through the backbone)
    last_embedded_chunks: tuple = (0, 0)

    def _embed_speech_chunks(self, engine, wav: torch.Tensor, step: int, C: int, window: int,
                             masks: torch.Tensor) -> torch.Tensor:
        """`engine.forward_strided` over the chunks that have at least one non-empty mask.

        The reference embeds every (chunk, speaker) pair (speaker_diarization.py:384-476), silent or not.  A mask
        that is zero everywhere pools to mean = 0 / 1e-8 and std = 0 whatever the backbone computed
        (models/blocks/pooling.py:49-61), so its embedding is the bias of the final Linear: the backbone pass of a
        chunk whose masks are ALL empty cannot change any output.  Such chunks are left out of the launch (the
        others are gathered into a compact buffer, one copy of 4 * window bytes per chunk); ONE of them is kept as
        the representative whose rows are written to all of them, so the "embeddings" artefact is what the full
        run produces, bit for bit (tests/test_pipeline_gpu.py) -- unless a backbone activation of a skipped chunk
        was not finite (0 * inf = NaN in the full run)."""
        self.last_embedded_chunks = (C, C)
        if not self.skip_inactive_chunks or masks is None or C < 3:
            return engine.forward_strided(wav, step, C, window, masks)
        speech = masks.flatten(1).any(dim=1)                       # (C,) on the device
        kept = torch.nonzero(speech).view(-1)
        num_kept = int(kept.numel())       # (host wait: the segmentation stage left the device before this call)
        if C - num_kept < 2:               # one silent chunk = the representative: nothing to save
            return engine.forward_strided(wav, step, C, window, masks)
        silent = torch.nonzero(~speech).view(-1)
        sel = torch.cat([kept, silent[:1]])
        need = (C - 1) * step + window
        if wav.numel() < need:             # the last chunk runs past the file: zeros, as the kernels assume
            wav = torch.nn.functional.pad(wav, (0, need - wav.numel()))
        compact = wav.unfold(0, window, step)[sel].contiguous()    # (num_kept + 1, window)
        out = engine.forward_strided(compact.view(-1), window, int(sel.numel()), window, masks[sel].contiguous())
        emb = torch.empty((C,) + tuple(out.shape[1:]), dtype=out.dtype, device=out.device)
        emb[silent] = out[-1]
        emb[kept] = out[:-1]
        self.last_embedded_chunks = (C, num_kept + 1)
        return emb

    def get_embeddings(self, file, binary_segmentations: SlidingWindowFeature,
                       exclude_overlap: bool = False, hook: Optional[Callable] = None,
                       waveform: Optional[torch.Tensor] = None, chunk_range=None):
        """speaker_diarization.py:332-478 -> (num_chunks, num_speakers, dimension) float32."""
        if waveform is None:
            waveform = self._load(file)
        dev_seg = frame_ops.as_device_segmentation(binary_segmentations.data, self._embedding.device)
        emb, _, _, batches = self._embed(waveform, dev_seg, binary_segmentations.sliding_window,
                                         0 if chunk_range is None else chunk_range[0],
                                         exclude_overlap, hook)
        embeddings = emb.cpu().numpy()
        if hook is not None:
            hook("embeddings", embeddings, total=batches, completed=batches)
        return embeddings

    def reconstruct(self, segmentations: SlidingWindowFeature, hard_clusters: np.ndarray,
                    count: SlidingWindowFeature) -> SlidingWindowFeature:
        """speaker_diarization.py:480-528 (per chunk, activation of cluster k = max over the local
        speakers assigned to k) followed by to_diarization (diarization.py:221-268), on the GPU."""
        dev_seg = frame_ops.as_device_segmentation(segmentations.data, self._segmentation.device)
        return frame_ops.Reconstructor(dev_seg, segmentations.sliding_window, self._frames,
                                       hard_clusters, count.data).discretize()

    # ---------------------------------------------------------------------------------- front end
    def _front_end(self, file: dict, hook: Callable, after_segmentation: Optional[Callable] = None,
                   raw=None) -> _FrontEnd:
        """`after_segmentation()` is called once the segmentation stage has LEFT the device (the pipelined
        batch forms release the previous file's tail there).  `raw`: see `_load`."""
        marks = [("start", time.perf_counter())]
        waveform = self._load(file, raw)
        marks.append(("load", time.perf_counter()))
        shard = parallel.current_shard()
        sr = self._audio.sample_rate
        window = self._segmentation.model.audio.get_num_samples(self._segmentation.duration)
        step = round(self._segmentation.step * sr)
        n_full, has_last = Inference.num_chunks(waveform.shape[1], window, step)
        total_chunks = n_full + has_last
        chunk_range = parallel.chunk_range(total_chunks, shard)
        chunks = self._chunk_grid()

        segmentations = self.get_segmentations(file, hook=hook, waveform=waveform, chunk_range=chunk_range)
        dev_seg = self._segmentation.last_device_output
        dev_scores = None
        if not self._segmentation.model.specifications.powerset:
            # non-powerset model: hysteresis thresholding of the sigmoid scores (:599-606, utils/signal.py:
            # 78-204) -- counting, masks and clustering use the binary form, reconstruction the raw scores
            if shard.world_size > 1:
                raise NotImplementedError("chunk sharding exchanges hard (powerset) segmentations only")
            dev_scores = dev_seg
            dev_seg = frame_ops.binarize(dev_scores, onset=self.segmentation.threshold, initial_state=False)
        marks.append(("segmentation", time.perf_counter()))
        if after_segmentation is not None:
            after_segmentation()
        enqueued = {"segmentation": self._segmentation.last_enqueued}

        dev_emb = active = clean = None
        if shard.world_size > 1:
            # every rank embeds its own chunk range, then ONE all-gather of (segmentation, embedding)
            # records that never leave the device with RCCL
            local_emb, _, _, _ = self._embed(waveform, dev_seg, segmentations.sliding_window,
                                             chunk_range[0], self.embedding_exclude_overlap, None)
            dev_seg, dev_emb = parallel.all_gather_chunks(dev_seg, local_emb, total_chunks, shard,
                                                          self._segmentation.device)
            dev_seg = dev_seg.contiguous()
            segmentations = SlidingWindowFeature(dev_seg.cpu().numpy().astype(np.float32), chunks)
        hook("segmentation", segmentations)

        count = frame_ops.speaker_count(dev_seg, chunks, self._frames)
        marks.append(("speaker_counting", time.perf_counter()))
        hook("speaker_counting", count)
        front = _FrontEnd(file=file, chunks=chunks, segmentations=segmentations, dev_seg=dev_seg,
                          count=count, marks=marks, enqueued=enqueued, dev_scores=dev_scores)
        if dev_scores is not None:             # the clustering filters on the BINARIZED segmentations (:642)
            front.segmentations = SlidingWindowFeature(dev_seg.cpu().numpy().astype(np.float32), chunks)
        if np.nanmax(count.data) == 0.0:
            front.silent = True
            return front                       # nobody speaks: no embeddings (:617-629)
        if self._embedding is None:            # OracleClustering: no embeddings (:631-636)
            active, clean = frame_ops.chunk_stats(dev_seg)
            front.active, front.clean = active.cpu().numpy(), clean.cpu().numpy()
            return front

        if dev_emb is None:
            dev_emb, active, clean, batches = self._embed(waveform, dev_seg, chunks, 0,
                                                          self.embedding_exclude_overlap, hook)
            front.enqueued["embeddings"] = time.perf_counter()
            front.embeddings = dev_emb.cpu().numpy()
            if hook is not None:
                hook("embeddings", front.embeddings, total=batches, completed=batches)
        else:
            active, clean = frame_ops.chunk_stats(dev_seg)
            front.embeddings = dev_emb.cpu().numpy()
        front.dev_emb = dev_emb
        front.active, front.clean = active.cpu().numpy(), clean.cpu().numpy()
        marks.append(("embeddings", time.perf_counter()))
        hook("embeddings", front.embeddings)
        return front

    # ----------------------------------------------------------------------------------- back end
    def _empty_output(self, file: dict):
        output = DiarizeOutput(speaker_diarization=Annotation(uri=file["uri"]),
                               exclusive_speaker_diarization=Annotation(uri=file["uri"]),
                               speaker_embeddings=np.zeros(
                                   (0, self._embedding.dimension if self._embedding is not None else 0)))
        return output.speaker_diarization if self.legacy else output

    def _cluster_one(self, front: _FrontEnd, num_speakers, min_speakers, max_speakers):
        hard, _, centroids = self.clustering(
            embeddings=front.embeddings, segmentations=front.segmentations, num_clusters=num_speakers,
            min_clusters=min_speakers, max_clusters=max_speakers, file=front.file, frames=self._frames,
            num_clean_frames=front.clean, device_embeddings=front.dev_emb)
        return hard, centroids

    def _back_end(self, front: _FrontEnd, hard_clusters: np.ndarray, centroids: Optional[np.ndarray],
                  min_speakers, max_speakers, hook: Callable):
        file = front.file
        marks = front.marks
        marks.append(("clustering", time.perf_counter()))
        found = int(np.max(hard_clusters)) + 1
        if found < min_speakers or found > max_speakers:
            warnings.warn(textwrap.dedent(f"""
                The detected number of speakers ({found}) for {file["uri"]} is outside
                the given bounds [{min_speakers}, {max_speakers}]. This can happen if the
                given audio file is too short to contain {min_speakers} or more speakers.
                Try to lower the desired minimal number of speakers.
                """))
        # a local speaker that never speaks belongs to no cluster (:681-685)
        hard_clusters = np.where(front.active == 0, -2, hard_clusters)
        capped = np.minimum(front.count.data, max_speakers).astype(np.int8)          # (:676)
        rec = frame_ops.Reconstructor(front.dev_seg if front.dev_scores is None else front.dev_scores,
                                      front.chunks, self._frames, hard_clusters, capped)   # (:687-691)
        regular = rec.discretize()
        marks.append(("reconstruction", time.perf_counter()))
        hook("discrete_diarization", regular)
        exclusive = rec.discretize(cap=1)                                            # (:702-707)

        gap = self.segmentation.min_duration_off
        diarization = to_annotation(regular, min_duration_on=0.0, min_duration_off=gap)
        exclusive_diarization = to_annotation(exclusive, min_duration_on=0.0, min_duration_off=gap)
        # cluster ids -> SPEAKER_00, SPEAKER_01, ... in labels() order (:730-737) -- or, when the file comes
        # with its reference annotation, -> the reference speakers they overlap most with (:718-729; output
        # unchanged otherwise, it only makes error analysis easier)
        labels = diarization.labels()
        names = self._label_names(file, diarization, labels)
        diarization = diarization.rename_labels(mapping=names)
        exclusive_diarization = exclusive_diarization.rename_labels(mapping=names)
        diarization.uri = exclusive_diarization.uri = file["uri"]
        marks.append(("annotation", time.perf_counter()))
        # wall-clock per stage of the last finished file (host clock, no device synchronisation)
        self.timings = {b[0]: b[1] - a[1] for a, b in zip(marks[:-1], marks[1:])}

        if centroids is not None:
            if len(labels) > centroids.shape[0]:                                     # (:763-766)
                centroids = np.pad(centroids, ((0, len(labels) - centroids.shape[0]), (0, 0)))
            # row order = sorted SPEAKER_xx names (:770-773)
            cluster_of = {name: label for label, name in names.items()}
            centroids = centroids[[cluster_of[name] for name in diarization.labels()]]
        output = DiarizeOutput(speaker_diarization=diarization,
                               exclusive_speaker_diarization=exclusive_diarization,
                               speaker_embeddings=centroids)
        return output.speaker_diarization if self.legacy else output

    def _label_names(self, file, diarization: Annotation, labels: list) -> dict:
        if isinstance(file, Mapping) and "annotation" in file and file["annotation"]:
            _, mapping = optimal_mapping(file["annotation"], diarization, return_mapping=True)
            return {label: mapping.get(label, label) for label in labels}   # extra speakers keep their id
        return dict(zip(labels, self.classes()))

    def _speaker_bounds(self, num_speakers, min_speakers, max_speakers, kwargs, file=None):
        """(:565-590) unknown keyword arguments are ignored with a warning; a clustering that needs the
        number of speakers (KMeans) takes it from the file's reference annotation when it has one."""
        if len(kwargs) > 0:
            warnings.warn(f"Ignoring unexpected keyword arguments: {', '.join(list(kwargs.keys()))}")
        num_speakers, min_speakers, max_speakers = set_num_speakers(
            num_speakers=num_speakers, min_speakers=min_speakers, max_speakers=max_speakers)
        if self._expects_num_speakers and num_speakers is None:
            if isinstance(file, Mapping) and "annotation" in file:
                num_speakers = len(file["annotation"].labels())   # the bounds stay (1, inf), as in the reference
            else:
                raise ValueError(f"num_speakers must be provided when using {self.klustering} clustering")
        return num_speakers, min_speakers, max_speakers

    # ---------------------------------------------------------------------------------------- apply
    def apply(self, file: AudioFile, num_speakers: Optional[int] = None,
              min_speakers: Optional[int] = None, max_speakers: Optional[int] = None,
              hook: Optional[Callable] = None, **kwargs):
        """speaker_diarization.py:530-784"""
        num_speakers, min_speakers, max_speakers = self._speaker_bounds(num_speakers, min_speakers,
                                                                        max_speakers, kwargs, file=file)
        hook = self.setup_hook(file, hook=hook)
        self._require_device()
        front = self._front_end(file, hook)
        if front.silent:
            return self._empty_output(file)
        with distance.device_to_ourselves():      # one file on its own: no front end runs beside its clustering
            hard, centroids = self._cluster_one(front, num_speakers, min_speakers, max_speakers)
        return self._back_end(front, hard, centroids, min_speakers, max_speakers, hook)

    def apply_batch(self, files: Iterable[AudioFile], num_speakers: Optional[int] = None,
                    min_speakers: Optional[int] = None, max_speakers: Optional[int] = None,
                    hook: Optional[Callable] = None, joint_clustering: bool = False,
                    **kwargs) -> Iterator[Tuple[AudioFile, Any]]:
        """Several files (core/pipeline.py:489-508 calls this for list inputs) -> (file, output) pairs
        in input order.

        Default: per-file results identical to `apply`, software-pipelined -- clustering and the back
        end of file i run on a second stream while the front end of file i+1 runs on the first.
        `joint_clustering=True`: ONE clustering over the embeddings of all files (of all ranks when
        torch.distributed is initialised and `parallel.set_shard` was not used to split single files):
        speakers get the same label in every file (BASELINE.json configs[4])."""
        files = [Audio.validate_file(f) for f in files]
        batch_level = {"num_speakers": num_speakers, "min_speakers": min_speakers,
                       "max_speakers": max_speakers, **kwargs}
        batch_level = {k: v for k, v in batch_level.items() if v is not None}
        if joint_clustering:
            # one clustering for all files: only batch-level bounds make sense
            ignored = [f["uri"] for f in files if f.get("pipeline_kwargs")]
            if ignored:
                warnings.warn("joint_clustering=True ignores the `pipeline_kwargs` of "
                              f"{', '.join(map(str, ignored))}")
            bounds = self._speaker_bounds(num_speakers, min_speakers, max_speakers, kwargs)
            yield from self._apply_jointly(files, bounds, hook, self._require_device())
            return

        def bounds_of(file) -> tuple:
            """core/pipeline.py:583 applies a file as `apply(file, **file["pipeline_kwargs"], **kwargs)`:
            each file keeps its own num/min/max_speakers (a name given both ways is a TypeError there too)."""
            own = dict(file.get("pipeline_kwargs", {}))
            both = set(own) & set(batch_level)
            if both:
                raise TypeError(f"apply() got multiple values for keyword argument {sorted(both)[0]!r}")
            merged = {**batch_level, **own}
            known = {k: merged.pop(k, None) for k in ("num_speakers", "min_speakers", "max_speakers")}
            merged.pop("hook", None)
            return self._speaker_bounds(known["num_speakers"], known["min_speakers"], known["max_speakers"],
                                        merged, file=file)

        all_bounds = [bounds_of(f) for f in files]      # (raises before any GPU work, like a bad call would)
        device = self._require_device()
        side = torch.cuda.Stream(device=device)

        # The dendrogram merge of file i is ONE workgroup that owns a CU for ~0.2 s per audio-hour while the
        # front end of file i+1 runs.  The persistent convolution kernels CLAIM their tiles at run time
        # (csrc/common.h: TileQueue), so the workgroups that cannot be placed beside the merge merely find
        # nothing left to do; with the earlier static tile partition they started a whole round late and every
        # convolution launch under the merge took 1.6x as long (tools/probes/interference_probe.py).
        # The SEGMENTATION kernels are ordinary one-round grids (k_lstm_rec: 450 workgroups that live for the whole
        # launch): the one that shares its CU with the merge is slowed and the launch ends with it (+26 %, stage
        # +20 ms per audio-hour, tools/batch_prof.py).  So the tail of file i is GATED (pipelining.pipelined): it
        # starts when the segmentation stage of file i+1 has left the device (or there is no next file) and
        # overlaps only the embedding stage.
        t_batch = time.perf_counter()
        self.batch_timeline = []               # per file: host-clock offsets (s) of the stage boundaries

        # the NEXT file is read from disk in a worker thread while the GPU runs the current one (0.2 s per audio-hour
        # of 16-bit WAV that the reference's own metric -- files on disk -> RTTM -- used to spend with the GPU idle)
        ahead = ReadAhead(files, self._read_ahead)

        def front_of(item, release: Callable):
            i, file, bounds = item
            file_hook = self.setup_hook(file, hook=hook)
            line = {"front_start": time.perf_counter() - t_batch}
            front = self._front_end(file, file_hook, after_segmentation=release, raw=ahead.take(i))
            line.update({name: stamp - t_batch for name, stamp in front.marks[1:]})
            line.update({name + "_queued": stamp - t_batch for name, stamp in front.enqueued.items()})
            line["submit"] = time.perf_counter() - t_batch
            self.batch_timeline.append(line)
            return front, file_hook, line, bounds

        def tail_of(state, alone: bool):
            front, file_hook, line, (num_speakers, min_speakers, max_speakers) = state
            line["tail_released"] = time.perf_counter() - t_batch
            if front.silent:
                out = self._empty_output(front.file)
            else:
                # `alone`: the last file's merge has the GPU to itself (distance.device_to_ourselves)
                hint = distance.device_to_ourselves() if alone else contextlib.nullcontext()
                with torch.cuda.device(device), torch.cuda.stream(side), hint:
                    hard, centroids = self._cluster_one(front, num_speakers, min_speakers, max_speakers)
                    out = self._back_end(front, hard, centroids, min_speakers, max_speakers, file_hook)
                    side.synchronize()
            line["tail_done"] = time.perf_counter() - t_batch
            return out

        items = [(i, f, b) for i, (f, b) in enumerate(zip(files, all_bounds))]

        def stream():
            for (_, file, _), out in pipelined(items, front_of, tail_of, self.TAIL_GATE_TIMEOUT):
                yield file, out

        try:
            if hook is None and len(items) > 2 and os.environ.get("PA_BATCH_RUN_AHEAD", "1") != "0":
                # The caller's loop body (the reference's benchmark: serialize() + write_rttm per file) runs while this
                # generator is suspended -- and the suspended generator is what would start the front end of the file
                # after next.  One result ahead, produced in its own thread, keeps the GPU busy meanwhile.  (Not with a
                # hook: hooks are the caller's code and expect the caller's thread.)
                yield from run_ahead(stream, depth=1, context=lambda: torch.cuda.device(device))
            else:
                yield from stream()
        finally:
            ahead.close()

    def _apply_jointly(self, files: List[dict], bounds, hook, device: torch.device):
        """front end per file; records of all files of all ranks gathered on the device; ONE clustering
        over the concatenation along the chunk axis (oracle: the same clustering called on the
        concatenated arrays); back end for the local files."""
        yield from self._joint_finish(self._joint_gather(files, hook, device), bounds)

    def apply_joint_batches(self, groups: Iterable[Iterable[AudioFile]], num_speakers: Optional[int] = None,
                            min_speakers: Optional[int] = None, max_speakers: Optional[int] = None,
                            hook: Optional[Callable] = None) -> Iterator[List[Tuple[AudioFile, Any]]]:
        """Several independent joint-clustering jobs (each `group` = the files of one
        `apply_batch(group, joint_clustering=True)` call, BASELINE.json configs[4]) software-pipelined like
        `apply_batch` pipelines files: the front ends and the record exchange of job i+1 run on the main
        stream / thread (collectives stay in one thread, in the same order on every rank) while the joint
        clustering and the back ends of job i run on a second stream.  Yields one [(file, output), ...] list
        per job, in order; results are those of the sequential calls."""
        bounds = self._speaker_bounds(num_speakers, min_speakers, max_speakers, {})
        device = self._require_device()
        side = torch.cuda.Stream(device=device)

        def gather(group, release: Callable):
            return self._joint_gather([Audio.validate_file(f) for f in group], hook, device,
                                      after_segmentation=release)

        def finish(job, alone: bool):        # (as in apply_batch: not beside a segmentation stage)
            with torch.cuda.device(device), torch.cuda.stream(side):
                out = list(self._joint_finish(job, bounds))
                side.synchronize()
            return out

        shard = parallel.shard_from_env() if parallel.current_shard().world_size == 1 else parallel.Shard()
        if shard.world_size == 1:
            for _, out in pipelined(groups, gather, finish, self.TAIL_GATE_TIMEOUT):
                yield out
            return

        # Several ranks: every rank needs the labels of job j, ONE rank has to compute them.  Done redundantly (above)
        # a stream of jobs runs at max(front end, clustering) per job -- 8 one-hour files: 0.78 s against 1.29 s, the
        # clustering sets the pace (profiles/r4_projected_n8.txt).  With owners (job j is clustered by rank j % world
        # on a third stream while its main stream goes on with the front ends, and the labels are broadcast over a
        # second process group in the tail thread) it runs at max(front end, clustering / world): the front end sets
        # the pace from two ranks on.  pipelining.pipelined_owned has the schedule.
        # The two schedules do not mix (a rank inside `pipelined` never answers the broadcasts the others wait for):
        # the choice is AGREED ON by all ranks -- the minimum of everybody's PA_JOINT_OWNERS -- and a rank that
        # cannot create the label group raises on every rank's side of that collective instead of falling back alone.
        if not parallel.agree_all(os.environ.get("PA_JOINT_OWNERS", "1") != "0", shard, device):
            for _, out in pipelined(groups, gather, finish, self.TAIL_GATE_TIMEOUT):
                yield out
            return
        group = self._label_group(shard)
        solving = torch.cuda.Stream(device=device)

        def solve(job):
            if job["empty"] or not job["all_emb_dev"].shape[0]:
                return None
            with torch.cuda.device(device), torch.cuda.stream(solving):
                solution = self._joint_cluster(job, bounds)
                solving.synchronize()
            return solution

        def share(j, owner, job, solution):
            return parallel.broadcast_object(solution, owner, shard, group, device)

        def back_ends(job, solution):
            with torch.cuda.device(device), torch.cuda.stream(side):
                if solution is None:
                    out = [(fr.file, self._empty_output(fr.file)) for fr in job["fronts"]]
                else:
                    out = list(self._joint_back_ends(job, solution[0], solution[1], bounds))
                side.synchronize()
            return out

        for _, out in pipelined_owned(groups, gather, solve, share, back_ends, shard.rank, shard.world_size,
                                      gate_timeout=self.TAIL_GATE_TIMEOUT):
            yield out

    def _label_group(self, shard):
        """the process group of the label broadcasts (the tail thread's collectives must not share a communicator
        with the main thread's record exchange); created once per default process group, by every rank together.
        A GLOO group whatever the main backend is: the labels and centroids are host arrays of a few hundred KB, and a
        host-side broadcast cannot interleave badly with the RCCL kernels of the record exchange that the main thread
        launches at the same time on the same device."""
        import torch.distributed as dist
        world = dist.group.WORLD
        cached = getattr(self, "_label_group_cache", None)
        if cached is None or cached[0] is not world:
            cached = (world, dist.new_group(list(range(shard.world_size)), backend="gloo"))
            self._label_group_cache = cached
        return cached[1]

    def _joint_gather(self, files: List[dict], hook, device: torch.device,
                      after_segmentation: Optional[Callable] = None) -> dict:
        """first half of a joint job: front ends of the local files + the exchange of the per-chunk records of
        all ranks (the only collectives of the path)"""
        hooks = [self.setup_hook(f, hook=hook) for f in files]
        # (`after_segmentation`: after the LAST local file's segmentation stage -- a joint tail is long anyway)
        fronts = [self._front_end(f, h, after_segmentation if i == len(files) - 1 else None)
                  for i, (f, h) in enumerate(zip(files, hooks))]
        voiced = [fr for fr in fronts if not fr.silent]
        shard = parallel.shard_from_env() if parallel.current_shard().world_size == 1 else parallel.Shard()
        job = {"fronts": fronts, "hooks": hooks, "voiced": voiced, "empty": False}
        if not voiced and shard.world_size == 1:
            job["empty"] = True
            return job
        F, S = (voiced[0].dev_seg.shape[1:] if voiced else (0, 0))
        D = self._embedding.dimension
        if shard.world_size > 1:
            records = [parallel.pack_records(fr.dev_seg, fr.dev_emb) for fr in voiced]
            # (F, S) travel implicitly: all ranks run the same models
            if not F:       # this rank has no voiced file of its own: shapes from the model (all ranks run the same)
                model = self._segmentation.model
                window = int(round(self._chunk_grid().duration * self._audio.sample_rate))
                F, S = int(model.num_frames(window)), len(model.specifications.classes)
            # ONE all-gather in steady state (the per-file chunk counts travel inside the buffer)
            gathered = parallel.all_gather_files(records, shard, device, record_bytes=F * S + 4 * S * D)
            segs, embs, owner = [], [], []
            for r, per_rank in enumerate(gathered):
                for j, rec in enumerate(per_rank):
                    seg, emb = parallel.unpack_records(rec, F, S, D)
                    segs.append(seg.contiguous())
                    embs.append(emb)
                    owner.append((r, j))
            mine = [owner.index((shard.rank, j)) for j in range(len(voiced))]
        else:
            segs = [fr.dev_seg for fr in voiced]
            embs = [fr.dev_emb for fr in voiced]
            mine = list(range(len(voiced)))
        all_emb_dev = torch.cat(embs, dim=0).contiguous()
        # (the host copy of the gathered embeddings -- 88 MB at eight one-hour files -- is made by the rank that
        #  clusters the job, `_joint_cluster`, not by every rank here)
        job.update(sizes=[s.shape[0] for s in segs], mine=mine, all_seg=torch.cat(segs, dim=0).contiguous(),
                   all_emb_dev=all_emb_dev)
        torch.cuda.current_stream(device).synchronize()   # the second half may run on another stream
        return job

    def _joint_finish(self, job: dict, bounds):
        """second half: ONE clustering over all records, then the back end of every local file"""
        if job["empty"]:
            for fr in job["fronts"]:
                yield fr.file, self._empty_output(fr.file)
            return
        hard, centroids = self._joint_cluster(job, bounds)
        yield from self._joint_back_ends(job, hard, centroids, bounds)

    def _joint_cluster(self, job: dict, bounds):
        """the ONE clustering of a joint job over the records of all files of all ranks -> (hard, centroids)"""
        num_speakers, min_speakers, max_speakers = bounds
        all_seg, all_emb = job["all_seg"], job["all_emb_dev"].cpu().numpy()
        _, clean = frame_ops.chunk_stats(all_seg)
        # the clustering only reads the SHAPE of the segmentations when the clean-frame counts are given
        seg_view = SlidingWindowFeature(all_seg.cpu().numpy(), self._chunk_grid())
        hard, _, centroids = self.clustering(
            embeddings=all_emb, segmentations=seg_view, num_clusters=num_speakers,
            min_clusters=min_speakers, max_clusters=max_speakers, frames=self._frames,
            num_clean_frames=clean.cpu().numpy(), device_embeddings=job.get("all_emb_dev"))
        return hard, centroids

    def _joint_back_ends(self, job: dict, hard, centroids, bounds):
        """the back end of every local file of a joint job, given the job's cluster labels"""
        _, min_speakers, max_speakers = bounds
        fronts, hooks, sizes, mine = job["fronts"], job["hooks"], job["sizes"], job["mine"]
        offsets = np.concatenate([[0], np.cumsum(sizes)])
        self.joint_hard_clusters = hard                # (sum C, S): kept for inspection / tests
        self.joint_sizes = sizes
        k = 0
        for fr, h in zip(fronts, hooks):
            if fr.silent:
                yield fr.file, self._empty_output(fr.file)
                continue
            a = offsets[mine[k]]
            k += 1
            yield fr.file, self._back_end_joint(fr, hard[a:a + fr.dev_seg.shape[0]].copy(), centroids,
                                                min_speakers, max_speakers, h)

    def _back_end_joint(self, front, hard, centroids, min_speakers, max_speakers, hook):
        """joint labels: SPEAKER_k = global cluster k in every file (no per-file renumbering)."""
        hard = np.where(front.active == 0, -2, hard)
        capped = np.minimum(front.count.data, max_speakers).astype(np.int8)
        rec = frame_ops.Reconstructor(front.dev_seg if front.dev_scores is None else front.dev_scores,
                                      front.chunks, self._frames, hard, capped)
        regular = rec.discretize()
        hook("discrete_diarization", regular)
        exclusive = rec.discretize(cap=1)
        gap = self.segmentation.min_duration_off
        names = {k: f"SPEAKER_{k:02d}" for k in range(max(regular.data.shape[1], 1))}
        out = []
        for d in (regular, exclusive):
            ann = to_annotation(d, min_duration_on=0.0, min_duration_off=gap).rename_labels(mapping=names)
            ann.uri = front.file["uri"]
            out.append(ann)
        output = DiarizeOutput(speaker_diarization=out[0], exclusive_speaker_diarization=out[1],
                               speaker_embeddings=centroids)
        return output.speaker_diarization if self.legacy else output
