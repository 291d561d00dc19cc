"""ctypes binding of libpyannote_amd.so (C ABI: include/pyannote_amd.h).

PyTorch-ROCm is used for device memory and streams only: tensors go in as raw device pointers,
kernels are launched on torch's current HIP stream.  There is NO fallback: if the library (or a GPU)
is missing, every entry point raises."""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import torch

_LIB = None
PA_MAX_LSTM_LAYERS = 8
PA_MAX_LINEAR = 4

c_fp = C.c_void_p  # device pointers travel as void*


class SegWeights(C.Structure):
    _fields_ = [
        ("sinc_stride", C.c_int32), ("lstm_layers", C.c_int32), ("lstm_hidden", C.c_int32),
        ("lstm_bidir", C.c_int32), ("num_linear", C.c_int32), ("linear_hidden", C.c_int32),
        ("num_classes", C.c_int32), ("num_speakers", C.c_int32),
        ("wav_gamma", C.c_float), ("wav_beta", C.c_float),
        ("sinc_filt", c_fp), ("norm0", c_fp), ("conv1_w", c_fp), ("conv1_b", c_fp), ("norm1", c_fp),
        ("conv2_w", c_fp), ("conv2_b", c_fp), ("norm2", c_fp),
        ("lstm_wih", c_fp * PA_MAX_LSTM_LAYERS), ("lstm_bias", c_fp * PA_MAX_LSTM_LAYERS),
        ("lstm_whh", c_fp * PA_MAX_LSTM_LAYERS),
        ("lin_w", c_fp * PA_MAX_LINEAR), ("lin_b", c_fp * PA_MAX_LINEAR),
        ("cls_w", c_fp), ("cls_b", c_fp), ("powerset_map", c_fp),
    ]


PA_MAX_RES_BLOCKS = 128


class EmbWeights(C.Structure):
    _fields_ = [
        ("num_mel", C.c_int32), ("embed_dim", C.c_int32), ("num_layers", C.c_int32),
        ("num_blocks", C.c_int32 * 4), ("planes", C.c_int32 * 4), ("bottleneck", C.c_int32),
        ("fb_window", c_fp), ("fb_tw256", c_fp), ("fb_tw512", c_fp), ("fb_mel_w", c_fp),
        ("fb_mel_lo", c_fp), ("fb_mel_hi", c_fp),
        ("stem_w", c_fp), ("stem_shift", c_fp),
        ("blk_w1", c_fp * PA_MAX_RES_BLOCKS), ("blk_shift1", c_fp * PA_MAX_RES_BLOCKS),
        ("blk_w2", c_fp * PA_MAX_RES_BLOCKS), ("blk_shift2", c_fp * PA_MAX_RES_BLOCKS),
        ("blk_u1", c_fp * PA_MAX_RES_BLOCKS), ("blk_u2", c_fp * PA_MAX_RES_BLOCKS),
        ("blk_wsc", c_fp * PA_MAX_RES_BLOCKS), ("blk_shiftsc", c_fp * PA_MAX_RES_BLOCKS),
        ("blk_w3", c_fp * PA_MAX_RES_BLOCKS), ("blk_shift3", c_fp * PA_MAX_RES_BLOCKS),
        ("seg1_w", c_fp), ("seg1_b", c_fp),
        ("blk_v1", c_fp * PA_MAX_RES_BLOCKS), ("blk_v2", c_fp * PA_MAX_RES_BLOCKS),
        ("fb_center_kernel", C.c_int32),
    ]


PA_W2V_MAX_CONV = 8
PA_W2V_MAX_LAYERS = 24


class W2vLayer(C.Structure):
    """pa_w2v_layer (include/pyannote_amd.h)"""
    _fields_ = [(n, c_fp) for n in ("qk_w", "qk_b", "v_w", "out_w", "out_b", "ln1_g", "ln1_b", "ff1_w", "ff1_b",
                                    "ff2_w", "ff2_b", "ln2_g", "ln2_b", "gate_w", "gate_b", "gate_const")]


class SserWeights(C.Structure):
    """pa_sser_weights (include/pyannote_amd.h)"""
    _fields_ = [
        ("num_conv", C.c_int32), ("conv_channels", C.c_int32 * PA_W2V_MAX_CONV),
        ("conv_kernel", C.c_int32 * PA_W2V_MAX_CONV), ("conv_stride", C.c_int32 * PA_W2V_MAX_CONV),
        ("extractor_layer_norm", C.c_int32), ("embed_dim", C.c_int32), ("num_layers", C.c_int32),
        ("num_heads", C.c_int32), ("ff_dim", C.c_int32), ("layer_norm_first", C.c_int32),
        ("pos_kernel", C.c_int32), ("pos_groups", C.c_int32), ("wavlm", C.c_int32), ("use_layer", C.c_int32),
        ("lstm_layers", C.c_int32), ("lstm_hidden", C.c_int32), ("lstm_bidir", C.c_int32),
        ("num_linear", C.c_int32), ("linear_hidden", C.c_int32), ("num_classes", C.c_int32),
        ("num_speakers", C.c_int32),
        ("conv_w", c_fp * PA_W2V_MAX_CONV), ("conv_b", c_fp * PA_W2V_MAX_CONV),
        ("conv_norm_g", c_fp * PA_W2V_MAX_CONV), ("conv_norm_b", c_fp * PA_W2V_MAX_CONV),
        ("proj_ln_g", c_fp), ("proj_ln_b", c_fp), ("proj_w", c_fp), ("proj_b", c_fp),
        ("pos_w", c_fp), ("pos_b", c_fp), ("enc_ln_g", c_fp), ("enc_ln_b", c_fp),
        ("layers", W2vLayer * PA_W2V_MAX_LAYERS), ("layer_mix", C.c_float * PA_W2V_MAX_LAYERS),
        ("lstm_wih", c_fp * PA_MAX_LSTM_LAYERS), ("lstm_bias", c_fp * PA_MAX_LSTM_LAYERS),
        ("lstm_whh", c_fp * PA_MAX_LSTM_LAYERS),
        ("lin_w", c_fp * PA_MAX_LINEAR), ("lin_b", c_fp * PA_MAX_LINEAR),
        ("cls_w", c_fp), ("cls_b", c_fp), ("powerset_map", c_fp),
    ]


PA_XVEC_TDNN = 5


class XvecWeights(C.Structure):
    """pa_xvec_weights (include/pyannote_amd.h)"""
    _fields_ = [
        ("sinc_stride", C.c_int32), ("dimension", C.c_int32),
        ("tdnn_channels", C.c_int32 * PA_XVEC_TDNN), ("tdnn_kernel", C.c_int32 * PA_XVEC_TDNN),
        ("tdnn_dilation", C.c_int32 * PA_XVEC_TDNN),
        ("wav_gamma", C.c_float), ("wav_beta", C.c_float),
        ("sinc_filt", c_fp), ("norm0", c_fp), ("conv1_w", c_fp), ("conv1_b", c_fp), ("norm1", c_fp),
        ("conv2_w", c_fp), ("conv2_b", c_fp), ("norm2", c_fp),
        ("tdnn_w", c_fp * PA_XVEC_TDNN), ("tdnn_b", c_fp * PA_XVEC_TDNN),
        ("bn_scale", c_fp), ("bn_shift", c_fp), ("emb_w", c_fp), ("emb_b", c_fp),
    ]


class LibraryNotBuilt(RuntimeError):
    pass


def lib_path() -> Path:
    # PA_LIB: development aid (e.g. an instrumented build next to the product library)
    import os
    if os.environ.get("PA_LIB"):
        return Path(os.environ["PA_LIB"])
    return Path(__file__).resolve().parent / "libpyannote_amd.so"


def load():
    """Load (once) and return the ctypes handle.  Raises LibraryNotBuilt if the .so is absent --
    the product never silently degrades to a CPU path."""
    global _LIB
    if _LIB is not None:
        return _LIB
    p = lib_path()
    if not p.exists():
        raise LibraryNotBuilt(
            f"{p} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    lib = C.CDLL(str(p))
    lib.pa_last_error.restype = C.c_char_p
    lib.pa_version.restype = C.c_int
    lib.pa_seg_workspace_bytes.restype = C.c_size_t
    lib.pa_seg_workspace_bytes.argtypes = [C.POINTER(SegWeights), C.c_int, C.c_int]
    lib.pa_seg_workspace_bytes_strided.restype = C.c_size_t
    lib.pa_seg_workspace_bytes_strided.argtypes = [C.POINTER(SegWeights), C.c_int, C.c_int, C.c_int64]
    lib.pa_seg_num_frames.argtypes = [C.c_int, C.c_int]
    lib.pa_seg_forward.argtypes = [C.POINTER(SegWeights), c_fp, C.c_int64, C.c_int64, C.c_int, C.c_int,
                                   c_fp, c_fp, c_fp, C.c_size_t, c_fp]
    lib.pa_row_stats.argtypes = [c_fp, C.c_long, C.c_long, C.c_int, C.c_int, C.c_float, c_fp, c_fp, c_fp]
    lib.pa_sinc_fir_pool.argtypes = [c_fp, C.c_long, C.c_long, C.c_int, C.c_int, C.c_int, c_fp, c_fp,
                                     C.c_float, C.c_float, c_fp, c_fp, c_fp]
    lib.pa_sinc_fir_span.argtypes = [c_fp, C.c_long, C.c_long, c_fp, c_fp, c_fp]
    lib.pa_sinc_fix_pool.argtypes = [c_fp, C.c_long, C.c_int, C.c_int, C.c_int, c_fp, c_fp, C.c_float, C.c_float,
                                     c_fp, c_fp, c_fp, c_fp]
    lib.pa_conv5_pool.argtypes = [c_fp, C.c_int, C.c_int, C.c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp,
                                  c_fp, c_fp]
    lib.pa_norm_transpose.argtypes = [c_fp, C.c_int, C.c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp]
    lib.pa_gemm_tn_ex.argtypes = [c_fp, C.c_int, c_fp, C.c_int, c_fp, c_fp, c_fp, C.c_long, C.c_int, C.c_int,
                                  C.c_int, C.c_int, C.c_int, c_fp]
    lib.pa_gemm_tn.argtypes = [c_fp, C.c_int, c_fp, C.c_int, c_fp, c_fp, C.c_long, C.c_int, C.c_int,
                               C.c_int, C.c_int, C.c_int, c_fp]
    lib.pa_lstm_rec.argtypes = [c_fp, c_fp, c_fp, C.c_int, C.c_int, C.c_int, c_fp]
    lib.pa_classifier.argtypes = [c_fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_fp, c_fp,
                                  C.c_int, c_fp, C.c_int, c_fp, c_fp, c_fp]
    lib.pa_prof_enable.argtypes = [C.c_int]
    lib.pa_prof_enable.restype = None
    lib.pa_prof_report.argtypes = [C.c_char_p, C.c_size_t]
    lib.pa_prof_report.restype = C.c_size_t
    _declare_optional(lib)
    _LIB = lib
    return lib


# (name, argtypes, restype) of entry points added by later kernel files; declared when present so
# that a partially built library fails at call time with a clear message rather than at import.
_OPTIONAL: list[tuple] = [
    ("pa_emb_num_fbank_frames", [C.c_int], C.c_int),
    ("pa_emb_num_pool_frames", [C.POINTER(EmbWeights), C.c_int], C.c_int),
    ("pa_emb_workspace_bytes", [C.POINTER(EmbWeights), C.c_int, C.c_int, C.c_int], C.c_size_t),
    ("pa_emb_forward", [C.POINTER(EmbWeights), c_fp, C.c_int64, C.c_int64, C.c_int, C.c_int, c_fp,
                        C.c_int, C.c_int, c_fp, c_fp, c_fp, C.c_size_t, c_fp], C.c_int),
    ("pa_emb_calibrate_workspace_bytes", [C.POINTER(EmbWeights), C.c_int, C.c_int], C.c_size_t),
    ("pa_emb_calibrate_winograd", [C.POINTER(EmbWeights), c_fp, C.c_int64, C.c_int64, C.c_int, C.c_int, c_fp, c_fp,
                                   c_fp, C.c_size_t, c_fp], C.c_int),
    ("pa_lstm_rec_h", [c_fp, c_fp, c_fp, C.c_int, C.c_int, C.c_int, C.c_int, c_fp], C.c_int),
    ("pa_absmax_diff", [c_fp, c_fp, C.c_long, c_fp, c_fp], C.c_int),
    ("pa_gemm_tn_s2", [c_fp, C.c_int, C.c_int, C.c_int, C.c_int, c_fp, C.c_int, c_fp, c_fp, C.c_long, C.c_int,
                       c_fp], C.c_int),
    ("pa_fbank", [c_fp, C.c_long, C.c_long, C.c_int, C.c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp,
                  C.c_int, c_fp, C.c_int, c_fp], C.c_int),
    ("pa_fbank_center_span", [c_fp, C.c_int, C.c_int, C.c_int, C.c_int, c_fp, c_fp], C.c_int),
    ("pa_resnet_stem", [c_fp, C.c_int, C.c_int, C.c_int, c_fp, c_fp, c_fp, c_fp], C.c_int),
    ("pa_conv3x3", [c_fp, C.c_int, C.c_int, C.c_int, C.c_int, c_fp, c_fp, c_fp, c_fp, C.c_int, C.c_int,
                    C.c_int, c_fp], C.c_int),
    ("pa_conv3x3_wino", [c_fp, C.c_int, C.c_int, C.c_int, C.c_int, c_fp, c_fp, c_fp, c_fp, C.c_int, C.c_int,
                         c_fp], C.c_int),
    ("pa_conv3x3_wino4", [c_fp, C.c_int, C.c_int, C.c_int, C.c_int, c_fp, c_fp, c_fp, c_fp, C.c_int, C.c_int,
                          c_fp], C.c_int),
    ("pa_conv3x3_wino4_rows", [c_fp, C.c_int, C.c_int, C.c_int, C.c_int, c_fp, c_fp, c_fp, c_fp, C.c_int, C.c_int,
                               C.c_int, c_fp], C.c_int),
    ("pa_conv3x3_wino_rows", [c_fp, C.c_int, C.c_int, C.c_int, C.c_int, c_fp, c_fp, c_fp, c_fp, C.c_int, C.c_int,
                              C.c_int, c_fp], C.c_int),
    ("pa_pdist_f64", [c_fp, C.c_int, C.c_int, c_fp, c_fp], C.c_int),
    ("pa_cdist_cosine_f64", [c_fp, C.c_int, c_fp, C.c_int, C.c_int, c_fp, c_fp, c_fp], C.c_int),
    ("pa_centroid_means", [c_fp, C.c_int, c_fp, c_fp, C.c_int, c_fp, c_fp], C.c_int),
    ("pa_linkage_workspace_bytes", [C.c_int], C.c_size_t),
    ("pa_linkage_centroid_f64", [c_fp, C.c_int, c_fp, c_fp, C.c_size_t, c_fp], C.c_int),
    ("pa_linkage_centroid_f64_ex", [c_fp, C.c_int, c_fp, c_fp, C.c_size_t, C.c_int, c_fp], C.c_int),
    ("pa_seg_chunk_stats", [c_fp, C.c_int, C.c_int, C.c_int, c_fp, c_fp, c_fp], C.c_int),
    ("pa_embedding_masks", [c_fp, C.c_int, C.c_int, C.c_int, c_fp, C.c_int, C.c_int, c_fp, c_fp], C.c_int),
    ("pa_speaker_count", [c_fp, C.c_int, C.c_int, C.c_int, c_fp, C.c_int, c_fp, c_fp, c_fp], C.c_int),
    ("pa_cluster_activations", [c_fp, C.c_int, C.c_int, C.c_int, c_fp, c_fp, C.c_int, C.c_int, c_fp, c_fp],
     C.c_int),
    ("pa_topk_binarize", [c_fp, c_fp, C.c_int, C.c_int, C.c_int, c_fp, c_fp, c_fp], C.c_int),
    ("pa_topk_binarize_f32", [c_fp, c_fp, C.c_int, C.c_int, C.c_int, c_fp, c_fp, c_fp], C.c_int),
    ("pa_binarize_hysteresis", [c_fp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, c_fp, c_fp],
     C.c_int),
    ("pa_cluster_max", [c_fp, C.c_int, C.c_int, C.c_int, c_fp, C.c_int, c_fp, c_fp], C.c_int),
    ("pa_aggregate", [c_fp, C.c_int, C.c_int, C.c_int, c_fp, C.c_int, c_fp, c_fp, C.c_float, C.c_float, C.c_int,
                      c_fp, c_fp], C.c_int),
    ("pa_resample_poly", [c_fp, C.c_long, c_fp, C.c_int, C.c_int, C.c_int, C.c_int, c_fp, C.c_long, c_fp],
     C.c_int),
    ("pa_plda_transform", [c_fp, C.c_int, C.c_int, C.c_int, C.c_int, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp],
     C.c_int),
    ("pa_vbx_workspace_bytes", [C.c_int, C.c_int, C.c_int], C.c_size_t),
    ("pa_vbx_iteration", [c_fp, c_fp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, c_fp, c_fp,
                          c_fp, C.c_size_t, c_fp], C.c_int),
    ("pa_stats_pool", [c_fp, C.c_int, C.c_int, C.c_int, C.c_int, c_fp, C.c_int, C.c_int, c_fp, c_fp,
                       c_fp], C.c_int),
    ("pa_stats_pool_rows", [c_fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_fp, C.c_int, C.c_int, c_fp,
                            c_fp, C.c_int, c_fp, c_fp, c_fp], C.c_int),
    ("pa_sser_num_frames", [C.POINTER(SserWeights), C.c_int], C.c_int),
    ("pa_sser_workspace_bytes", [C.POINTER(SserWeights), C.c_int, C.c_int], C.c_size_t),
    ("pa_sser_forward", [C.POINTER(SserWeights), c_fp, C.c_int64, C.c_int64, C.c_int, C.c_int, c_fp, c_fp, c_fp,
                         c_fp, C.c_size_t, c_fp], C.c_int),
    ("pa_xvec_num_frames", [C.POINTER(XvecWeights), C.c_int], C.c_int),
    ("pa_xvec_workspace_bytes", [C.POINTER(XvecWeights), C.c_int, C.c_int, C.c_int], C.c_size_t),
    ("pa_xvec_forward", [C.POINTER(XvecWeights), c_fp, C.c_int64, C.c_int64, C.c_int, C.c_int, c_fp,
                         C.c_int, C.c_int, c_fp, c_fp, c_fp, C.c_size_t, c_fp], C.c_int),
]


def _declare_optional(lib):
    for name, argtypes, restype in _OPTIONAL:
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.argtypes = argtypes
            if restype is not None:
                fn.restype = restype


def ptr(t: torch.Tensor | None):
    """Raw device pointer of a contiguous CUDA(HIP) tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("libpyannote_amd kernels need device tensors (got a CPU tensor)")
    if not t.is_contiguous():
        raise RuntimeError("libpyannote_amd kernels need contiguous tensors")
    return C.c_void_p(t.data_ptr())


def stream():
    """HIP stream the kernels are launched on = torch's current stream of the CURRENT device; every
    launch path runs under `on_device` so that this is the device the tensors live on."""
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def on_device(get_device):
    """Decorator: run the wrapped launcher with `get_device(*args, **kwargs)` as the current HIP device
    (kernel launches, hipMemsetAsync and the profiler's events go to the current device's context;
    with `pipeline.to(torch.device("cuda:1"))` the tensors are on GPU 1 while the caller's current
    device may still be 0)."""
    import functools

    def deco(fn):
        @functools.wraps(fn)
        def wrapper(*args, **kwargs):
            dev = get_device(*args, **kwargs)
            if dev is not None and getattr(dev, "type", None) == "cuda" and torch.cuda.is_available() \
                    and dev.index is not None and dev.index != torch.cuda.current_device():
                with torch.cuda.device(dev):
                    return fn(*args, **kwargs)
            return fn(*args, **kwargs)
        return wrapper
    return deco


def check(rc: int, what: str = ""):
    if rc == 0:
        return
    msg = load().pa_last_error().decode()
    if rc == 2:
        raise MemoryError(f"{what}: {msg}")
    if rc == 3:
        raise ValueError(f"{what}: {msg}")
    raise RuntimeError(f"{what}: {msg}")


def prof_enable(on: bool = True):
    load().pa_prof_enable(1 if on else 0)


def prof_report() -> dict:
    """per-kernel {"launches", "ms", "flops", "bytes"} since the last report (HIP events on the
    launch stream; algorithmic flops / bytes as declared by each launcher)."""
    import json
    buf = C.create_string_buffer(1 << 16)
    load().pa_prof_report(buf, len(buf))
    return json.loads(buf.value.decode() or "{}")


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("pyannote_audio_amd needs an AMD GPU (gfx950): torch.cuda.is_available() "
                           "is False and there is no CPU fallback")
    load()
