"""Weight pre-packing: reference state-dict layout -> device images the gfx950 kernels read.

Done once at model load on the host (torch CPU index ops), then uploaded; the per-call path never
touches these again.  Layouts are documented in include/pyannote_amd.h and DESIGN.md.

State-dict layout accepted (core/model.py:244-262):
  PyanNet : sincnet.wav_norm1d.*, sincnet.conv1d.0.filterbank.{low_hz_,band_hz_},
            sincnet.conv1d.{1,2}.*, sincnet.norm1d.{0,1,2}.*, lstm.* (monolithic or ModuleList),
            linear.{i}.*, classifier.*
"""
from __future__ import annotations

import functools
import hashlib
import math
import os
from typing import Optional

import ctypes as C
from itertools import combinations

import numpy as np
import torch

from . import ffi


# ---------------------------------------------------------------------------------------------
# ParamSincFB filter synthesis (asteroid_filterbanks 0.4.0 `ParamSincFB.filters`, third party:
# restated from the published algorithm; models/blocks/sincnet.py:58-69 is the call site)
# ---------------------------------------------------------------------------------------------
def sinc_filters(low_hz_: torch.Tensor, band_hz_: torch.Tensor, kernel_size: int = 251,
                 sample_rate: float = 16000.0, min_low_hz: float = 50, min_band_hz: float = 50
                 ) -> torch.Tensor:
    """(n_filters/2, 1) x2 learnable scalars -> (n_filters, kernel_size) fp32 taps: first the
    cosine-phase (even) filters, then the sine-phase (odd) ones."""
    low_hz_ = low_hz_.detach().float().cpu().view(-1, 1)
    band_hz_ = band_hz_.detach().float().cpu().view(-1, 1)
    half = kernel_size // 2
    window = torch.from_numpy(np.hamming(kernel_size)[:half]).float()
    n = 2 * np.pi * (torch.arange(-half, 0.0).view(1, -1) / float(sample_rate))
    low = min_low_hz + torch.abs(low_hz_)
    high = torch.clamp(low + min_band_hz + torch.abs(band_hz_), min_low_hz, float(sample_rate) / 2)
    band = (high - low)[:, 0]
    ft_low, ft_high = torch.matmul(low, n), torch.matmul(high, n)
    cos_left = ((torch.sin(ft_high) - torch.sin(ft_low)) / (n / 2)) * window
    cos = torch.cat([cos_left, 2 * band.view(-1, 1), torch.flip(cos_left, dims=[1])], dim=1)
    sin_left = ((torch.cos(ft_low) - torch.cos(ft_high)) / (n / 2)) * window
    sin = torch.cat([sin_left, torch.zeros_like(band.view(-1, 1)), -torch.flip(sin_left, dims=[1])],
                    dim=1)
    cos = cos / (2 * band[:, None])
    sin = sin / (2 * band[:, None])
    return torch.cat([cos, sin], dim=0).contiguous()


def powerset_mapping(num_classes: int, max_set_size: int) -> torch.Tensor:
    """utils/powerset.py:80-109: rows = powerset classes in `combinations` order."""
    rows = []
    for size in range(0, max_set_size + 1):
        for cs in combinations(range(num_classes), size):
            r = [0] * num_classes
            for c in cs:
                r[c] = 1
            rows.append(r)
    return torch.tensor(rows, dtype=torch.uint8)


def _mfma_b_image(wk: torch.Tensor, n_tiles: int) -> torch.Tensor:
    """wk: (16*n_tiles, K) with K % 4 == 0 -> [tile][kt][lane] where lane = kq*16 + n holds
    wk[16*tile + n][4*kt + kq] (the B operand of v_mfma_f32_16x16x4_f32)."""
    n16, K = wk.shape
    assert n16 == 16 * n_tiles and K % 4 == 0
    return wk.view(n_tiles, 16, K // 4, 4).permute(0, 2, 3, 1).contiguous().view(-1)


def _lstm_row_perm() -> torch.Tensor:
    """perm[col'] = torch gate row, col' = w*128 + (q*2+s)*16 + n  <->  q*128 + 32w + 16s + n."""
    perm = torch.empty(512, dtype=torch.long)
    for w in range(4):
        for q in range(4):
            for s in range(2):
                for n in range(16):
                    perm[w * 128 + (q * 2 + s) * 16 + n] = q * 128 + 32 * w + 16 * s + n
    return perm


def _lstm_whh_image(whh: torch.Tensor) -> torch.Tensor:
    """(512,128) weight_hh -> [w][q8][kt][lane]: W[q*128+32w+16s+(lane&15)][(lane>>4)*32+kt]."""
    w_ = torch.arange(4).view(4, 1, 1, 1, 1)
    q8 = torch.arange(8).view(1, 8, 1, 1, 1)
    kt = torch.arange(32).view(1, 1, 32, 1, 1)
    kq = torch.arange(4).view(1, 1, 1, 4, 1)
    n = torch.arange(16).view(1, 1, 1, 1, 16)
    rows = (q8 // 2) * 128 + 32 * w_ + 16 * (q8 % 2) + n
    cols = kq * 32 + kt
    rows, cols = torch.broadcast_tensors(rows, cols)
    return whh[rows, cols].contiguous().view(-1)


class SegmentationPack:
    """Device-resident, kernel-ready PyanNet weights + the `pa_seg_weights` struct."""

    def __init__(self, state_dict: dict, hparams: dict, num_classes: int, num_speakers: int,
                 max_set_size: int, device: torch.device):
        sd = {k: v.detach().float().cpu() for k, v in state_dict.items()}
        sinc = {"stride": 10, **(hparams.get("sincnet") or {})}
        lstm = {"hidden_size": 128, "num_layers": 2, "bidirectional": True, "monolithic": True,
                **(hparams.get("lstm") or {})}
        linear = {"hidden_size": 128, "num_layers": 2, **(hparams.get("linear") or {})}
        if int(sinc["stride"]) not in SINC_STRIDES:
            raise NotImplementedError(f"SincNet stride {sinc['stride']}: the kernel is built for {SINC_STRIDES}")
        H, ndir, lw = lstm_geometry(lstm, linear)
        self.device = device
        self._keep: list[torch.Tensor] = []
        w = ffi.SegWeights()
        w.lstm_layers = L = int(lstm["num_layers"])
        w.lstm_hidden, w.lstm_bidir = H, int(ndir == 2)
        w.num_linear, w.linear_hidden = int(linear["num_layers"]), lw
        w.num_classes, w.num_speakers = num_classes, num_speakers
        self.sinc_taps = pack_sincnet(sd, w, self._up)  # (80, 251) kept for tests
        w.sinc_stride = int(sinc["stride"])

        pack_lstm_head(sd, w, lstm, self._up)
        # max_set_size None / 0 = a multi-label (non-powerset) checkpoint: sigmoid scores, no look-up table
        self.powerset = bool(max_set_size)
        if self.powerset:
            self.mapping = powerset_mapping(num_speakers, max_set_size)
            assert self.mapping.shape[0] == num_classes
            w.powerset_map = self._up(self.mapping)
        else:
            assert num_classes == num_speakers
            self.mapping = None
            w.powerset_map = None
        self.struct = w

    def _up(self, t: torch.Tensor):
        d = t.contiguous().to(self.device)
        self._keep.append(d)
        return C.c_void_p(d.data_ptr())


def _lstm_row_perm_gen(H: int) -> torch.Tensor:
    """pa_lstm_rec_h: perm[(4 u + q) * 16 + n] = torch gate row q H + 16 u + n"""
    u = torch.arange(H // 16).view(-1, 1, 1)
    q = torch.arange(4).view(1, 4, 1)
    n = torch.arange(16).view(1, 1, 16)
    return (q * H + 16 * u + n).reshape(-1)


def _lstm_whh_image_gen(whh: torch.Tensor) -> torch.Tensor:
    """(4H, H) weight_hh -> [u][q][k4][lane][j] = W[q H + 16 u + (lane & 15)][16 k4 + 4 j + (lane >> 4)]
    (the operand stream of k_lstm_rec_gen, csrc/seg_lstm.hip)"""
    H = whh.shape[1]
    u = torch.arange(H // 16).view(-1, 1, 1, 1, 1)
    q = torch.arange(4).view(1, 4, 1, 1, 1)
    k4 = torch.arange(H // 16).view(1, 1, -1, 1, 1)
    lane = torch.arange(64).view(1, 1, 1, 64, 1)
    j = torch.arange(4).view(1, 1, 1, 1, 4)
    rows = q * H + 16 * u + (lane & 15)
    cols = 16 * k4 + 4 * j + (lane >> 4)
    rows, cols = torch.broadcast_tensors(rows, cols)
    return whh[rows, cols].contiguous().view(-1)


def lstm_geometry(lstm: dict, linear: dict) -> tuple:
    """(hidden size, directions, linear width) the kernels run, or NotImplementedError naming the constraint
    (PyanNet.py:64-72 / SSeRiouSS.py:141-170 accept any nn.LSTM / Linear configuration)"""
    H, bidir = int(lstm["hidden_size"]), bool(lstm["bidirectional"])
    if H < 16 or H % 16 or H > 512 or (not bidir and H % 32):
        raise NotImplementedError(f"LSTM hidden size {H}: the kernels need a multiple of 16 (32 when "
                                  "unidirectional) up to 512")
    lw = int(linear["hidden_size"])
    if int(linear["num_layers"]) > 0 and (lw < 32 or lw % 32):
        raise NotImplementedError(f"Linear width {lw}: the kernels need a multiple of 32")
    return H, 2 if bidir else 1, lw


def pack_lstm_head(sd: dict, w, lstm: dict, up):
    """bi-LSTM stack (PyTorch gate order i, f, g, o; monolithic `lstm.weight_ih_l{k}` or split
    `lstm.{k}.weight_ih_l0` keys, PyanNet.py:98-123 / SSeRiouSS.py:141-170), Linear head and classifier ->
    the operand images shared by pa_seg_weights and pa_sser_weights."""
    L = int(lstm["num_layers"])
    H, ndir = int(w.lstm_hidden), 2 if w.lstm_bidir else 1
    fast = H == 128 and ndir == 2          # the register-resident kernel and its operand layouts
    perm = _lstm_row_perm() if fast else _lstm_row_perm_gen(H)
    image = _lstm_whh_image if fast else _lstm_whh_image_gen
    for l in range(L):
        if lstm["monolithic"]:
            key = lambda name, rev: f"lstm.{name}_l{l}" + ("_reverse" if rev else "")
        else:
            key = lambda name, rev: f"lstm.{l}.{name}_l0" + ("_reverse" if rev else "")
        wih, bias, whh = [], [], []
        for rev in (False, True)[:ndir]:
            wi = sd[key("weight_ih", rev)][perm]
            if wi.shape[1] % 32:            # (layer 0 of PyanNet: 60 SincNet channels -> the GEMM's K = 64)
                wi = torch.nn.functional.pad(wi, (0, 32 - wi.shape[1] % 32))
            wih.append(wi)
            bias.append((sd[key("bias_ih", rev)] + sd[key("bias_hh", rev)])[perm])
            whh.append(image(sd[key("weight_hh", rev)]))
        w.lstm_wih[l] = up(torch.cat(wih, 0)).value
        w.lstm_bias[l] = up(torch.cat(bias, 0)).value
        w.lstm_whh[l] = up(torch.cat(whh, 0)).value
    for l in range(w.num_linear):
        w.lin_w[l] = up(sd[f"linear.{l}.weight"]).value
        w.lin_b[l] = up(sd[f"linear.{l}.bias"]).value
    w.cls_w = up(sd["classifier.weight"])
    w.cls_b = up(sd["classifier.bias"])


#: SincNet strides k_sinc_fir_pool is instantiated for (csrc/seg_frontend.hip; sincnet.py:58-69 accepts any, the
#: released checkpoints use 10 -- only that one takes the shared per-span sinc layer)
SINC_STRIDES = (1, 2, 4, 5, 8, 10, 16, 20)


def pack_sincnet(sd: dict, w, up) -> torch.Tensor:
    """SincNet weights (models/blocks/sincnet.py:40-80) -> the MFMA operand images shared by
    pa_seg_weights and pa_xvec_weights; returns the (80, 251) taps."""
    w.sinc_stride = 10
    w.wav_gamma = float(sd["sincnet.wav_norm1d.weight"][0])
    w.wav_beta = float(sd["sincnet.wav_norm1d.bias"][0])
    taps = sinc_filters(sd["sincnet.conv1d.0.filterbank.low_hz_"], sd["sincnet.conv1d.0.filterbank.band_hz_"])
    w.sinc_filt = up(_mfma_b_image(torch.nn.functional.pad(taps, (0, 1)), 5))
    for i, c in enumerate((80, 60, 60)):
        setattr(w, f"norm{i}", up(torch.cat([sd[f"sincnet.norm1d.{i}.weight"], sd[f"sincnet.norm1d.{i}.bias"]])))
    for i, cin in ((1, 80), (2, 60)):
        cw = sd[f"sincnet.conv1d.{i}.weight"]  # (60, cin, 5)
        wk = torch.zeros(64, 5 * cin)
        wk[:60] = cw.permute(0, 2, 1).reshape(60, 5 * cin)  # k = tap*cin + c
        setattr(w, f"conv{i}_w", up(_mfma_b_image(wk, 4)))
        cb = torch.zeros(64)
        cb[:60] = sd[f"sincnet.conv1d.{i}.bias"]
        setattr(w, f"conv{i}_b", up(cb))
    return taps


#: constructor arguments of torchaudio's WavLM-base bundles (torchaudio.pipelines.WAVLM_BASE / _BASE_PLUS ->
#: torchaudio.models.wavlm_base()), what SSeRiouSS's default `wav2vec="WAVLM_BASE"` builds (SSeRiouSS.py:100-109)
WAV2VEC_BUNDLES = {
    name: dict(extractor_mode="group_norm",
               extractor_conv_layer_config=[(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512, 2, 2)] * 2,
               extractor_conv_bias=False, encoder_embed_dim=768, encoder_pos_conv_kernel=128,
               encoder_pos_conv_groups=16, encoder_num_layers=12, encoder_num_heads=12, encoder_num_buckets=320,
               encoder_max_distance=800, encoder_ff_interm_features=3072, encoder_layer_norm_first=False, wavlm=True)
    for name in ("WAVLM_BASE", "WAVLM_BASE_PLUS")}


def wav2vec_config(wav2vec) -> dict:
    """hyper-parameter `wav2vec` of SSeRiouSS (a bundle name, or the keyword arguments of
    torchaudio.models.wav2vec2_model) -> architecture description"""
    if isinstance(wav2vec, str):
        if wav2vec in WAV2VEC_BUNDLES:
            return dict(WAV2VEC_BUNDLES[wav2vec])
        import os
        if os.path.isfile(wav2vec):
            # a self-supervised checkpoint {"config": wav2vec2_model kwargs, "state_dict": ...}
            # (SSeRiouSS.py:111-119); its weights are superseded by the model checkpoint's own `wav2vec.*` entries,
            # only the architecture is read here, so the safe loader suffices (plain containers + tensors)
            checkpoint = torch.load(wav2vec, map_location="cpu", weights_only=True)
            if "config" not in checkpoint:
                raise ValueError(f"wav2vec checkpoint {wav2vec!r} has no 'config' entry (SSeRiouSS.py:113)")
            wav2vec = checkpoint["config"]
        else:
            raise NotImplementedError(f"wav2vec {wav2vec!r}: neither one of the built torchaudio bundles "
                                      f"{sorted(WAV2VEC_BUNDLES)} nor the path of a checkpoint with a 'config' entry; "
                                      "explicit wav2vec2_model configurations (a dict) work too")
    cfg = dict(wav2vec)
    cfg.setdefault("extractor_conv_layer_config", None)
    if cfg["extractor_conv_layer_config"] is None:
        cfg["extractor_conv_layer_config"] = [(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512, 2, 2)] * 2
    cfg["wavlm"] = False
    return cfg


def relative_position_bucket(relative_positions: torch.Tensor, num_buckets: int, max_distance: int) -> torch.Tensor:
    """WavLM's bidirectional T5-style bucketing of k - q (wavlm_attention.py `_relative_positions_bucket`)"""
    num_buckets //= 2
    buckets = (relative_positions > 0).to(torch.long) * num_buckets
    rel = torch.abs(relative_positions)
    max_exact = num_buckets // 2
    is_small = rel < max_exact
    large = max_exact + (torch.log(rel.float() / max_exact) / math.log(max_distance / max_exact)
                         * (num_buckets - max_exact)).to(torch.long)
    large = torch.min(large, torch.full_like(large, num_buckets - 1))
    return buckets + torch.where(is_small, rel, large)


class SSeRiouSSPack:
    """Device-resident, kernel-ready SSeRiouSS weights + the `pa_sser_weights` struct
    (models/segmentation/SSeRiouSS.py:84-215; state-dict names of torchaudio's wav2vec2 / WavLM modules:
    wav2vec.feature_extractor.conv_layers.{i}.{conv,layer_norm}, wav2vec.encoder.feature_projection.*,
    wav2vec.encoder.transformer.{pos_conv_embed.conv,layer_norm,layers.{i}.*}, wav2vec_weights, lstm.*,
    linear.*, classifier.*)."""

    def __init__(self, state_dict: dict, hparams: dict, num_classes: int, num_speakers: int,
                 max_set_size, device: torch.device):
        sd = {k: v.detach().float().cpu() for k, v in state_dict.items() if v.dtype.is_floating_point}
        cfg = wav2vec_config(hparams.get("wav2vec") or "WAVLM_BASE")
        lstm = {"hidden_size": 128, "num_layers": 4, "bidirectional": True, "monolithic": True,
                **(hparams.get("lstm") or {})}
        linear = {"hidden_size": 128, "num_layers": 2, **(hparams.get("linear") or {})}
        lstm_h, lstm_dirs, lin_w = lstm_geometry(lstm, linear)
        self.device = device
        self.cfg = cfg
        self._keep: list[torch.Tensor] = []
        self._bias_cache: dict = {}
        up = self._up
        w = ffi.SserWeights()
        shapes = cfg["extractor_conv_layer_config"]
        D, H, F = cfg["encoder_embed_dim"], cfg["encoder_num_heads"], cfg["encoder_ff_interm_features"]
        nl = cfg["encoder_num_layers"]
        if len(shapes) > ffi.PA_W2V_MAX_CONV or nl > ffi.PA_W2V_MAX_LAYERS:
            raise NotImplementedError("at most 8 feature-extractor and 24 transformer layers are built")
        if any(c % 32 for c, _, _ in shapes) or D % 32 or F % 32 or (D // H) % 32 or shapes[0][1] > 16:
            raise NotImplementedError("channel counts, embed_dim, ff size and head size must be multiples of 32")
        w.num_conv = len(shapes)
        w.extractor_layer_norm = int(cfg["extractor_mode"] == "layer_norm")
        fe = "wav2vec.feature_extractor.conv_layers"
        cin = 1
        for i, (cout, k, s_) in enumerate(shapes):
            w.conv_channels[i], w.conv_kernel[i], w.conv_stride[i] = cout, k, s_
            cw = sd[f"{fe}.{i}.conv.weight"]                       # (cout, cin, k)
            packed = cw.reshape(cout, k) if i == 0 else cw.permute(0, 2, 1).reshape(cout, k * cin)
            w.conv_w[i] = up(packed).value
            w.conv_b[i] = up(sd[f"{fe}.{i}.conv.bias"]).value if f"{fe}.{i}.conv.bias" in sd else None
            if f"{fe}.{i}.layer_norm.weight" in sd:
                w.conv_norm_g[i] = up(sd[f"{fe}.{i}.layer_norm.weight"]).value
                w.conv_norm_b[i] = up(sd[f"{fe}.{i}.layer_norm.bias"]).value
            cin = cout
        enc = "wav2vec.encoder"
        w.proj_ln_g, w.proj_ln_b = up(sd[f"{enc}.feature_projection.layer_norm.weight"]), \
            up(sd[f"{enc}.feature_projection.layer_norm.bias"])
        w.proj_w, w.proj_b = up(sd[f"{enc}.feature_projection.projection.weight"]), \
            up(sd[f"{enc}.feature_projection.projection.bias"])
        # positional convolution: weight_norm(dim=2) materialised, w = v * g / ||v||_{dims 0,1}
        pc = f"{enc}.transformer.pos_conv_embed.conv"
        if f"{pc}.parametrizations.weight.original0" in sd:
            g_, v_ = sd[f"{pc}.parametrizations.weight.original0"], sd[f"{pc}.parametrizations.weight.original1"]
        elif f"{pc}.weight_g" in sd:
            g_, v_ = sd[f"{pc}.weight_g"], sd[f"{pc}.weight_v"]
        else:
            g_, v_ = None, sd[f"{pc}.weight"]
        pw = v_ if g_ is None else v_ * (g_ / torch.linalg.vector_norm(v_, dim=(0, 1), keepdim=True))
        groups, KW = cfg["encoder_pos_conv_groups"], cfg["encoder_pos_conv_kernel"]
        CG = D // groups                                           # (D, CG, KW) -> [g][j][ci][co]
        w.pos_w = up(pw.reshape(groups, CG, CG, KW).permute(0, 3, 2, 1))
        w.pos_b = up(sd[f"{pc}.bias"])
        w.pos_kernel, w.pos_groups = KW, groups
        w.enc_ln_g, w.enc_ln_b = up(sd[f"{enc}.transformer.layer_norm.weight"]), \
            up(sd[f"{enc}.transformer.layer_norm.bias"])
        w.embed_dim, w.num_layers, w.num_heads, w.ff_dim = D, nl, H, F
        w.layer_norm_first, w.wavlm = int(bool(cfg["encoder_layer_norm_first"])), int(cfg["wavlm"])
        for i in range(nl):
            lp = f"{enc}.transformer.layers.{i}"
            Lw = w.layers[i]
            if cfg["wavlm"]:
                ipw, ipb = sd[f"{lp}.attention.attention.in_proj_weight"], sd[f"{lp}.attention.attention.in_proj_bias"]
                qk_w, qk_b, v_w, v_b = ipw[:2 * D], ipb[:2 * D], ipw[2 * D:], ipb[2 * D:]
                ow, ob = sd[f"{lp}.attention.attention.out_proj.weight"], sd[f"{lp}.attention.attention.out_proj.bias"]
                Lw.gate_w = up(sd[f"{lp}.attention.gru_rel_pos_linear.weight"])
                Lw.gate_b = up(sd[f"{lp}.attention.gru_rel_pos_linear.bias"])
                Lw.gate_const = up(sd[f"{lp}.attention.gru_rel_pos_const"].reshape(-1))
            else:
                qk_w = torch.cat([sd[f"{lp}.attention.q_proj.weight"], sd[f"{lp}.attention.k_proj.weight"]])
                qk_b = torch.cat([sd[f"{lp}.attention.q_proj.bias"], sd[f"{lp}.attention.k_proj.bias"]])
                v_w, v_b = sd[f"{lp}.attention.v_proj.weight"], sd[f"{lp}.attention.v_proj.bias"]
                ow, ob = sd[f"{lp}.attention.out_proj.weight"], sd[f"{lp}.attention.out_proj.bias"]
            Lw.qk_w, Lw.qk_b, Lw.v_w, Lw.out_w = up(qk_w), up(qk_b), up(v_w), up(ow)
            # soft-max rows sum to 1: P (V + 1 b_v^T) = P V + b_v^T  ->  fold W_o b_v into the output bias
            Lw.out_b = up((ob.double() + ow.double() @ v_b.double()).float())
            Lw.ln1_g, Lw.ln1_b = up(sd[f"{lp}.layer_norm.weight"]), up(sd[f"{lp}.layer_norm.bias"])
            Lw.ff1_w = up(sd[f"{lp}.feed_forward.intermediate_dense.weight"])
            Lw.ff1_b = up(sd[f"{lp}.feed_forward.intermediate_dense.bias"])
            Lw.ff2_w = up(sd[f"{lp}.feed_forward.output_dense.weight"])
            Lw.ff2_b = up(sd[f"{lp}.feed_forward.output_dense.bias"])
            Lw.ln2_g, Lw.ln2_b = up(sd[f"{lp}.final_layer_norm.weight"]), up(sd[f"{lp}.final_layer_norm.bias"])
        self.rel_attn_embed = sd.get(f"{enc}.transformer.layers.0.attention.rel_attn_embed.weight")
        w.use_layer = int(hparams.get("wav2vec_layer", -1))
        if w.use_layer == 0 or w.use_layer > nl:
            # torchaudio's extract_features(num_layers=0) raises the same way when the reference's forward runs
            raise ValueError(f"`num_layers` must be between [1, {nl}]")
        if w.use_layer < 0:
            mix = torch.softmax(sd["wav2vec_weights"], dim=0)      # SSeRiouSS.py:309-311
            for i in range(nl):
                w.layer_mix[i] = float(mix[i])
        w.lstm_layers = int(lstm["num_layers"])
        w.lstm_hidden, w.lstm_bidir = lstm_h, int(lstm_dirs == 2)
        w.num_linear, w.linear_hidden = int(linear["num_layers"]), lin_w
        w.num_classes, w.num_speakers = num_classes, num_speakers
        pack_lstm_head(sd, w, lstm, up)
        self.powerset = bool(max_set_size)
        if self.powerset:
            self.mapping = powerset_mapping(num_speakers, max_set_size)
            assert self.mapping.shape[0] == num_classes
            w.powerset_map = up(self.mapping)
        else:
            assert num_classes == num_speakers
            self.mapping = None
            w.powerset_map = None
        self.struct = w

    def relative_bias(self, num_frames: int):
        """[H][T][T] table rel_attn_embed[bucket(k - q)] of a chunk with T frames (compute_bias of the first
        layer, shared by all layers); None for a wav2vec 2.0 encoder"""
        if self.rel_attn_embed is None:
            return None
        if num_frames not in self._bias_cache:
            pos = torch.arange(num_frames, dtype=torch.long)
            bucket = relative_position_bucket(pos[None, :] - pos[:, None], self.cfg["encoder_num_buckets"],
                                              self.cfg["encoder_max_distance"])
            table = self.rel_attn_embed[bucket].permute(2, 0, 1).contiguous()
            self._bias_cache[num_frames] = table.to(self.device)
        return self._bias_cache[num_frames]

    def _up(self, t: torch.Tensor):
        d = t.contiguous().to(self.device)
        self._keep.append(d)
        return C.c_void_p(d.data_ptr())


class XVectorPack:
    """Device-resident, kernel-ready XVectorSincNet weights + the `pa_xvec_weights` struct
    (models/embedding/xvector.py:205-252).  State-dict layout: sincnet.*, tdnns.{3l}.{weight,bias} (Conv1d),
    tdnns.{3l+2}.{weight,bias,running_mean,running_var} (BatchNorm1d, eval), embedding.{weight,bias}.
    Every BatchNorm follows a LeakyReLU, so it cannot be folded backwards; being an affine map it is folded
    FORWARD, in float64, into the next convolution (W_j diag(s), b + sum_j W_j t); the last one is handed to
    the pooling kernel, which applies it on load (it cannot move past the pooling: an all-zero mask pools
    to mean = std = 0, not to the BatchNorm's shift)."""

    KERNEL, DILATION = (5, 3, 3, 1, 1), (1, 2, 3, 1, 1)

    def __init__(self, state_dict: dict, hparams: dict, device: torch.device):
        sd = {k: v.detach().float().cpu() for k, v in state_dict.items() if v.dtype.is_floating_point}
        if int((hparams.get("sincnet") or {}).get("stride", 10)) != 10:
            raise NotImplementedError("kernels are built for SincNet stride 10")
        self.device = device
        self._keep: list[torch.Tensor] = []
        w = ffi.XvecWeights()
        self.sinc_taps = pack_sincnet(sd, w, self._up)
        self.folded_tdnn: list = []                # [(taps (k, cout, cin_pad), bias (cout))] kept for tests
        scale = shift = None                       # affine map of the previous BatchNorm (float64)
        for l in range(ffi.PA_XVEC_TDNN):
            cw = sd[f"tdnns.{3 * l}.weight"].double()          # (cout, cin, k)
            cb = sd[f"tdnns.{3 * l}.bias"].double()
            cout, cin, k = cw.shape
            if k != self.KERNEL[l] or cout % 4:
                raise NotImplementedError(f"unexpected TDNN layer {l}: {tuple(cw.shape)}")
            if scale is not None:
                cb = cb + torch.einsum("oik,i->o", cw, shift)
                cw = cw * scale.view(1, -1, 1)
            cin_pad = 64 if l == 0 else cin
            taps = torch.zeros(k, cout, cin_pad, dtype=torch.float64)
            taps[:, :, :cin] = cw.permute(2, 0, 1)
            w.tdnn_w[l] = self._up(taps.float()).value
            w.tdnn_b[l] = self._up(cb.float()).value
            self.folded_tdnn.append((self._keep[-2], self._keep[-1]))
            w.tdnn_channels[l], w.tdnn_kernel[l], w.tdnn_dilation[l] = cout, k, self.DILATION[l]
            bn = f"tdnns.{3 * l + 2}"
            scale = sd[bn + ".weight"].double() / torch.sqrt(sd[bn + ".running_var"].double() + 1e-5)
            shift = sd[bn + ".bias"].double() - sd[bn + ".running_mean"].double() * scale
        ew, eb = sd["embedding.weight"].double(), sd["embedding.bias"].double()
        C_ = int(w.tdnn_channels[ffi.PA_XVEC_TDNN - 1])
        assert ew.shape[1] == 2 * C_
        ld = (2 * C_ + 31) // 32 * 32
        packed = torch.zeros(ew.shape[0], ld, dtype=torch.float64)
        packed[:, :2 * C_] = ew
        w.bn_scale, w.bn_shift = self._up(scale.float()), self._up(shift.float())
        self.last_batchnorm = (self._keep[-2], self._keep[-1])
        w.emb_w, w.emb_b = self._up(packed.float()), self._up(eb.float())
        self.embedding = (self._keep[-2], self._keep[-1])
        w.dimension = int(ew.shape[0])
        self.struct = w

    def _up(self, t: torch.Tensor):
        d = t.contiguous().to(self.device)
        self._keep.append(d)
        return C.c_void_p(d.data_ptr())


# ---------------------------------------------------------------------------------------------
# WeSpeaker ResNet34 (models/embedding/wespeaker/)
# ---------------------------------------------------------------------------------------------
def kaldi_mel_banks(num_bins: int = 80, padded: int = 512, sample_freq: float = 16000.0,
                    low_freq: float = 20.0, high_freq: float = 0.0) -> torch.Tensor:
    """torchaudio.compliance.kaldi.get_mel_banks (vtln_warp = 1), third party: restated from the
    published algorithm -> (num_bins, padded // 2 + 1) fp32 with the right-most column zero
    (call site: wespeaker/__init__.py:88-99)."""
    import math
    num_fft_bins = padded / 2
    nyquist = 0.5 * sample_freq
    if high_freq <= 0.0:
        high_freq += nyquist
    fft_bin_width = sample_freq / padded
    mel_low = 1127.0 * math.log(1.0 + low_freq / 700.0)
    mel_high = 1127.0 * math.log(1.0 + high_freq / 700.0)
    delta = (mel_high - mel_low) / (num_bins + 1)
    b = torch.arange(num_bins).unsqueeze(1)
    left, center, right = mel_low + b * delta, mel_low + (b + 1.0) * delta, mel_low + (b + 2.0) * delta
    mel = (1127.0 * (1.0 + (fft_bin_width * torch.arange(num_fft_bins)) / 700.0).log()).unsqueeze(0)
    up = (mel - left) / (center - left)
    down = (right - mel) / (right - center)
    bins = torch.max(torch.zeros(1), torch.min(up, down))
    return torch.nn.functional.pad(bins, (0, 1)).contiguous()


def winograd_weights(cw: torch.Tensor) -> torch.Tensor:
    """(cout, cin, 3, 3) -> [16][cout][cin] float32: U = G g G^T of Winograd F(2x2, 3x3), computed in
    float64 (xi = 4a + b indexes the 4x4 transform domain)."""
    G = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]],
                     dtype=torch.float64)
    U = torch.einsum("ap,oipq,bq->aboi", G, cw.double(), G)
    return U.reshape(16, cw.shape[0], cw.shape[1]).float().contiguous()


def winograd_pack(U: torch.Tensor) -> torch.Tensor:
    """[16][cout][cin] -> the kernel's staging image [cout/32][cin/16][512 rows][4 slots][4]: one
    CONTIGUOUS 32-KB slab per (32-cout slice, 16-cin stage) so that its LDS-DMA is a linear full-line
    stream (the strided [16][cout][cin] form reads 64-byte pieces 4*cin bytes apart: at cin = 256 they
    fall on a quarter of the L2 channels).  Row r = 32 xi + n holds channel quad g at physical slot
    (g + 2 ((r >> 2) & 1)) & 3 -- the bank-conflict-free LDS layout of csrc/emb_winograd.hip, applied on
    the host so that the DMA needs no per-lane address arithmetic."""
    _, cout, cin = U.shape
    assert cout % 32 == 0 and cin % 16 == 0
    A = U.reshape(16, cout // 32, 32, cin // 16, 4, 4).permute(1, 3, 0, 2, 4, 5)
    A = A.reshape(cout // 32, cin // 16, 512, 4, 4)
    r = torch.arange(512)
    quad_of_slot = (torch.arange(4)[None, :] - 2 * ((r >> 2) & 1)[:, None]) & 3      # (512, 4)
    idx = quad_of_slot.view(1, 1, 512, 4, 1).expand(A.shape[0], A.shape[1], 512, 4, 4)
    return torch.gather(A, 3, idx).contiguous()


#: layers that use F(4x4,3x3) by default, from measurements on MI355X (profiles/r4_wino4_anatomy.txt; one audio-hour
#: per step: F(2x2) everywhere 883 ms, "3" 859, "34" 852, "234" 848; the 32-channel layer 1 is faster with F(2x2):
#: its tiles have only four 8-channel stages to amortise the F(4x4) epilogue, and it is close to HBM-bound)
WINOGRAD4_DEFAULT_LAYERS = "234"

#: G of Winograd F(4x4, 3x3) (Lavin & Gray 2016, interpolation points 0, +-1, +-2, inf)
WINOGRAD4_G = [[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
               [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]]


def winograd4_weights(cw: torch.Tensor) -> torch.Tensor:
    """(cout, cin, 3, 3) -> [36][cout][cin] float32: U = G g G^T of Winograd F(4x4, 3x3), computed in float64
    (xi = 6a + b indexes the 6x6 transform domain)."""
    G = torch.tensor(WINOGRAD4_G, dtype=torch.float64)
    U = torch.einsum("ap,oipq,bq->aboi", G, cw.double(), G)
    return U.reshape(36, cw.shape[0], cw.shape[1]).float().contiguous()


def winograd4_pack(U: torch.Tensor) -> torch.Tensor:
    """[36][cout][cin] -> the staging image of csrc/emb_winograd4.hip, [cout/32][cin/8][row = 32 xi + n][8]: one
    CONTIGUOUS 36-KB slab per (32-cout slice, 8-cin stage), so that its LDS-DMA is a linear stream and lane
    (n & 15, g) of the MFMA's A operand reads the input-channel pair g of output channel n at byte
    32 row + 8 (g ^ 2 s), s = bit 3 of n: rows 8..15 and 24..31 of a slice hold their two channel quads swapped, which
    makes the 32 lanes of a ds_read_b64 lane group hit 64 different LDS banks (csrc/emb_winograd4_geom.h)."""
    _, cout, cin = U.shape
    assert cout % 32 == 0 and cin % 8 == 0
    A = U.reshape(36, cout // 32, 32, cin // 8, 8).permute(1, 3, 0, 2, 4).contiguous()   # [slice][stage][xi][n][8]
    swapped = torch.cat([A[..., 4:], A[..., :4]], dim=-1)
    rows = ((torch.arange(32) >> 3) & 1).bool().view(1, 1, 1, 32, 1)
    A = torch.where(rows, swapped, A)
    return A.reshape(cout // 32, cin // 8, 36 * 32, 8).contiguous()


def _fold_bn(sd: dict, prefix: str, eps: float = 1e-5):
    scale = sd[prefix + ".weight"] / torch.sqrt(sd[prefix + ".running_var"] + eps)
    shift = sd[prefix + ".bias"] - sd[prefix + ".running_mean"] * scale
    return scale, shift


#: Winograd guard measurements of this process, by (digest of the ResNet weights, F(4x4) layers, fbank centring): see
#: `EmbeddingPack._guard_winograd`
_GUARD_MEASUREMENTS: dict = {}


class EmbeddingPack:
    """Device-resident, kernel-ready WeSpeaker ResNet weights + the `pa_emb_weights` struct.
    State-dict layout: resnet.conv1/bn1, resnet.layer{1..4}.{i}.{conv1,bn1,conv2,bn2,shortcut.0,
    shortcut.1}, resnet.seg_1."""

    def __init__(self, state_dict: dict, device: torch.device, num_blocks=None,
                 num_mel: int = 80, sample_rate: int = 16000, winograd: Optional[bool] = None,
                 guard: Optional[bool] = None, center_kernel: int = 0):
        if winograd is None:
            winograd = os.environ.get("PA_WINOGRAD", "1") != "0"
        self.winograd = winograd
        # ResNet layers (1 .. 4) whose stride-1 3x3 convolutions run through Winograd F(4x4,3x3) instead of F(2x2,3x3)
        # (csrc/emb_winograd4.hip; BasicBlock networks): PA_WINOGRAD4="" switches it off, "34" = layers 3 and 4
        self.winograd4_layers = {int(c) for c in os.environ.get("PA_WINOGRAD4", WINOGRAD4_DEFAULT_LAYERS)
                                 if c in "1234"} if winograd else set()
        sd = {k: v.detach().float().cpu() for k, v in state_dict.items() if v.dtype.is_floating_point}
        self.device = device
        self._keep: list[torch.Tensor] = []
        digest = hashlib.sha1()
        for k in sorted(sd):
            if k.startswith("resnet."):
                digest.update(k.encode())
                digest.update(sd[k].contiguous().numpy().tobytes())
        self._weights_digest = digest.hexdigest()
        w = ffi.EmbWeights()
        w.num_mel, w.num_layers = num_mel, 4
        # 0: global mean subtraction; odd K: running mean of K frames (model.WeSpeakerResNet34.fbank_center_kernel)
        if center_kernel < 0 or (center_kernel and center_kernel % 2 == 0):
            raise ValueError(f"center_kernel must be 0 or odd, got {center_kernel}")
        w.fb_center_kernel = int(center_kernel)
        # architecture from the keys: Bottleneck blocks have a conv3 (resnet.py:148-212); block counts =
        # highest block index per layer (ResNet34 3,4,6,3 / 152: 3,8,36,3 / 221: 6,16,48,3 / 293: 10,20,64,3)
        self.bottleneck = "resnet.layer1.0.conv3.weight" in sd
        if num_blocks is None:
            num_blocks = [1 + max(int(k.split(".")[2]) for k in sd if k.startswith(f"resnet.layer{l + 1}."))
                          for l in range(4)]
        self.num_blocks = tuple(int(n) for n in num_blocks)
        if sum(self.num_blocks) > ffi.PA_MAX_RES_BLOCKS:
            raise NotImplementedError(f"{sum(self.num_blocks)} residual blocks > {ffi.PA_MAX_RES_BLOCKS}")
        w.bottleneck = int(self.bottleneck)
        planes = [sd[f"resnet.layer{l + 1}.0.conv1.weight"].shape[0] for l in range(4)]
        if planes[0] != 32:
            raise NotImplementedError("stem kernel is built for m_channels = 32")
        for l in range(4):
            w.num_blocks[l] = int(num_blocks[l])
            w.planes[l] = int(planes[l])
        w.embed_dim = int(sd["resnet.seg_1.weight"].shape[0])
        # fbank tables (fp64 -> fp32)
        w.fb_window = self._up(torch.hamming_window(400, periodic=False, alpha=0.54, beta=0.46))
        m = np.arange(256, dtype=np.float64)
        tw256 = np.stack([np.cos(2 * np.pi * m / 256), -np.sin(2 * np.pi * m / 256)], -1)
        k = np.arange(257, dtype=np.float64)
        tw512 = np.stack([np.cos(2 * np.pi * k / 512), -np.sin(2 * np.pi * k / 512)], -1)
        w.fb_tw256 = self._up(torch.from_numpy(tw256.astype(np.float32)))
        w.fb_tw512 = self._up(torch.from_numpy(tw512.astype(np.float32)))
        mel = kaldi_mel_banks(num_mel, 512, float(sample_rate))
        self.mel = mel
        nz = mel > 0
        lo = torch.tensor([int(torch.nonzero(r)[0]) if r.any() else 0 for r in nz], dtype=torch.int32)
        hi = torch.tensor([int(torch.nonzero(r)[-1]) if r.any() else -1 for r in nz], dtype=torch.int32)
        w.fb_mel_w, w.fb_mel_lo, w.fb_mel_hi = self._up(mel), self._up(lo), self._up(hi)
        # stem
        sc, sh = _fold_bn(sd, "resnet.bn1")
        cw = sd["resnet.conv1.weight"] * sc.view(-1, 1, 1, 1)  # (32,1,3,3)
        w.stem_w = self._up(cw[:, 0].permute(1, 2, 0).reshape(9, 32))
        w.stem_shift = self._up(sh)
        blk = 0
        for l in range(4):
            for i in range(num_blocks[l]):
                pre = f"resnet.layer{l + 1}.{i}"
                first_stride = 2 if (i == 0 and l > 0) else 1
                if self.bottleneck:
                    # 1x1 -> 3x3 (stride) -> 1x1, BatchNorm folded into each (resnet.py:148-212)
                    sc, sh = _fold_bn(sd, f"{pre}.bn1")
                    w.blk_w1[blk] = self._up(sd[f"{pre}.conv1.weight"][:, :, 0, 0] * sc.view(-1, 1)).value
                    w.blk_shift1[blk] = self._up(sh).value
                    sc, sh = _fold_bn(sd, f"{pre}.bn2")
                    cw = sd[f"{pre}.conv2.weight"] * sc.view(-1, 1, 1, 1)
                    w.blk_w2[blk] = self._up(cw.permute(2, 3, 0, 1).reshape(9, cw.shape[0], cw.shape[1])).value
                    w.blk_shift2[blk] = self._up(sh).value
                    if winograd and first_stride == 1:
                        w.blk_u2[blk] = self._up(winograd_pack(winograd_weights(cw))).value
                    sc, sh = _fold_bn(sd, f"{pre}.bn3")
                    w.blk_w3[blk] = self._up(sd[f"{pre}.conv3.weight"][:, :, 0, 0] * sc.view(-1, 1)).value
                    w.blk_shift3[blk] = self._up(sh).value
                else:
                    for j, (cn, bn) in enumerate((("conv1", "bn1"), ("conv2", "bn2")), 1):
                        sc, sh = _fold_bn(sd, f"{pre}.{bn}")
                        cw = sd[f"{pre}.{cn}.weight"] * sc.view(-1, 1, 1, 1)  # (cout,cin,3,3)
                        img = cw.permute(2, 3, 0, 1).reshape(9, cw.shape[0], cw.shape[1])
                        getattr(w, f"blk_w{j}")[blk] = self._up(img).value
                        getattr(w, f"blk_shift{j}")[blk] = self._up(sh).value
                        stride = first_stride if j == 1 else 1
                        if winograd and stride == 1:
                            getattr(w, f"blk_u{j}")[blk] = self._up(winograd_pack(winograd_weights(cw))).value
                            if l + 1 in self.winograd4_layers and cw.shape[1] >= 32:
                                getattr(w, f"blk_v{j}")[blk] = self._up(winograd4_pack(winograd4_weights(cw))).value
                if f"{pre}.shortcut.0.weight" in sd:
                    sc, sh = _fold_bn(sd, f"{pre}.shortcut.1")
                    cw = sd[f"{pre}.shortcut.0.weight"][:, :, 0, 0] * sc.view(-1, 1)
                    w.blk_wsc[blk] = self._up(cw).value
                    w.blk_shiftsc[blk] = self._up(sh).value
                blk += 1
        w.seg1_w = self._up(sd["resnet.seg_1.weight"])
        w.seg1_b = self._up(sd["resnet.seg_1.bias"])
        self.struct = w
        self.winograd_guard: list[dict] = []
        if guard is None:
            guard = os.environ.get("PA_WINOGRAD_GUARD", "1") != "0"
        if guard and winograd and not self.bottleneck and device.type == "cuda":
            self._guard_winograd()

    #: the guard's margins on  max |Winograd - direct| / max |direct|  over a convolution's output map (after shift,
    #: residual and ReLU) for the calibration chunks.  On the seeded / BatchNorm-randomised ResNet34 of the test suite
    #: the statistic is 0.3 - 2.5e-6 for F(4x4) and 0.6 - 8e-7 for F(2x2) (the direct fp32 kernel itself is 0.5 - 5e-7
    #: away from a float64 evaluation); the margins leave a factor ~6 / ~12 above that.  A demotion costs speed only.
    WINOGRAD_GUARD_MARGINS = {"f4": 1.5e-5, "f2": 1.0e-5}

    @staticmethod
    @functools.lru_cache(maxsize=2)
    def calibration_chunks(num_samples: int = 48000) -> torch.Tensor:
        """the guard's fixed input: two seeded, speech-like 3-s chunks (glottal-pulse-like harmonic series at 120 /
        210 Hz under a 4-Hz syllable envelope + noise floor), (2, num_samples) float32 at 16 kHz, RMS ~ 0.1"""
        g = torch.Generator().manual_seed(20250923)
        t = torch.arange(num_samples, dtype=torch.float64) / 16000.0
        out = []
        for f0 in (120.0, 210.0):
            k = torch.arange(1, int(7000 // f0), dtype=torch.float64)
            phase = 2 * math.pi * torch.rand(len(k), generator=g, dtype=torch.float64)
            tilt = 1.0 / k * (1.0 + 0.8 * torch.cos(2 * math.pi * k * f0 / 2300.0))      # a coarse formant ripple
            glide = f0 * (1.0 + 0.08 * torch.sin(2 * math.pi * 0.7 * t))
            inst = 2 * math.pi * torch.cumsum(glide, 0) / 16000.0
            x = (tilt[:, None] * torch.sin(k[:, None] * inst[None, :] + phase[:, None])).sum(0)
            env = 0.55 + 0.45 * torch.sin(2 * math.pi * 4.0 * t + phase[0])
            x = x * env + 0.05 * torch.randn(num_samples, generator=g, dtype=torch.float64)
            out.append(0.1 * x / x.pow(2).mean().sqrt())
        return torch.stack(out).float()

    def _guard_winograd(self):
        """Numerical guard of the Winograd paths: with the LOADED weights, every stride-1
        3x3 convolution is evaluated on the activations the calibration chunks produce in this very network -- by the
        direct kernel and by each Winograd image -- and an image whose result strays from the direct one by more
        than its margin is dropped: F(4x4) -> F(2x2) -> direct, per convolution.  F(4x4,3x3) multiplies by 8 and 1/24
        in its transforms: harmless for weights and activations of ordinary spread (BatchNorm keeps them there),
        but a checkpoint whose convolution cancels a large common component of its input (output << sum |w||x|)
        loses those digits first.  Decisions are kept in `self.winograd_guard` and logged."""
        import logging
        lib = ffi.load()
        w = self.struct
        log = logging.getLogger("pyannote_audio_amd")
        m4 = self._guard_margin("PA_WINOGRAD_GUARD_F4", "f4")
        m2 = self._guard_margin("PA_WINOGRAD_GUARD_F2", "f2")
        # the measurement depends on the weights only (the calibration chunks are fixed): a second pack of the same
        # checkpoint in this process -- another pipeline, another device -- reuses the numbers instead of running the
        # calibration forward again (three kernels per convolution)
        key = (self._weights_digest, tuple(sorted(self.winograd4_layers)), int(w.fb_center_kernel))
        cached = _GUARD_MEASUREMENTS.get(key)
        if cached is not None:
            rep = cached
        else:
            rep = self._measure_winograd(lib, w)
            _GUARD_MEASUREMENTS[key] = rep
        self._apply_guard(rep, m4, m2, log)

    @classmethod
    def _guard_margin(cls, env: str, which: str) -> float:
        raw = os.environ.get(env)
        if raw is None:
            return cls.WINOGRAD_GUARD_MARGINS[which]
        try:
            value = float(raw)
        except ValueError:
            raise ValueError(f"{env}={raw!r}: the guard margin must be a number") from None
        if not (value > 0.0 and math.isfinite(value)):
            raise ValueError(f"{env}={raw!r}: the guard margin must be positive and finite "
                             "(PA_WINOGRAD_GUARD=0 switches the guard off)")
        return value

    def _measure_winograd(self, lib, w):
        with torch.cuda.device(self.device):
            wav = self.calibration_chunks().to(self.device)
            B, N = wav.shape
            report = torch.zeros(8 * ffi.PA_MAX_RES_BLOCKS, dtype=torch.float32, device=self.device)
            emb = torch.empty((B, int(w.embed_dim)), dtype=torch.float32, device=self.device)
            ws = torch.empty(lib.pa_emb_calibrate_workspace_bytes(w, B, N), dtype=torch.uint8, device=self.device)
            ffi.check(lib.pa_emb_calibrate_winograd(w, ffi.ptr(wav.view(-1)), B * N, N, B, N, ffi.ptr(report),
                                                    ffi.ptr(emb), ffi.ptr(ws), ws.numel(), ffi.stream()),
                      "pa_emb_calibrate_winograd")
            return report.cpu().view(-1, 2, 4).numpy()

    def _apply_guard(self, rep, m4: float, m2: float, log):
        w = self.struct
        blk = 0
        for l in range(4):
            for i in range(self.num_blocks[l]):
                for j in (0, 1):
                    ref4, d4, ref2, d2 = (float(v) for v in rep[blk, j])
                    if ref4 == 0.0 and ref2 == 0.0:
                        continue          # a strided convolution, or no Winograd image: nothing to guard
                    entry = {"layer": l + 1, "block": i, "conv": j + 1,
                             "f4": d4 / ref4 if ref4 > 0 else None, "f2": d2 / ref2 if ref2 > 0 else None}
                    ok4 = ref4 > 0 and entry["f4"] <= m4          # (NaN compares false)
                    ok2 = ref2 > 0 and entry["f2"] <= m2
                    if ref4 > 0 and not ok4:
                        getattr(w, f"blk_v{j + 1}")[blk] = None
                    if ref2 > 0 and not ok2:    # (also F(4x4)'s stand-in on maps where F(4x4) does not pay)
                        getattr(w, f"blk_u{j + 1}")[blk] = None
                    path = "f4" if ok4 else ("f2" if ok2 else "direct")
                    entry["path"] = path
                    self.winograd_guard.append(entry)
                    if path != ("f4" if ref4 > 0 else "f2"):
                        log.warning("layer%d.%d.conv%d: Winograd F(4x4) error %s, F(2x2) error %s of the output "
                                    "maximum (margins %.1e / %.1e) -> %s kernel", l + 1, i, j + 1,
                                    "%.2e" % entry["f4"] if entry["f4"] is not None else "n/a",
                                    "%.2e" % entry["f2"] if entry["f2"] is not None else "n/a", m4, m2, path)
                blk += 1

    def _up(self, t: torch.Tensor):
        d = t.contiguous().to(self.device)
        self._keep.append(d)
        return C.c_void_p(d.data_ptr())
