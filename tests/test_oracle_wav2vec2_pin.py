"""Pin the oracle's restatement of torchaudio's wav2vec 2.0 / WavLM encoder (oracle/wav2vec2.py; reference call
site models/segmentation/SSeRiouSS.py:98-124, 289-328) against an INDEPENDENT implementation of the same published
architectures that is installed in this image: HuggingFace `transformers` (`Wav2Vec2Model`, `WavLMModel`),
constructed offline with random weights.  torchaudio itself is not installed.

How the two are tied together: the HF state dict is renamed key by key into torchaudio's layout -- the mapping
torchaudio publishes as `torchaudio.models.wav2vec2.utils.import_huggingface_model` (feature extractor names
unchanged, `feature_projection.*` and `encoder.*` moved under `encoder.feature_projection` /
`encoder.transformer`; for WavLM the q/k/v projections concatenated into `attention.attention.in_proj_*`) -- and
loaded into the ORACLE with `strict=True`: every oracle parameter must receive an HF tensor of the same shape.
torchaudio's own integration tests require `imported.encoder.transformer(x) == original.encoder(x).last_hidden_state`
for exactly this mapping, so HF is a faithful stand-in for what `Wav2Vec2Model.extract_features` must return:

    oracle.extract_features(x)[0][i]  ==  HF hidden_states[i + 1]       (the output of encoder layer i)

with ONE documented difference: for pre-LN ("stable layer norm", torchaudio `encoder_layer_norm_first=True`) models
HF reports the LAST hidden state after the encoder's final LayerNorm, torchaudio's `get_intermediate_outputs`
before it -- the test applies that LayerNorm to the oracle's last output before comparing.

Tolerance: both sides are float32 torch on the CPU; they differ in the association of the attention scaling
(q * d^-1/2 before vs inside the product) and of the fused q/k/v projection: 2e-5 absolute on LayerNorm-scaled
activations (|x| ~ 1), 1e-4 relative to the peak on the un-normalised pre-LN stream.
"""
import numpy as np
import pytest
import torch

from oracle import wav2vec2 as ow

transformers = pytest.importorskip("transformers")


def _rename_common(hf_state: dict) -> dict:
    out = {}
    for k, v in hf_state.items():
        if k == "masked_spec_embed" or k.startswith("adapter"):
            continue
        if k.startswith("feature_extractor."):
            out[k] = v
        elif k.startswith("feature_projection."):
            out["encoder." + k] = v
        elif k.startswith("encoder."):
            out["encoder.transformer." + k[len("encoder."):]] = v
        else:
            raise AssertionError(f"unmapped HF key {k}")
    return out


def _rename_wavlm(state: dict, num_layers: int) -> dict:
    """q/k/v projections -> nn.MultiheadAttention's packed in_proj (torchaudio's `transform_wavlm_encoder_state`)"""
    out = dict(state)
    for i in range(num_layers):
        p = f"encoder.transformer.layers.{i}.attention."
        for kind in ("weight", "bias"):
            out[p + f"attention.in_proj_{kind}"] = torch.cat([out.pop(p + f"{n}_proj.{kind}") for n in "qkv"], dim=0)
            out[p + f"attention.out_proj.{kind}"] = out.pop(p + f"out_proj.{kind}")
    return out


def _randomise(model: torch.nn.Module, seed: int):
    """HF initialises LayerNorms to (1, 0), biases to 0 and the WavLM gate constant to 1: draw everything at random
    so that a swapped pair of parameters or a dropped bias cannot cancel out"""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.dim() == 1 and ("layer_norm" in name or "norm" in name) and name.endswith("weight"):
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
            elif p.dim() == 1 or name.endswith("gru_rel_pos_const"):
                p.copy_((1.0 if name.endswith("gru_rel_pos_const") else 0.0) + 0.1 * torch.randn(p.shape, generator=g))
            elif "parametrizations" in name or "rel_attn_embed" in name:
                p.copy_(p + 0.3 * torch.randn(p.shape, generator=g))


CONV = dict(conv_dim=(32, 32, 48), conv_stride=(5, 2, 2), conv_kernel=(10, 3, 2))
ENC = dict(hidden_size=64, num_hidden_layers=3, num_attention_heads=4, intermediate_size=96,
           num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4, hidden_dropout=0.0, attention_dropout=0.0,
           activation_dropout=0.0, feat_proj_dropout=0.0, layerdrop=0.0, apply_spec_augment=False)


def _oracle_config(hf_cfg, wavlm: bool) -> dict:
    cfg = dict(
        extractor_mode="group_norm" if hf_cfg.feat_extract_norm == "group" else "layer_norm",
        extractor_conv_layer_config=list(zip(hf_cfg.conv_dim, hf_cfg.conv_kernel, hf_cfg.conv_stride)),
        extractor_conv_bias=hf_cfg.conv_bias, encoder_embed_dim=hf_cfg.hidden_size,
        encoder_pos_conv_kernel=hf_cfg.num_conv_pos_embeddings,
        encoder_pos_conv_groups=hf_cfg.num_conv_pos_embedding_groups, encoder_num_layers=hf_cfg.num_hidden_layers,
        encoder_num_heads=hf_cfg.num_attention_heads, encoder_ff_interm_features=hf_cfg.intermediate_size,
        encoder_layer_norm_first=hf_cfg.do_stable_layer_norm)
    if wavlm:
        cfg.update(encoder_num_buckets=hf_cfg.num_buckets, encoder_max_distance=hf_cfg.max_bucket_distance)
    return cfg


def _build(kind: str, stable: bool, seed: int):
    norm = dict(feat_extract_norm="layer" if stable else "group", conv_bias=stable, do_stable_layer_norm=stable)
    if kind == "wavlm":
        cfg = transformers.WavLMConfig(**CONV, **ENC, **norm, num_buckets=32, max_bucket_distance=80)
        hf = transformers.WavLMModel(cfg)
    else:
        cfg = transformers.Wav2Vec2Config(**CONV, **ENC, **norm)
        hf = transformers.Wav2Vec2Model(cfg)
    hf.eval()
    _randomise(hf, seed)
    state = _rename_common(hf.state_dict())
    if kind == "wavlm":
        state = _rename_wavlm(state, cfg.num_hidden_layers)
    oracle = (ow.wavlm_model if kind == "wavlm" else ow.wav2vec2_model)(**_oracle_config(cfg, kind == "wavlm"))
    oracle.load_state_dict(state, strict=True)      # every oracle parameter is fed, nothing of HF is left over
    return hf, oracle.eval(), cfg


@pytest.mark.parametrize("kind,stable", [("wav2vec2", False), ("wav2vec2", True), ("wavlm", False), ("wavlm", True)])
def test_extract_features_matches_huggingface(kind, stable):
    hf, oracle, cfg = _build(kind, stable, seed=11 + 2 * stable + (kind == "wavlm"))
    g = torch.Generator().manual_seed(5)
    wav = 0.3 * torch.randn(2, 4000, generator=g)
    with torch.inference_mode():
        ref = hf(wav, output_hidden_states=True).hidden_states
        got, _ = oracle.extract_features(wav)
        assert len(got) == cfg.num_hidden_layers and len(ref) == cfg.num_hidden_layers + 1
        if stable:   # HF's last entry is after the encoder's final LayerNorm (module docstring)
            got = list(got[:-1]) + [oracle.encoder.transformer.layer_norm(got[-1])]
        for i, (a, b) in enumerate(zip(got, ref[1:])):
            assert a.shape == b.shape
            err = (a - b).abs().max().item()
            bound = 2e-5 + 1e-4 * b.abs().max().item() if stable else 2e-5 * max(1.0, b.abs().max().item())
            assert err <= bound, f"{kind} stable={stable}: layer {i} differs by {err:.3e} (bound {bound:.1e})"
        # a truncated request returns the same prefix (SSeRiouSS.py:289-296 passes num_layers)
        part, _ = oracle.extract_features(wav, num_layers=2)
        assert len(part) == 2 and torch.equal(part[1], oracle.extract_features(wav)[0][1])


@pytest.mark.parametrize("kind", ["wav2vec2", "wavlm"])
def test_feature_extractor_and_projection_match_huggingface(kind):
    """the stages in front of the transformer one by one (the way torchaudio's own HF integration test walks them)"""
    hf, oracle, _ = _build(kind, False, seed=3)
    g = torch.Generator().manual_seed(6)
    wav = 0.3 * torch.randn(2, 3000, generator=g)
    with torch.inference_mode():
        ref = hf.feature_extractor(wav).transpose(1, 2)
        got = oracle.feature_extractor(wav)
        assert got.shape == ref.shape and (got - ref).abs().max().item() <= 1e-5
        ref_p = hf.feature_projection(ref)[0]
        got_p = oracle.encoder.feature_projection(got)
        assert (got_p - ref_p).abs().max().item() <= 2e-5
        ref_c = hf.encoder.pos_conv_embed(ref_p)
        got_c = oracle.encoder.transformer.pos_conv_embed(got_p)
        assert (got_c - ref_c).abs().max().item() <= 2e-5


def test_relative_position_buckets_match_huggingface():
    """WavLM's bucketing of k - q (shared by the oracle, the product's weights.relative_position_bucket and HF)"""
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from pyannote_audio_amd.weights import relative_position_bucket
    hf, oracle, cfg = _build("wavlm", False, seed=9)
    att_hf = hf.encoder.layers[0].attention
    att_or = oracle.encoder.transformer.layers[0].attention
    rel = torch.arange(-300, 301)[None, :] - torch.zeros(1, 1, dtype=torch.long)
    want = att_hf._relative_positions_bucket(rel)
    assert torch.equal(att_or.relative_position_bucket(rel), want)
    assert torch.equal(relative_position_bucket(rel, cfg.num_buckets, cfg.max_bucket_distance), want)
    with torch.inference_mode():
        assert np.allclose(att_or.compute_bias(40, 40).numpy(), att_hf.compute_bias(40, 40).numpy(), atol=0)
