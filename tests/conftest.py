import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import pyannote_audio_amd.ffi as ffi
    ffi.require_gpu()  # loud failure if the HIP library is missing on a GPU box
    return torch.device("cuda:0")


def report(name, got, want, log=None):
    """max-abs / max-rel error summary, appended to gpurun_out/parity.log"""
    import torch
    got = got.detach().float().cpu()
    want = want.detach().float().cpu()
    diff = (got - want).abs()
    denom = want.abs().clamp_min(1e-6)
    line = (f"{name}: shape={tuple(got.shape)} max_abs={diff.max().item():.3e} "
            f"max_rel={(diff / denom).max().item():.3e} ref_absmax={want.abs().max().item():.3e} "
            f"nan={int(torch.isnan(got).sum())}")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "parity.log"), "a") as fp:
        fp.write(line + "\n")
    print(line)
    return diff.max().item()


def north_star_ratio(name, got, want, rtol=1e-4, atol=1e-5):
    """BASELINE.json north_star / SURVEY.md section 8d float tolerance for segmentation log-probs and
    embeddings: |got - want| <= atol + rtol * |want| element-wise (rtol 1e-4, atol 1e-5).  Returns
    max |d| / (atol + rtol |want|) (<= 1 passes) and logs it next to the max-abs/max-rel summary."""
    import torch
    got = torch.as_tensor(got).detach().double().cpu()
    want = torch.as_tensor(want).detach().double().cpu()
    report(name, got, want)
    ratio = ((got - want).abs() / (atol + rtol * want.abs())).max().item()
    with open(os.path.join(ROOT, "gpurun_out", "parity.log"), "a") as fp:
        fp.write(f"{name}: north-star ratio (rtol={rtol:g}, atol={atol:g}) = {ratio:.3f}\n")
    print(f"{name}: north-star ratio = {ratio:.3f}")
    return ratio


PYANNET_HPARAMS = {
    "sincnet": {"stride": 10, "sample_rate": 16000},
    "lstm": {"hidden_size": 128, "num_layers": 4, "bidirectional": True, "monolithic": True,
             "dropout": 0.0, "batch_first": True},
    "linear": {"hidden_size": 128, "num_layers": 2},
    "sample_rate": 16000, "num_channels": 1,
}
WESPEAKER_HPARAMS = {"sample_rate": 16000, "num_channels": 1, "num_mel_bins": 80, "frame_length": 25,
                     "frame_shift": 10, "dither": 0.0, "window_type": "hamming", "use_energy": False}


def write_pipeline_dir(root, seg_model, emb_model, config_extra=None, powerset=True):
    """synthetic `speaker-diarization-3.1`-style directory: config.yaml + two reference-format
    checkpoints under $model/segmentation and $model/embedding."""
    import yaml
    from pyannote_audio_amd.model import (PyanNet, WeSpeakerResNet34, embedding_specifications,
                                          save_checkpoint, segmentation_specifications)
    root = str(root)
    os.makedirs(os.path.join(root, "segmentation"), exist_ok=True)
    os.makedirs(os.path.join(root, "embedding"), exist_ok=True)
    save_checkpoint(os.path.join(root, "segmentation", "pytorch_model.bin"), seg_model.state_dict(),
                    PYANNET_HPARAMS, PyanNet.ARCHITECTURE, segmentation_specifications(10.0, powerset))
    if hasattr(emb_model, "tdnns"):        # XVectorSincNet (models/embedding/xvector.py:205-252)
        from pyannote_audio_amd.model import XVectorSincNet
        save_checkpoint(os.path.join(root, "embedding", "pytorch_model.bin"), emb_model.state_dict(),
                        {"sincnet": {"stride": 10, "sample_rate": 16000}, "dimension": emb_model.embedding.out_features,
                         "sample_rate": 16000, "num_channels": 1},
                        XVectorSincNet.ARCHITECTURE, embedding_specifications())
    else:
        save_checkpoint(os.path.join(root, "embedding", "pytorch_model.bin"), emb_model.state_dict(),
                        WESPEAKER_HPARAMS, WeSpeakerResNet34.ARCHITECTURE, embedding_specifications())
    config = {
        "version": "3.1.0",
        "pipeline": {"name": "pyannote.audio.pipelines.SpeakerDiarization",
                     "params": {"clustering": "AgglomerativeClustering",
                                "embedding": "$model/embedding", "embedding_batch_size": 32,
                                "embedding_exclude_overlap": True,
                                "segmentation": "$model/segmentation", "segmentation_batch_size": 32}},
        "params": {"clustering": {"method": "centroid", "min_cluster_size": 12,
                                  "threshold": 0.7045654963945799},
                   "segmentation": {"min_duration_off": 0.0} if powerset else
                   {"threshold": 0.5, "min_duration_off": 0.0}},
    }
    if config_extra:
        config.update(config_extra)   # (whole top-level sections are replaced)
    with open(os.path.join(root, "config.yaml"), "w") as fp:
        yaml.safe_dump(config, fp)
    return os.path.join(root, "config.yaml")


@pytest.fixture(scope="session")
def synthetic_models():
    from oracle.synthetic import calibrated_pyannet, calibrated_wespeaker
    return calibrated_pyannet(calib_seconds=60.0), calibrated_wespeaker(calib_seconds=24.0)


@pytest.fixture(scope="session")
def pipeline_dir(tmp_path_factory, synthetic_models):
    d = tmp_path_factory.mktemp("sd31")
    write_pipeline_dir(d, *synthetic_models)
    return str(d)
