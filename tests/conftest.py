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
