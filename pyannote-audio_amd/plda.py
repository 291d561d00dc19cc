"""PLDA for VBx clustering (mirrors core/plda.py:33-135 and utils/vbx.py:181-218).

Loading = the reference's algebra in float64 on the host, once (two 128x128 inversions and one
generalised eigen-decomposition with the same scipy `eigh` call); the per-file projection of the
training embeddings into the PLDA space runs on the GPU (`pa_plda_transform`, csrc/vbx.hip)."""
from __future__ import annotations

import os
from pathlib import Path
from typing import Optional, Union

import numpy as np
import torch

from . import ffi


class PLDA:
    def __init__(self, transform_npz: Union[str, Path], plda_npz: Union[str, Path],
                 lda_dimension: int = 128):
        from scipy.linalg import eigh
        x = np.load(transform_npz)
        p = np.load(plda_npz)
        self.mean1 = np.asarray(x["mean1"], dtype=np.float64)
        self.mean2 = np.asarray(x["mean2"], dtype=np.float64)
        self.lda = np.asarray(x["lda"], dtype=np.float64)            # (din, dmid)
        self.plda_mu = np.asarray(p["mu"], dtype=np.float64)
        tr, psi = p["tr"], p["psi"]
        # within- and between-class covariances, then the whitening basis sorted by decreasing
        # between-class variance (utils/vbx.py:194-203)
        within = np.linalg.inv(tr.T.dot(tr))
        between = np.linalg.inv((tr.T / psi).dot(tr))
        acvar, wccn = eigh(between, within)
        self._psi = acvar[::-1]
        self._tr = wccn.T[::-1]                                        # (dmid, dmid)
        self.lda_dimension = lda_dimension
        self.device: Optional[torch.device] = None
        self._dev: dict = {}

    @property
    def phi(self) -> np.ndarray:
        """between-class covariance (diagonal) in the PLDA space (core/plda.py:43-45)"""
        return self._psi[: self.lda_dimension]

    def to(self, device: torch.device) -> "PLDA":
        self.device = device
        self._dev = {}
        return self

    def _tensors(self, device):
        key = (str(device), int(self.lda_dimension))
        if self._dev.get("key") != key:
            up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(device)  # noqa: E731
            self._dev = {"mean1": up(self.mean1), "lda": up(self.lda), "mean2": up(self.mean2),
                         "mu": up(self.plda_mu), "trT": up(self._tr[: self.lda_dimension].T), "key": key}
        return self._dev

    def transform_device(self, embeddings: np.ndarray, device: torch.device) -> torch.Tensor:
        """(n, din) float32 -> (n, lda_dimension) float64 DEVICE tensor"""
        n, din = embeddings.shape
        dmid = self.lda.shape[1]
        if din != self.lda.shape[0]:
            raise ValueError(f"PLDA expects {self.lda.shape[0]}-dimensional embeddings, got {din}")
        t = self._tensors(device)
        X = torch.from_numpy(np.ascontiguousarray(embeddings, dtype=np.float32)).to(device)
        fea = torch.empty((n, self.lda_dimension), dtype=torch.float64, device=device)
        with torch.cuda.device(device):
            ffi.check(ffi.load().pa_plda_transform(
                ffi.ptr(X), n, din, dmid, self.lda_dimension, ffi.ptr(t["mean1"]), ffi.ptr(t["lda"]),
                ffi.ptr(t["mean2"]), ffi.ptr(t["mu"]), ffi.ptr(t["trT"]), ffi.ptr(fea), ffi.stream()),
                "pa_plda_transform")
        return fea

    def __call__(self, embeddings: np.ndarray) -> np.ndarray:
        """core/plda.py:47-60: embeddings -> PLDA space (host array; computed on the GPU)"""
        if self.device is None or self.device.type != "cuda":
            raise RuntimeError("PLDA must be moved to the GPU first (pipeline.to(torch.device('cuda'))): "
                               "there is no CPU path")
        return self.transform_device(embeddings, self.device).cpu().numpy()

    @classmethod
    def from_pretrained(cls, checkpoint: Union[Path, str], subfolder: Optional[str] = None,
                        revision: Optional[str] = None, token=None, cache_dir=None, **kwargs
                        ) -> Optional["PLDA"]:
        """core/plda.py:62-135, local directories only (no Hugging Face hub in this build)."""
        if not os.path.isdir(checkpoint):
            raise ValueError(f"{checkpoint}: PLDA checkpoints must be local directories holding "
                             "xvec_transform.npz and plda.npz (hub downloads are not available)")
        if revision is not None:
            raise ValueError("Revisions cannot be used with local checkpoints.")
        root = Path(checkpoint) / subfolder if subfolder else Path(checkpoint)
        return cls(root / "xvec_transform.npz", root / "plda.npz")


def get_plda(plda, token=None, cache_dir=None) -> Optional[PLDA]:
    """pipelines/utils/getter.py: PLDA instance, directory, or kwargs of PLDA.from_pretrained"""
    if plda is None or isinstance(plda, PLDA):
        return plda
    if isinstance(plda, (str, Path)):
        return PLDA.from_pretrained(plda)
    if isinstance(plda, dict):
        kw = dict(plda)
        kw.pop("token", None)
        kw.pop("cache_dir", None)
        return PLDA.from_pretrained(**kw)
    raise TypeError(f"Unsupported type ({type(plda)}) for loading PLDA: expected `str` or `dict`.")
