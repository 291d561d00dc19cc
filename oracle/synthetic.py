"""Synthetic audio + calibrated synthetic checkpoints (TEST INFRASTRUCTURE ONLY).

No pretrained weights or corpora are reachable offline, and a randomly
initialised PyanNet predicts one constant class.  To exercise the WHOLE pipeline (speaker counting,
overlap exclusion, clustering with several clusters, reconstruction) we synthesise a multi-speaker
"conversation" with known activity and fit ONLY the last Linear layer of the seeded network by ridge
regression on its own (random) LSTM features -- an extreme-learning-machine read-out.  The result is a
checkpoint in the reference layout whose outputs vary with the input like a trained model's do."""
from __future__ import annotations

import numpy as np
import torch

from .models import PyanNet, WeSpeakerResNet34, seeded_pyannet, seeded_wespeaker


def synth_conversation(duration_s: float, sr: int = 16000, num_speakers: int = 3, seed: int = 0,
                       overlap_prob: float = 0.25):
    """-> waveform (1, n) float32 in [-1, 1], activity (num_speakers, n) bool."""
    rng = np.random.default_rng(seed)
    n = int(duration_s * sr)
    t = np.arange(n) / sr
    wav = np.zeros(n, dtype=np.float64)
    act = np.zeros((num_speakers, n), dtype=bool)
    voices = []
    for s in range(num_speakers):
        f0 = 95 + 55 * s + rng.uniform(-5, 5)
        sig = np.zeros(n)
        for h in range(1, 14):
            amp = (1.0 / h) * np.exp(-((f0 * h - (450 + 500 * s)) / 1000.0) ** 2)
            sig += amp * np.sin(2 * np.pi * f0 * h * t * (1 + 0.01 * np.sin(2 * np.pi * 0.7 * t))
                                + rng.uniform(0, 6.28))
        sig *= 0.6 + 0.4 * np.sin(2 * np.pi * (3 + s) * t)
        voices.append(sig / np.abs(sig).max())

    def add(s, a, b):
        a, b = max(0, a), min(n, b)
        if b - a < 200:
            return
        env = np.ones(b - a)
        k = min(400, (b - a) // 2)
        env[:k] = np.linspace(0, 1, k)
        env[-k:] = np.linspace(1, 0, k)
        wav[a:b] += 0.3 * voices[s][a:b] * env
        act[s, a:b] = True

    pos = 0.0
    while pos < duration_s:
        pos += rng.uniform(0.1, 1.2)
        s = int(rng.integers(num_speakers))
        d = rng.uniform(0.8, 4.0)
        a, b = int(pos * sr), int((pos + d) * sr)
        if a >= n:
            break
        add(s, a, b)
        if rng.uniform() < overlap_prob and num_speakers > 1:
            s2 = (s + 1 + int(rng.integers(num_speakers - 1))) % num_speakers
            a2 = a + (b - a) // 2
            b2 = b + int(rng.uniform(0.3, 1.5) * sr)
            add(s2, a2, b2)
            pos = min(b2, n) / sr
        else:
            pos = min(b, n) / sr
    wav += 0.003 * rng.standard_normal(n)
    return torch.from_numpy(np.clip(wav, -1, 1).astype(np.float32))[None], act


_POWERSET = [(), (0,), (1,), (2,), (0, 1), (0, 2), (1, 2)]


def _frame_targets(act_chunk: np.ndarray, num_frames: int, rf_size=991, rf_step=270) -> np.ndarray:
    """powerset class per frame; local speaker index = order of first activity in the chunk."""
    S, N = act_chunk.shape
    centers = (np.arange(num_frames) * rf_step + rf_size // 2).clip(0, N - 1)
    a = act_chunk[:, centers]                       # (S, F)
    first = [np.argmax(a[s]) if a[s].any() else 10 ** 9 for s in range(S)]
    order = np.argsort(first, kind="stable")
    a = a[order][:3]
    cls = np.zeros(num_frames, dtype=np.int64)
    for f in range(num_frames):
        active = tuple(i for i in range(a.shape[0]) if a[i, f])[:2]
        cls[f] = _POWERSET.index(active)
    return cls


def uncalibrated_pyannet(seed: int = 1234, num_layers: int = 4) -> PyanNet:
    """The deterministic part of `calibrated_pyannet`: seeded weights, recurrent / linear stacks scaled."""
    model = seeded_pyannet(seed=seed, num_layers=num_layers, classifier_gain=1.0)
    # default-initialised LSTM/Linear stacks barely react to their input (feature std ~5e-4): scale
    # them so that the read-out has something time-varying to work with
    with torch.no_grad():
        for name, p in model.lstm.named_parameters():
            if "weight_ih" in name:
                p.mul_(4.0)
            elif "weight_hh" in name:
                p.mul_(2.0)
        for lin in model.linear:
            lin.weight.mul_(3.0)
    return model


def models_from_readout(classifier_weight, classifier_bias, seg1_bias, seg_seed: int = 1234,
                        emb_seed: int = 4321):
    """(PyanNet, WeSpeakerResNet34) with a STORED read-out instead of a fitted one: the fit is a float64
    solve whose last bits depend on the host's BLAS, so golden vectors carry the fitted parameters
    (tests/golden/reference_v1.npz) and every box rebuilds bit-identical checkpoints from them."""
    seg = uncalibrated_pyannet(seed=seg_seed)
    emb = seeded_wespeaker(seed=emb_seed)
    with torch.no_grad():
        seg.classifier.weight.copy_(torch.as_tensor(classifier_weight))
        seg.classifier.bias.copy_(torch.as_tensor(classifier_bias))
        emb.resnet.seg_1.bias.copy_(torch.as_tensor(seg1_bias))
    return seg, emb


def calibrated_multilabel_pyannet(seed: int = 1234, num_layers: int = 4, calib_seconds: float = 60.0,
                                  chunk_s: float = 10.0, ridge: float = 1e-2, gain: float = 8.0) -> PyanNet:
    """NON-powerset variant (multi-label problem: one sigmoid score per local speaker, core/model.py:
    286-294): same seeded trunk, a ridge read-out fitted to the three per-speaker activities."""
    base = uncalibrated_pyannet(seed=seed, num_layers=num_layers)
    model = PyanNet(num_classes=3, lstm={"num_layers": num_layers}, powerset=False)
    sd = {k: v for k, v in base.state_dict().items() if not k.startswith("classifier.")}
    model.load_state_dict(sd, strict=False)
    model.eval()
    wav, act = synth_conversation(calib_seconds, seed=seed + 1)
    N = int(chunk_s * 16000)
    feats, targets = [], []
    hooked = {}
    handle = model.classifier.register_forward_hook(lambda m, i, o: hooked.__setitem__("x", i[0]))
    with torch.inference_mode():
        for s in range(0, wav.shape[1] - N + 1, N // 2):
            model(wav[:, s:s + N][None])
            x = hooked["x"][0].numpy()
            feats.append(x)
            cls = _frame_targets(act[:, s:s + N], x.shape[0])
            targets.append(np.array([[1.0 if k in _POWERSET[c] else 0.0 for k in range(3)] for c in cls]))
    handle.remove()
    X = np.concatenate(feats).astype(np.float64)
    Y = np.concatenate(targets)
    mu, ym = X.mean(0), Y.mean(0)
    Xc = X - mu
    W = np.linalg.solve(Xc.T @ Xc + ridge * len(X) * np.eye(X.shape[1]), Xc.T @ (Y - ym))   # (128, 3)
    b = ym - mu @ W
    with torch.no_grad():     # sigmoid(gain * (regression - 1/2)): > 0.5 where the regression says "active"
        model.classifier.weight.copy_(torch.from_numpy((gain * W.T).astype(np.float32)))
        model.classifier.bias.copy_(torch.from_numpy((gain * (b - 0.5)).astype(np.float32)))
    return model


def calibrated_pyannet(seed: int = 1234, num_layers: int = 4, calib_seconds: float = 120.0,
                       chunk_s: float = 10.0, ridge: float = 1e-2, gain: float = 6.0) -> PyanNet:
    """seeded PyanNet whose classifier is a ridge read-out fitted on a synthetic conversation."""
    model = uncalibrated_pyannet(seed=seed, num_layers=num_layers)
    wav, act = synth_conversation(calib_seconds, seed=seed + 1)
    N = int(chunk_s * 16000)
    starts = list(range(0, wav.shape[1] - N + 1, N // 2))
    feats, targets = [], []
    hooked = {}
    handle = model.classifier.register_forward_hook(lambda m, i, o: hooked.__setitem__("x", i[0]))
    with torch.inference_mode():
        for s in starts:
            model(wav[:, s:s + N][None])
            x = hooked["x"][0].numpy()
            feats.append(x)
            targets.append(_frame_targets(act[:, s:s + N], x.shape[0]))
    handle.remove()
    X = np.concatenate(feats).astype(np.float64)
    y = np.concatenate(targets)
    mu = X.mean(0)
    Xc = X - mu
    Y = np.eye(7)[y] - np.eye(7)[y].mean(0)
    W = np.linalg.solve(Xc.T @ Xc + ridge * len(X) * np.eye(X.shape[1]), Xc.T @ Y)   # (128, 7)
    b = np.eye(7)[y].mean(0) - mu @ W
    with torch.no_grad():
        model.classifier.weight.copy_(torch.from_numpy((gain * W.T).astype(np.float32)))
        model.classifier.bias.copy_(torch.from_numpy((gain * b).astype(np.float32)))
    return model


def calibrated_wespeaker(seed: int = 4321, calib_seconds: float = 40.0) -> WeSpeakerResNet34:
    """seeded ResNet34 whose seg_1 bias centres the embeddings of a synthetic conversation, so that
    cosine distances between (random-feature) embeddings reflect who is speaking instead of a large
    common offset."""
    model = seeded_wespeaker(seed=seed)
    wav, act = synth_conversation(calib_seconds, seed=seed + 1)
    N = 48000
    stats = []
    hooked = {}
    handle = model.resnet.seg_1.register_forward_hook(lambda m, i, o: hooked.__setitem__("x", i[0]))
    with torch.inference_mode():
        for s in range(0, wav.shape[1] - N + 1, N):
            model(wav[:, s:s + N][None])
            stats.append(hooked["x"][0])
    handle.remove()
    mu = torch.stack(stats).mean(0)
    with torch.no_grad():
        model.resnet.seg_1.bias.copy_(-(model.resnet.seg_1.weight @ mu))
    return model
