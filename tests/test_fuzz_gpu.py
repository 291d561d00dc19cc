"""Seeded random shapes through the convolution kernels, the fbank front end and the statistics pooling (MI355X): the
hand-picked cases of tests/test_emb_gpu.py cover the ResNet's own shapes and the edges somebody thought of; these cover
the ones nobody did -- every extent from 1 up, every supported channel pair, with / without residual and ReLU, guard
bands of NaN around every output.  Same references (torch on the CPU) and the same bounds as the hand-picked tests."""
import ctypes as C
import os

import pytest
import torch
import torch.nn.functional as F

from conftest import north_star_ratio

pytestmark = pytest.mark.gpu

# extended sessions: PA_FUZZ_SEED_OFFSET=k shifts every seed below (the default suite runs offset 0; profiles/ holds the
# log of a session over other offsets)
SEED_OFFSET = int(os.environ.get("PA_FUZZ_SEED_OFFSET", "0"))


def _conv_case(rng, channels, max_b=24, max_h=44, max_w=260):
    cin, cout = channels[int(torch.randint(len(channels), (1,), generator=rng))]
    H = int(torch.randint(1, max_h + 1, (1,), generator=rng))
    W = int(torch.randint(1, max_w + 1, (1,), generator=rng))
    B = int(torch.randint(1, max_b + 1, (1,), generator=rng))
    # keep the CPU reference affordable: ~2e9 multiply-adds at most
    while B > 1 and B * H * W * cin * cout * 9 > 2e9:
        B //= 2
    return cin, cout, H, W, B, bool(torch.randint(2, (1,), generator=rng)), bool(torch.randint(2, (1,), generator=rng))


def _tensors(rng, cin, cout, H, W, B, stride=1):
    x = torch.randn(B, cin, H, W, generator=rng)
    wt = torch.randn(cout, cin, 3, 3, generator=rng) / (3 * cin ** 0.5)
    sh = torch.randn(cout, generator=rng)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    res = torch.randn(B, cout, Ho, Wo, generator=rng)
    return x, wt, sh, res, Ho, Wo


def _reference(x, wt, sh, res, use_res, relu, stride=1):
    ref = F.conv2d(x, wt, stride=stride, padding=1) + sh.view(1, -1, 1, 1)
    if use_res:
        ref = ref + res
    return F.relu(ref) if relu else ref


def _guarded(B, Ho, Wo, cout, device):
    buf = torch.full((B + 2, Ho, Wo, cout), float("nan"), device=device)
    return buf, buf[1:B + 1]


@pytest.mark.parametrize("seed", [101, 202])
def test_fuzz_winograd_f4(gpu_device, seed):
    import pyannote_audio_amd.ffi as ffi
    from pyannote_audio_amd.weights import winograd4_pack, winograd4_weights
    lib = ffi.load()
    rng = torch.Generator().manual_seed(seed + SEED_OFFSET)
    chans = [(32, 32), (64, 64), (128, 128), (256, 256), (64, 32), (32, 64), (40, 32), (128, 64)]
    for _ in range(14):
        cin, cout, H, W, B, use_res, relu = _conv_case(rng, chans)
        x, wt, sh, res, Ho, Wo = _tensors(rng, cin, cout, H, W, B)
        ref = _reference(x, wt, sh, res, use_res, relu)
        xd = x.permute(0, 2, 3, 1).contiguous().to(gpu_device)
        ud = winograd4_pack(winograd4_weights(wt)).to(gpu_device)
        rd = res.permute(0, 2, 3, 1).contiguous().to(gpu_device)
        shd = sh.to(gpu_device)
        buf, y = _guarded(B, H, W, cout, gpu_device)
        ffi.check(lib.pa_conv3x3_wino4(ffi.ptr(xd), B, H, W, cin, ffi.ptr(ud), ffi.ptr(shd), ffi.ptr(rd) if use_res else None,
                                       C.c_void_p(y.data_ptr()), cout, int(relu), ffi.stream()), "wino4")
        torch.cuda.synchronize()
        case = (cin, cout, H, W, B, use_res, relu)
        assert torch.isnan(buf[0]).all() and torch.isnan(buf[B + 1]).all(), case
        got = y.permute(0, 3, 1, 2).cpu()
        assert not torch.isnan(got).any(), case
        err = (got - ref).abs().max().item() / max(ref.abs().max().item(), 1e-6)
        assert err <= 1e-4 and north_star_ratio(f"fuzz_wino4_{case}", got, ref) <= 8.0, (case, err)


@pytest.mark.parametrize("seed", [303, 404])
def test_fuzz_winograd_f2(gpu_device, seed):
    import pyannote_audio_amd.ffi as ffi
    from pyannote_audio_amd.weights import winograd_pack, winograd_weights
    lib = ffi.load()
    rng = torch.Generator().manual_seed(seed + SEED_OFFSET)
    chans = [(32, 32), (64, 64), (128, 128), (256, 256), (64, 32), (32, 64), (128, 64)]
    for _ in range(14):
        cin, cout, H, W, B, use_res, relu = _conv_case(rng, chans)
        x, wt, sh, res, Ho, Wo = _tensors(rng, cin, cout, H, W, B)
        ref = _reference(x, wt, sh, res, use_res, relu)
        xd = x.permute(0, 2, 3, 1).contiguous().to(gpu_device)
        ud = winograd_pack(winograd_weights(wt)).to(gpu_device)
        rd = res.permute(0, 2, 3, 1).contiguous().to(gpu_device)
        shd = sh.to(gpu_device)
        buf, y = _guarded(B, H, W, cout, gpu_device)
        ffi.check(lib.pa_conv3x3_wino(ffi.ptr(xd), B, H, W, cin, ffi.ptr(ud), ffi.ptr(shd), ffi.ptr(rd) if use_res else None,
                                      C.c_void_p(y.data_ptr()), cout, int(relu), ffi.stream()), "wino")
        torch.cuda.synchronize()
        case = (cin, cout, H, W, B, use_res, relu)
        assert torch.isnan(buf[0]).all() and torch.isnan(buf[B + 1]).all(), case
        got = y.permute(0, 3, 1, 2).cpu()
        assert not torch.isnan(got).any(), case
        assert north_star_ratio(f"fuzz_wino_{case}", got, ref) <= 1.0, case


@pytest.mark.parametrize("seed,stride", [(505, 1), (606, 2), (707, 2)])
def test_fuzz_direct_conv(gpu_device, seed, stride):
    import pyannote_audio_amd.ffi as ffi
    lib = ffi.load()
    rng = torch.Generator().manual_seed(seed + SEED_OFFSET)
    chans = [(32, 32), (64, 64), (128, 128), (256, 256)] if stride == 1 else [(32, 64), (64, 128), (128, 256)]
    for _ in range(12):
        cin, cout, H, W, B, use_res, relu = _conv_case(rng, chans)
        x, wt, sh, res, Ho, Wo = _tensors(rng, cin, cout, H, W, B, stride)
        ref = _reference(x, wt, sh, res, use_res, relu, stride)
        xd = x.permute(0, 2, 3, 1).contiguous().to(gpu_device)
        wd = wt.permute(2, 3, 0, 1).reshape(9, cout, cin).contiguous().to(gpu_device)
        rd = res.permute(0, 2, 3, 1).contiguous().to(gpu_device)
        shd = sh.to(gpu_device)
        buf, y = _guarded(B, Ho, Wo, cout, gpu_device)
        ffi.check(lib.pa_conv3x3(ffi.ptr(xd), B, H, W, cin, ffi.ptr(wd), ffi.ptr(shd), ffi.ptr(rd) if use_res else None,
                                 C.c_void_p(y.data_ptr()), cout, stride, int(relu), ffi.stream()), "conv3x3")
        torch.cuda.synchronize()
        case = (cin, cout, H, W, B, use_res, relu, stride)
        assert torch.isnan(buf[0]).all() and torch.isnan(buf[B + 1]).all(), case
        got = y.permute(0, 3, 1, 2).cpu()
        assert not torch.isnan(got).any(), case
        assert north_star_ratio(f"fuzz_conv_{case}", got, ref) <= 1.0, case


def test_fuzz_fbank_lengths(gpu_device):
    """pa_fbank at lengths around every framing edge (one frame exactly, one sample more, one sample short of the next
    hop, ...): frame count and values against the oracle's kaldi fbank (energy domain, the bound of test_fbank)"""
    import pyannote_audio_amd.ffi as ffi
    from oracle import seeded_wespeaker
    from pyannote_audio_amd.weights import EmbeddingPack
    lib = ffi.load()
    model = seeded_wespeaker(seed=4321)
    pack = EmbeddingPack(model.state_dict(), gpu_device, guard=False)   # (kept alive: it owns the device tables)
    w = pack.struct
    rng = torch.Generator().manual_seed(9 + SEED_OFFSET)
    lengths = [400, 401, 559, 560, 561, 719, 720, 1000, 4799, 4800, 16000, 23456, 48001]
    for N in lengths:
        B = 2
        x = (0.1 * torch.randn(B, 1, N, generator=rng)).clamp(-1, 1)
        with torch.inference_mode():
            ref = model.compute_fbank(x)
        T = lib.pa_emb_num_fbank_frames(N)
        assert T == ref.shape[1] == 1 + (N - 400) // 160, N
        xd = x.view(-1).to(gpu_device)
        out = torch.full((B, T, 80), float("nan"), device=gpu_device)
        ffi.check(lib.pa_fbank(ffi.ptr(xd), xd.numel(), N, B, N, w.fb_window, w.fb_tw256, w.fb_tw512, w.fb_mel_w,
                               w.fb_mel_lo, w.fb_mel_hi, 80, ffi.ptr(out), 1, ffi.stream()), "fbank")
        torch.cuda.synchronize()
        # energy domain, relative to max(value, 1e-3 x the chunk's peak) -- against a FLOAT64 evaluation of the oracle:
        # on white noise the float32 oracle itself (pocketfft) is up to ~2e-4 away from it, and so is the kernel's
        # radix-4 FFT; the kernel may be as far as the float32 oracle is (x 2), and never further than 4e-4
        with torch.inference_mode():
            ref64 = model.double().compute_fbank(x.double())
            model.float()
        e64 = torch.exp(ref64)
        scale = torch.maximum(e64, 1e-3 * e64.amax(dim=(1, 2), keepdim=True))
        rel = ((torch.exp(out.cpu().double()) - e64).abs() / scale).max().item()
        rel_oracle = ((torch.exp(ref.double()) - e64).abs() / scale).max().item()
        assert rel < max(2e-4, 2.0 * rel_oracle) and rel < 4e-4, (N, rel, rel_oracle)
    assert lib.pa_emb_num_fbank_frames(399) == 0


def test_fuzz_embedding_lengths_and_masks(gpu_device):
    """whole embedding model at odd chunk lengths with random / empty / single-frame masks against the oracle"""
    from oracle import seeded_wespeaker
    from pyannote_audio_amd.embedding import EmbeddingEngine
    from pyannote_audio_amd.weights import EmbeddingPack
    model = seeded_wespeaker(seed=4321)
    eng = EmbeddingEngine(EmbeddingPack(model.state_dict(), gpu_device), max_chunks=4)
    rng = torch.Generator().manual_seed(33 + SEED_OFFSET)
    for N, Fm in ((16000, 59), (23456, 87), (48000, 173), (8000, 30), (5000, 19)):
        B, S = 5, 3
        x = (0.1 * torch.randn(B, 1, N, generator=rng)).clamp(-1, 1)
        masks = (torch.rand(B, S, Fm, generator=rng) < 0.6).float()
        masks[0, 1] = 0.0                       # nobody speaks
        masks[1, 2] = 0.0
        masks[1, 2, Fm // 2] = 1.0              # one frame
        masks[2] = 1.0                          # everybody, all the time
        with torch.inference_mode():
            ref = model(x, weights=masks)
        out = eng.forward(x.to(gpu_device), masks.to(gpu_device))
        torch.cuda.synchronize()
        assert north_star_ratio(f"fuzz_emb_N{N}", out, ref) <= 1.0, N


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_fuzz_count_and_reconstruct(gpu_device, seed):
    """speaker_count / reconstruct / to_diarization on random chunk grids (1 .. 80 chunks of 1 .. 700 frames, 1 .. 5 local
    speakers, steps from a tenth of a chunk to a whole chunk, up to 8 clusters, unassigned speakers): bit-exact
    against the oracle's restatement of the reference's loops"""
    import numpy as np
    from oracle import pipeline as op
    from pyannote_audio_amd import frames as fo
    from pyannote_audio_amd.core import SlidingWindow
    rng = np.random.default_rng(seed + SEED_OFFSET)
    for _ in range(10):
        C, F, S = int(rng.integers(1, 81)), int(rng.integers(1, 701)), int(rng.integers(1, 6))
        frame_step = 0.016875
        dur = F * frame_step + 0.045          # a chunk of F frames
        step = float(rng.choice([0.1, 0.25, 0.5, 0.73, 1.0])) * dur
        seg = (rng.uniform(size=(C, F, S)) < rng.uniform(0.05, 0.9)).astype(np.float32)
        seg[rng.uniform(size=C) < 0.15] = 0.0                      # silent chunks
        chunks_o, frames_o = op.SW(0.0, dur, step), op.SW(0.0, 0.0619375, frame_step)
        chunks = SlidingWindow(start=0.0, duration=dur, step=step)
        frames = SlidingWindow(start=0.0, duration=0.0619375, step=frame_step)
        dev = fo.as_device_segmentation(seg, gpu_device)
        want_count, _ = op.speaker_count(seg, chunks_o, frames_o)
        got_count = fo.speaker_count(dev, chunks, frames)
        case = (C, F, S, round(step / dur, 2))
        assert got_count.data.shape == want_count.shape and np.array_equal(got_count.data, want_count), case
        K = int(rng.integers(1, 9))
        hard = rng.integers(0, K, size=(C, S))
        hard[rng.uniform(size=(C, S)) < 0.25] = -2
        hard[0, 0] = K - 1
        count = np.minimum(want_count, int(rng.integers(1, 4))).astype(np.int8)
        want = op.reconstruct(seg, chunks_o, hard, count, frames_o)
        got = fo.Reconstructor(dev, chunks, frames, hard, count).discretize().data
        assert got.shape == want.shape and np.array_equal(got, want), case


@pytest.mark.parametrize("seed", [21, 22])
def test_fuzz_segmentation_strided(gpu_device, seed):
    """pa_seg_forward on random chunk grids: 1 .. 40 chunks of 1.2 .. 11 s, hops from 1/16 of a chunk to a whole chunk
    (chunk starts at any sample: the shared-sinc path needs hops that are multiples of the sinc stride, the per-chunk
    path takes the rest), the last chunk running past the end of the waveform by a random amount (zero padded, as
    core/inference.py:270-278 pads an orphan chunk): log-probabilities within the north-star tolerance, hard decisions
    identical outside the 1e-4 gap"""
    from oracle import Powerset, seeded_pyannet
    from pyannote_audio_amd.segmentation import SegmentationEngine
    from pyannote_audio_amd.weights import SegmentationPack
    model = seeded_pyannet(seed=1234, num_layers=4)
    eng = SegmentationEngine(SegmentationPack(model.state_dict(), {"lstm": {"num_layers": 4}}, 7, 3, 2, gpu_device))
    rng = torch.Generator().manual_seed(seed + SEED_OFFSET)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=rng))   # noqa: E731
    for _ in range(6):
        N = ri(19200, 176000)
        B = ri(1, 40)
        stride = max(1, N // ri(1, 16)) if ri(0, 1) else ri(1, N)
        if ri(0, 1):
            stride = max(10, stride // 10 * 10)
        missing = ri(0, N // 2) if B > 1 or ri(0, 1) else 0
        total = stride * (B - 1) + N - missing
        wav = (0.1 * torch.randn(total, generator=rng) + 0.03 * torch.sin(torch.arange(total) * 0.013)).clamp(-1, 1)
        chunks = torch.zeros(B, 1, N)
        for b in range(B):
            piece = wav[b * stride: b * stride + N]
            chunks[b, 0, :piece.numel()] = piece
        with torch.inference_mode():
            ref = model(chunks)
        logp, ml = eng.forward_strided(wav.to(gpu_device), stride, B, N)
        torch.cuda.synchronize()
        case = (B, N, stride, missing)
        assert logp.shape == ref.shape, case
        assert north_star_ratio(f"fuzz_seg_{case}", logp, ref) <= 1.0, case
        top2 = ref.topk(2, dim=-1).values
        safe = (top2[..., 0] - top2[..., 1]) > 1e-4
        assert torch.equal(ml.cpu()[safe], Powerset(3, 2)(ref).to(torch.uint8)[safe]), case


def test_one_large_launch_group_equals_many_small_ones(gpu_device):
    """Since round 6 a launch group of the embedding engine is bounded by its workspace (128 GB; 96 GB at first), not by a chunk count: 7 000
    segments of 3 s run as ONE group (7.6e9 activation elements in layer 1: beyond 32-bit element indices) where they
    used to run as four.  Chunks are independent: the embeddings must be bit-identical to those of groups of 500."""
    from oracle import seeded_wespeaker
    from pyannote_audio_amd.embedding import EmbeddingEngine
    from pyannote_audio_amd.weights import EmbeddingPack
    model = seeded_wespeaker(seed=4321)
    pack = EmbeddingPack(model.state_dict(), gpu_device)
    C, N, step = 7000, 48000, 4000
    g = torch.Generator().manual_seed(77)
    wav = (0.1 * torch.randn(step * (C - 1) + N, generator=g)).clamp(-1, 1).to(gpu_device)
    masks = (torch.rand(C, 3, 173, generator=g) < 0.7).float().to(gpu_device)
    big = EmbeddingEngine(pack)
    assert big._group_size(C, N, 3) == C                      # one group
    small = EmbeddingEngine(pack, max_chunks=500)
    a = big.forward_strided(wav, step, C, N, masks)
    big.release_workspace()
    b = small.forward_strided(wav, step, C, N, masks)
    torch.cuda.synchronize()
    assert a.shape == (C, 3, 256) and not torch.isnan(a).any()
    assert torch.equal(a, b)
    # ... and a handful of them against the oracle
    idx = [0, 1, 3499, 6999]
    chunks = torch.stack([wav[i * step: i * step + N].cpu() for i in idx]).unsqueeze(1)
    with torch.inference_mode():
        ref = model(chunks, weights=masks[idx].cpu())
    assert north_star_ratio("large_group_vs_oracle", a[idx], ref) <= 1.0
