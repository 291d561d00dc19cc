"""GPU parity of the frame-domain kernels (csrc/frames.hip) against the CPU oracle's restatement of
speaker_count / reconstruct / to_diarization / filter statistics: bit-exact (integer work)."""
import numpy as np
import pytest
import torch

from pyannote_audio_amd import frames as frame_ops
from pyannote_audio_amd.core import SlidingWindow

pytestmark = pytest.mark.gpu


def _random_segmentation(C, F, S, seed):
    rng = np.random.default_rng(seed)
    seg = np.zeros((C, F, S), dtype=np.float32)
    for c in range(C):
        for s in range(S):
            if rng.uniform() < 0.75:
                for _ in range(rng.integers(1, 4)):
                    a = rng.integers(0, F)
                    b = min(F, a + rng.integers(1, F))
                    seg[c, a:b, s] = 1
    return seg


@pytest.mark.parametrize("C,F,S,dur,step,seed", [(37, 589, 3, 10.0, 1.0, 0), (64, 293, 3, 5.0, 0.5, 1),
                                                  (1, 589, 3, 10.0, 1.0, 2), (25, 589, 4, 10.0, 2.5, 3)])
def test_count_and_reconstruct_bit_exact(gpu_device, C, F, S, dur, step, seed):
    from oracle import pipeline as op
    from pyannote_audio_amd import frames as fo
    from pyannote_audio_amd.core import SlidingWindow
    seg = _random_segmentation(C, F, S, seed)
    rng = np.random.default_rng(100 + seed)
    chunks_o, frames_o = op.SW(0.0, dur, step), op.SW(0.0, 0.0619375, 0.016875)
    chunks = SlidingWindow(start=0.0, duration=dur, step=step)
    frames = SlidingWindow(start=0.0, duration=0.0619375, step=0.016875)
    dev = fo.as_device_segmentation(seg, gpu_device)

    want_count, _ = op.speaker_count(seg, chunks_o, frames_o)
    got_count = fo.speaker_count(dev, chunks, frames)
    assert got_count.data.dtype == np.uint8 and got_count.data.shape == want_count.shape
    assert np.array_equal(got_count.data, want_count)

    for K in (1, 2, 5):
        hard = rng.integers(0, K, size=(C, S))
        hard[rng.uniform(size=(C, S)) < 0.2] = -2
        hard[0, 0] = K - 1                      # keep max(hard) + 1 == K
        for cap in (255, 1):
            count = np.minimum(want_count, cap).astype(np.int8)
            want = op.reconstruct(seg, chunks_o, hard, count, frames_o)
            rec = fo.Reconstructor(dev, chunks, frames, hard, count)
            got = rec.discretize().data
            assert got.shape == want.shape, (K, cap)
            assert np.array_equal(got, want), (K, cap)
            if cap == 255:
                got1 = rec.discretize(cap=1).data
                want1 = op.reconstruct(seg, chunks_o, hard, np.minimum(want_count, 1).astype(np.int8),
                                       frames_o)
                assert np.array_equal(got1[:, :want1.shape[1]], want1)
                assert not got1[:, want1.shape[1]:].any()


def test_chunk_stats_and_masks(gpu_device):
    from pyannote_audio_amd import frames as fo
    C, F, S = 50, 589, 3
    seg = _random_segmentation(C, F, S, 7)
    dev = fo.as_device_segmentation(seg, gpu_device)
    active, clean = (t.cpu().numpy() for t in fo.chunk_stats(dev))
    single = seg.sum(axis=2, keepdims=True) == 1
    assert np.array_equal(active, seg.sum(axis=1).astype(np.int32))
    assert np.array_equal(clean, (seg * single).sum(axis=1).astype(np.int32))
    # get_embeddings mask selection (speaker_diarization.py:375-427)
    for min_num_frames, exclude in ((24, True), (-1, False), (400, True)):
        masks = fo.embedding_masks(dev, fo.chunk_stats(dev)[1], exclude, min_num_frames).cpu().numpy()
        clean_seg = seg * (seg.sum(axis=2, keepdims=True) < 2)
        want = np.empty((C, S, F), dtype=np.float32)
        for c in range(C):
            for s in range(S):
                use_clean = exclude and clean_seg[c, :, s].sum() > min_num_frames
                want[c, s] = clean_seg[c, :, s] if use_clean else seg[c, :, s]
        assert np.array_equal(masks, want)


# ---------------------------------------------------------------------------- a14: non-powerset (soft) path
def test_hysteresis_kernel_matches_oracle(gpu_device):
    """pa_binarize_hysteresis vs oracle.pipeline.hysteresis (= the reference's `binarize`, pinned by
    tests/test_reference_pipeline.py): NaN scores, scores exactly on a threshold, all initial states."""
    from oracle import pipeline as op
    rng = np.random.default_rng(6)
    scores = rng.uniform(size=(37, 589, 3)).astype(np.float32)
    scores[2, 10:14, 1] = np.nan
    scores[3, ::9, 0] = 0.6
    scores[4, ::7, 2] = 0.4
    dev = torch.from_numpy(scores).to(gpu_device)
    for onset, offset, init in [(0.5, None, False), (0.6, 0.4, None), (0.6, 0.4, True), (0.3, None, None)]:
        got = frame_ops.binarize(dev, onset=onset, offset=offset, initial_state=init).cpu().numpy()
        want = op.hysteresis(scores, onset=onset, offset=offset, initial_state=init)
        assert got.dtype == np.uint8 and np.array_equal(got.astype(np.float64), want)


def test_soft_reconstruction_bit_exact(gpu_device):
    """reconstruct + to_diarization from SOFT scores (non-powerset models: speaker_diarization.py:687-691):
    cluster max -> float32 overlap-add sum in chunk order -> top-count[t], identical to the oracle's."""
    from oracle import pipeline as op
    rng = np.random.default_rng(3)
    C, F, S = 61, 589, 3
    scores = rng.uniform(size=(C, F, S)).astype(np.float32)
    hard = rng.integers(-2, 4, size=(C, S))
    hard[hard == -1] = 0
    hard[7] = -2                                  # a chunk with no assigned speaker at all
    chunks, frames = op.SW(0.0, 10.0, 1.0), op.SW(0.0, 991 / 16000, 270 / 16000)
    binar = (scores > 0.5).astype(np.float32)
    count, cf = op.speaker_count(binar, chunks, frames)
    for cap in (255, 1):
        cnt = np.minimum(count, cap).astype(np.int8)
        want = op.reconstruct(scores, chunks, hard.copy(), cnt, cf)
        rec = frame_ops.Reconstructor(torch.from_numpy(scores).to(gpu_device),
                                      SlidingWindow(start=0.0, duration=10.0, step=1.0),
                                      SlidingWindow(start=0.0, duration=991 / 16000, step=270 / 16000),
                                      hard, cnt)
        got = rec.discretize().data
        assert got.shape == want.shape and np.array_equal(got, want)


def test_soft_reconstruction_full_hour(gpu_device):
    """the soft (non-powerset) reconstruction at BASELINE configs[3] size: 3 591 chunks, 213 334 frames --
    float32 overlap-add in chunk order + float top-k, bit-identical to the oracle's chunk / frame loops."""
    from oracle import pipeline as op
    rng = np.random.default_rng(9)
    C, F, S = 3591, 589, 3
    scores = rng.uniform(size=(C, F, S)).astype(np.float32)
    hard = rng.integers(0, 4, size=(C, S))
    hard[rng.uniform(size=(C, S)) < 0.15] = -2
    chunks, frames = op.SW(0.0, 10.0, 1.0), op.SW(0.0, 991 / 16000, 270 / 16000)
    count, cf = op.speaker_count((scores > 0.5).astype(np.float32), chunks, frames)
    cnt = np.minimum(count, 2).astype(np.int8)
    want = op.reconstruct(scores, chunks, hard.copy(), cnt, cf)
    rec = frame_ops.Reconstructor(torch.from_numpy(scores).to(gpu_device),
                                  SlidingWindow(start=0.0, duration=10.0, step=1.0),
                                  SlidingWindow(start=0.0, duration=991 / 16000, step=270 / 16000), hard, cnt)
    got = rec.discretize().data
    assert got.shape == want.shape == (213334, 4) and np.array_equal(got, want)
