"""Host logic of SpeakerDiarization._embed_speech_chunks (the gather of the chunks that have an active speaker, the
representative of the others, the scatter) against a stand-in engine that follows the kernels' chunking contract:
chunk b = wav[b * stride : b * stride + window], zeros past the end; an empty mask yields a constant row."""
from types import SimpleNamespace

import torch

from pyannote_audio_amd.speaker_diarization import SpeakerDiarization


class _Engine:
    def __init__(self):
        self.calls = []

    def forward_strided(self, wav, stride, n, window, masks):
        self.calls.append((int(wav.numel()), stride, n, window))
        out = torch.empty((n, masks.shape[1], 3))
        ramp = torch.arange(window, dtype=torch.float64)
        for b in range(n):
            chunk = torch.zeros(window, dtype=torch.float64)
            piece = wav[b * stride: b * stride + window].double()
            chunk[:piece.numel()] = piece
            for s in range(masks.shape[1]):
                m = float(masks[b, s].sum())
                out[b, s] = torch.tensor([float((chunk * ramp).sum()) if m > 0 else 0.0, m, 7.0])
        return out


def _run(skip, wav, step, C, window, masks):
    me = SimpleNamespace(skip_inactive_chunks=skip, last_embedded_chunks=(0, 0))
    eng = _Engine()
    emb = SpeakerDiarization._embed_speech_chunks(me, eng, wav, step, C, window, masks)
    return emb, me.last_embedded_chunks, eng.calls


def test_skip_equals_full_run_and_launches_only_the_speech_chunks():
    g = torch.Generator().manual_seed(0)
    window, step, C, S, F = 50, 10, 23, 3, 11
    wav = torch.randn((C - 1) * step + window - 17, generator=g)    # the last chunk runs 17 samples past the end
    masks = (torch.rand((C, S, F), generator=g) > 0.5).float()
    silent = [0, 4, 5, 6, 13, C - 1]
    masks[silent] = 0.0
    masks[9, 1:] = 0.0                                              # one active speaker is enough to keep a chunk
    full, done_full, calls_full = _run(False, wav, step, C, window, masks)
    short, done_short, calls_short = _run(True, wav, step, C, window, masks)
    assert done_full == (C, C) and calls_full == [(wav.numel(), step, C, window)]
    kept = C - len(silent)
    assert done_short == (C, kept + 1)
    assert calls_short == [((kept + 1) * window, window, kept + 1, window)]   # compact buffer: stride = window
    assert torch.equal(full, short)


def test_nothing_or_one_chunk_to_skip_takes_the_plain_launch():
    g = torch.Generator().manual_seed(1)
    window, step, C, S, F = 40, 8, 9, 2, 5
    wav = torch.randn((C - 1) * step + window, generator=g)
    masks = torch.ones((C, S, F))
    for silent in ([], [3]):
        m = masks.clone()
        m[silent] = 0.0
        emb, done, calls = _run(True, wav, step, C, window, m)
        assert done == (C, C) and calls == [(wav.numel(), step, C, window)]
    # masks=None (plain embedding extraction): never touched
    me = SimpleNamespace(skip_inactive_chunks=True, last_embedded_chunks=(0, 0))

    class E:
        def forward_strided(self, *a):
            return "plain"
    assert SpeakerDiarization._embed_speech_chunks(me, E(), wav, step, C, window, None) == "plain"
