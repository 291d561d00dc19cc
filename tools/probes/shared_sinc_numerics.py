"""CPU numerics study for "the sinc layer once per file instead of once per chunk" (ROUND_NOTES.md, ideas).

Chunks of the 10 s / 1 s sliding window overlap 90 % and start at multiples of 1 600 sinc positions; the waveform
InstanceNorm is affine, so   sinc((x - mu) * g + b) = g * (sinc(x) - mu * S1) + b * S1,   S1[f] = sum of filter f's taps,
g = gamma / sqrt(var + eps).  This script evaluates the segmentation model both ways in float32 (oracle modules, CPU)
on the same audio and reports how far the log-probabilities move and how many hard decisions flip, next to what
float32 itself leaves undecided (the same comparison between float32 and a float64 evaluation of the reference order).
usage: python tools/probes/shared_sinc_numerics.py [seconds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import torch.nn.functional as F
from oracle.synthetic import calibrated_pyannet, synth_conversation

torch.set_num_threads(8)
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
model = calibrated_pyannet().eval()
wav, _ = synth_conversation(seconds, seed=11)
wav = torch.as_tensor(wav, dtype=torch.float32).reshape(1, -1)
N, STEP = 160000, 16000
C = (wav.shape[1] - N) // STEP + 1
chunks = torch.stack([wav[0, c * STEP: c * STEP + N] for c in range(C)]).unsqueeze(1)     # (C, 1, N)
sn = model.sincnet


def rest(first, m):
    """everything after the sinc convolution, from its (C, 80, P) output"""
    s = m.sincnet
    out = F.leaky_relu(s.norm1d[0](s.pool1d[0](torch.abs(first))))
    for c in (1, 2):
        out = F.leaky_relu(s.norm1d[c](s.pool1d[c](s.conv1d[c](out))))
    out = out.transpose(1, 2)
    out, _ = m.lstm(out)
    for lin in m.linear:
        out = F.leaky_relu(lin(out))
    return m.activation(m.classifier(out))


with torch.inference_mode():
    filt = sn.conv1d[0].filterbank.filters()                          # (80, 1, 251)
    ref_first = sn.conv1d[0](sn.wav_norm1d(chunks))                   # reference order
    ref = rest(ref_first, model)
    # shared: ONE convolution of the raw file, then the per-chunk affine fix-up
    S = F.conv1d(wav.unsqueeze(0), filt, stride=sn.stride)[0]        # (80, P_file)
    S1 = filt.sum(dim=(1, 2))                                          # (80,)
    P = ref_first.shape[2]
    mu = chunks.mean(dim=2, keepdim=True)
    var = chunks.var(dim=2, unbiased=False, keepdim=True)
    g = sn.wav_norm1d.weight.view(1, 1, 1) / torch.sqrt(var + sn.wav_norm1d.eps)
    b = sn.wav_norm1d.bias.view(1, 1, 1)
    sl = torch.stack([S[:, c * (STEP // sn.stride): c * (STEP // sn.stride) + P] for c in range(C)])   # (C, 80, P)
    alt_first = g * (sl - mu * S1.view(1, 80, 1)) + b * S1.view(1, 80, 1)
    alt = rest(alt_first, model)
    # float64 evaluation of the reference order: what float32 itself cannot decide
    m64 = calibrated_pyannet().eval().double()
    ref64 = rest(m64.sincnet.conv1d[0](m64.sincnet.wav_norm1d(chunks.double())), m64).float()

scale = ref_first.abs().max().item()
print(f"{C} chunks of 10 s; sinc output: max |shared - reference| = {(alt_first - ref_first).abs().max().item():.3e} "
      f"(peak {scale:.3e}, relative {(alt_first - ref_first).abs().max().item() / scale:.2e})")
for name, other in (("shared sinc (f32)", alt), ("reference order in f64", ref64)):
    d = (other - ref).abs().max().item()
    flips = (other.argmax(-1) != ref.argmax(-1))
    top2 = ref.topk(2, dim=-1).values
    gap = (top2[..., 0] - top2[..., 1])
    print(f"{name:24s}: max |dlogp| = {d:.3e}; hard decisions that differ: {int(flips.sum())} of {flips.numel()} "
          f"(of them outside a 1e-4 top-2 gap: {int((flips & (gap > 1e-4)).sum())})")
