"""CPU numerics study: Winograd F(4x4, 3x3) instead of F(2x2, 3x3) on the stride-1 3x3 convolutions of the WeSpeaker
ResNet34.  F(4x4, 3x3) executes 36 multiplies per 16 outputs and (cin, cout)
pair instead of 64 (1.78x fewer MFMAs), but its transforms multiply by 4, 5, 8 and 1/24: float32 loses 1-2 digits.
The script swaps the selected convolutions of the oracle model for a float32 emulation of the Winograd algorithm
(U = G g G^T prepared in float64, V = B^T d B, element-wise products accumulated over the input channels and
Y = A^T M A all in float32) and reports, for the embeddings of real-looking chunks, the ratio to the north-star
tolerance |d| <= 1e-5 + 1e-4 |ref| against the float32 direct evaluation (what the GPU tests compare with).
usage: python tools/probes/winograd_f4_numerics.py [chunks]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import torch.nn as nn
import torch.nn.functional as F
from oracle.synthetic import calibrated_wespeaker, synth_conversation

torch.set_num_threads(8)
MATS = {
    2: (np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64),
        np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float64),
        np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)),
    4: (np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0],
                  [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=np.float64),
        np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
                  [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=np.float64),
        np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=np.float64)),
}


class WinoConv(nn.Module):
    """stride-1, pad-1 3x3 convolution through F(m x m, 3 x 3) in float32"""

    def __init__(self, conv: nn.Conv2d, m: int):
        super().__init__()
        BT, G, AT = MATS[m]
        self.m, self.t = m, m + 2
        U = np.einsum("ai,koij,bj->kcab".replace("c", "o"), G, conv.weight.detach().double().numpy(), G)
        self.U = torch.from_numpy(U).float()                     # (K, C, t, t)
        self.BT, self.AT = torch.from_numpy(BT).float(), torch.from_numpy(AT).float()

    def forward(self, x):
        B, C, H, W = x.shape
        m, t = self.m, self.t
        th, tw = -(-H // m), -(-W // m)
        xp = F.pad(x, (1, tw * m + 1 - W, 1, th * m + 1 - H))
        d = xp.unfold(2, t, m).unfold(3, t, m)                   # (B, C, th, tw, t, t)
        V = torch.einsum("ai,bcyxij,dj->bcyxad", self.BT, d, self.BT)
        M = torch.einsum("kcad,bcyxad->bkyxad", self.U, V)
        Y = torch.einsum("pa,bkyxad,qd->bkyxpq", self.AT, M, self.AT)
        Y = Y.permute(0, 1, 2, 4, 3, 5).reshape(B, -1, th * m, tw * m)
        return Y[:, :, :H, :W].contiguous()


def swap(model, m, layers):
    n = 0
    for li in layers:
        for blk in getattr(model.resnet, f"layer{li}"):
            for name in ("conv1", "conv2"):
                c = getattr(blk, name)
                if c.stride == (1, 1) and c.kernel_size == (3, 3):
                    setattr(blk, name, WinoConv(c, m))
                    n += 1
    return n


num = int(sys.argv[1]) if len(sys.argv) > 1 else 6
wav, _ = synth_conversation(10.0 + num, seed=5)
chunks = torch.stack([wav[:, c * 16000: c * 16000 + 160000] for c in range(num)])
g = torch.Generator().manual_seed(0)
masks = (torch.rand((num, 589), generator=g) < 0.7).float()


def ratio(got, ref):
    return ((got.double() - ref.double()).abs() / (1e-5 + 1e-4 * ref.double().abs())).max().item()


with torch.inference_mode():
    ref = calibrated_wespeaker(calib_seconds=12.0).eval()(chunks, weights=masks)
    ref64 = calibrated_wespeaker(calib_seconds=12.0).eval().double()(chunks.double(), weights=masks.double()).float()
    print(f"{num} chunks of 10 s; float32 direct vs float64 direct: ratio {ratio(ref, ref64):.3f}")
    for m, layers in ((2, (1, 2, 3, 4)), (4, (4,)), (4, (3, 4)), (4, (2, 3, 4)), (4, (1, 2, 3, 4))):
        model = calibrated_wespeaker(calib_seconds=12.0).eval()
        n = swap(model, m, layers)
        out = model(chunks, weights=masks)
        print(f"F({m}x{m},3x3) on layers {layers} ({n} convolutions): ratio to the bound vs float32 direct "
              f"{ratio(out, ref):.3f}, vs float64 {ratio(out, ref64):.3f}; max |d| {(out - ref).abs().max().item():.2e} "
              f"of max |ref| {ref.abs().max().item():.2e}", flush=True)
