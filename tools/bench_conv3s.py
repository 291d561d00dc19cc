"""Per-shape timing of the F(4x4) kernel on the maps of 3 s segments (development aid): row-shaped against
tile-linear units.  usage: PA_WINO4_LINEAR=0|1 python tools/bench_conv3s.py [B] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pyannote_audio_amd.ffi as ffi
from pyannote_audio_amd.weights import winograd4_pack, winograd4_weights

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
lib = ffi.load()
for (H, W, c, res) in [(40, 149, 64, False), (40, 149, 64, True), (20, 75, 128, False), (20, 75, 128, True),
                       (10, 38, 256, False), (10, 38, 256, True)]:
    X = torch.randn(B, H, W, c, device=dev)
    Wg = torch.randn(c, c, 3, 3) * 0.05
    U = winograd4_pack(winograd4_weights(Wg)).to(dev)
    sh = torch.randn(c, device=dev)
    R = torch.randn(B, H, W, c, device=dev) if res else None
    Y = torch.empty(B, H, W, c, device=dev)
    def run():
        ffi.check(lib.pa_conv3x3_wino4(ffi.ptr(X), B, H, W, c, ffi.ptr(U), ffi.ptr(sh), ffi.ptr(R), ffi.ptr(Y), c, 1,
                                       ffi.stream()), "wino4")
    for _ in range(10): run()
    torch.cuda.synchronize()
    ffi.prof_enable(True)
    for _ in range(reps): run()
    torch.cuda.synchronize()
    r = ffi.prof_report()["k_conv3x3_wino4"]
    ffi.prof_enable(False)
    ms = r["ms"] / r["launches"]
    print(f"wino4 {H}x{W} {c} res={int(res)} B={B}: {ms:.3f} ms  {r['flops']/r['ms']/1e9:.1f} TFLOP/s direct-form "
          f"({r['flops']/r['ms']/1e9/157.3/4*100:.1f}% of peak issued)", flush=True)
