"""Per-shape timing of pa_conv3x3 on the ResNet34 layer shapes (development aid; HIP events through
the library profiler).  usage: [WINO=1|4] [ONLY_S1=1] [PA_LIB=variant.so] python tools/bench_conv.py [B] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pyannote_audio_amd.ffi as ffi

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
lib = ffi.load()
T = 998
shapes = [  # (H, W, cin, cout, stride, residual)
    (80, T, 32, 32, 1, True), (80, T, 32, 64, 2, False), (40, 499, 64, 64, 1, True),
    (40, 499, 64, 128, 2, False), (20, 250, 128, 128, 1, True), (20, 250, 128, 256, 2, False),
    (10, 125, 256, 256, 1, True)]
if os.environ.get("ONLY_S1") == "1":   # the stride-1 (Winograd) layers, with and without the residual input
    shapes = [(H, W, ci, co, s, r) for (H, W, ci, co, s, _) in shapes if s == 1 for r in (False, True)]
if os.environ.get("ONLY_CIN"):         # one input width only (e.g. ONLY_CIN=32: layer 1)
    shapes = [sh for sh in shapes if sh[2] == int(os.environ["ONLY_CIN"])]
for (H, W, ci, co, s, res) in shapes:
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    X = torch.randn(B, H, W, ci, device=dev)
    Wg = torch.randn(9, co, ci, device=dev) * 0.05
    sh = torch.randn(co, device=dev)
    R = torch.randn(B, Ho, Wo, co, device=dev) if res else None
    Y = torch.empty(B, Ho, Wo, co, device=dev)
    from pyannote_audio_amd.weights import winograd_pack, winograd_weights, winograd4_pack, winograd4_weights
    wino = os.environ.get("WINO", "0") == "1" and s == 1
    wino4 = os.environ.get("WINO", "0") == "4" and s == 1          # Winograd F(4x4,3x3), csrc/emb_winograd4.hip
    Ug = winograd_pack(winograd_weights(Wg.permute(1, 2, 0).reshape(co, ci, 3, 3).cpu())).to(dev) if wino else None
    if wino4:
        Ug = winograd4_pack(winograd4_weights(Wg.permute(1, 2, 0).reshape(co, ci, 3, 3).cpu())).to(dev)
    def run():
        if wino4:
            ffi.check(lib.pa_conv3x3_wino4(ffi.ptr(X), B, H, W, ci, ffi.ptr(Ug), ffi.ptr(sh), ffi.ptr(R), ffi.ptr(Y),
                                           co, 1, ffi.stream()), "wino4")
            return
        if wino:
            ffi.check(lib.pa_conv3x3_wino(ffi.ptr(X), B, H, W, ci, ffi.ptr(Ug), ffi.ptr(sh), ffi.ptr(R), ffi.ptr(Y),
                                          co, 1, ffi.stream()), "wino")
            return
        ffi.check(lib.pa_conv3x3(ffi.ptr(X), B, H, W, ci, ffi.ptr(Wg), ffi.ptr(sh), ffi.ptr(R), ffi.ptr(Y),
                                 co, s, 1, ffi.stream()), "conv")
    for _ in range(int(os.environ.get('WARM', '10'))): run()
    torch.cuda.synchronize()
    ffi.prof_enable(True)
    for _ in range(reps): run()
    torch.cuda.synchronize()
    rep = ffi.prof_report()
    r = rep.get("k_conv3x3_wino4") or rep.get("k_conv3x3_wino") or rep["k_conv3x3"]
    ffi.prof_enable(False)
    ms = r["ms"] / r["launches"]
    print(f"conv {H}x{W} {ci}->{co} s{s} res={int(res)}: {ms:.3f} ms  {r['flops']/r['ms']/1e9:.1f} TFLOP/s "
          f"({r['flops']/r['ms']/1e9/157.3*100:.1f}% of f32 MFMA peak)", flush=True)
