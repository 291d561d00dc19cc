"""k_stem A/B (development aid): time and a checksum of the output for the library selected by PA_LIB.
usage: [PA_LIB=variant.so] python tools/stem_ab.py [B] [T]"""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pyannote_audio_amd.ffi as ffi
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
T = int(sys.argv[2]) if len(sys.argv) > 2 else 998
dev = torch.device("cuda:0")
lib = ffi.load()
g = torch.Generator().manual_seed(3)
fb = torch.randn(B, T, 80, generator=g).to(dev)
w9 = (torch.randn(9, 32, generator=g) * 0.3).to(dev)
sh = torch.randn(32, generator=g).to(dev)
out = torch.empty(B, 80, T, 32, device=dev)
def run():
    ffi.check(lib.pa_resnet_stem(ffi.ptr(fb), B, T, 80, ffi.ptr(w9), ffi.ptr(sh), ffi.ptr(out), ffi.stream()), "stem")
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
digest = hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:16]
print(f"PA_LIB={os.environ.get('PA_LIB', '(product)')}: B={B} T={T}: {ms:.3f} ms per launch = {out.numel() * 4 / ms / 1e9:.2f} TB/s written; sha1 {digest}")
