"""DMA cost beside an MFMA run: bunched vs spread, linear vs strided segments (tools/probes/interleave_probe.hip).
usage: sh tools/probes/build.sh && python tools/probes/interleave_probe.py"""
import ctypes as C, os
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "..", "_dbg", "libprobes.so"))
lib.interleave_probe.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                 C.c_void_p, C.c_int, C.c_void_p]
dev = torch.device("cuda:0")
src = torch.randn(512 * 1024 * 1024 // 4, device=dev)       # 512 MB
sink = torch.zeros(4096, device=dev)
iters = 200
st = torch.cuda.current_stream().cuda_stream
print("cycles per iteration (128 MFMAs = 4096 ideal) per wave; per_cu workgroups of 4 waves on every CU")
for per_cu in (1, 2):
    grid = 256 * per_cu
    out = torch.zeros(grid * 4, dtype=torch.int64, device=dev)
    for pattern, span, label in [(0, 32 * 1024, "linear 1 KB, 32-KB span (L2)"),
                                 (128, 4 << 20, "64-B segs / 128 B, 4-MB span"),
                                 (512, 4 << 20, "64-B segs / 512 B, 4-MB span"),
                                 (1024, 4 << 20, "64-B segs / 1 KB, 4-MB span"),
                                 (128, 8 << 20, "64-B segs / 128 B, 8 MB x 64 = HBM")]:
        for d in (0, 8, 16):
            row = []
            for spread in (0, 1):
                for _ in range(2):
                    rc = lib.interleave_probe(src.data_ptr(), iters, d, spread, pattern, span, per_cu, out.data_ptr(),
                                              sink.data_ptr(), grid, st)
                    torch.cuda.synchronize()
                assert rc == 0
                row.append(out.double().mean().item() / iters)
            print(f"per_cu={per_cu} {label:38s} D={d:2d}: bunched {row[0]:7.0f}  spread {row[1]:7.0f}", flush=True)
            if d == 0 and pattern != 0:
                pass
