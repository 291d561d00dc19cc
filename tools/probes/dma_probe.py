"""Sustained LDS-DMA rate per CU vs request shape / issuing waves / footprint (development probe).
build: tools/probes/build.sh ; run on the GPU box: python tools/probes/dma_probe.py"""
import ctypes as C, os, sys
import torch
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "_dbg", "libprobes.so"))
lib.dma_probe.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
dev = torch.device("cuda:0")
big = torch.randn(1 << 29, device=dev)          # 2 GiB
out = torch.zeros(256 * 8, dtype=torch.int64, device=dev)
names = {0: "linear 1KB", 1: "64B segs", 2: "128B segs", 3: "256B segs"}
def run(foot_mb, mode, stride, nwaves, pieces, iters=400, grid=256):
    nbytes = foot_mb << 20
    for _ in range(2):
        out.zero_()
        rc = lib.dma_probe(C.c_void_p(big.data_ptr()), nbytes, mode, stride, nwaves, pieces, iters,
                           C.c_void_p(out.data_ptr()), grid, None)
        torch.cuda.synchronize()
    cyc = out.view(256, 8)[:grid, :nwaves].double().mean().item()
    total = nwaves * pieces * iters * 1024
    print(f"foot {foot_mb:5d} MB  {names[mode]:10s} stride {stride:5d}  waves {nwaves}  pieces/iter {pieces:2d}: "
          f"{total / cyc:6.2f} B/clk/CU  ({cyc / (pieces * iters):6.0f} clk per piece per wave)", flush=True)
for foot in (16, 128, 2048):
    for mode, stride in ((0, 0), (1, 128), (1, 256), (1, 1024), (2, 128), (2, 1024), (3, 1024)):
        for nw, pc in ((4, 10), (8, 5), (8, 10)):
            run(foot, mode, stride, nw, pc)
# a single CU alone (no contention from the other 255)
for mode, stride in ((0, 0), (1, 1024), (2, 1024)):
    run(16, mode, stride, 4, 10, grid=1)
    run(16, mode, stride, 8, 10, grid=1)
