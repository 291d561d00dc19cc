"""Cycles per v_pk_fma_f32 by operand kind (development probe; build: tools/probes/build.sh)."""
import ctypes as C, os
import torch
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "_dbg", "libprobes.so"))
dev = torch.device("cuda:0")
src = torch.randn(1 << 16, device=dev)
out = torch.zeros(256 * 4, dtype=torch.int64, device=dev)
sink = torch.zeros(16, device=dev)
names = ["v_pk_fma_f32 v, v, v, v", "v_pk_fma_f32 v, v, s[pair], v", "v_pk_add_f32 v, v, v",
         "2 x v_fma_f32 v, v, s, v (unpacked)", "v_pk_mul_f32 v, v, s[pair]"]
for mode, name in enumerate(names):
    for _ in range(2):
        lib.pkfma_probe(C.c_void_p(src.data_ptr()), 2000, mode, C.c_void_p(out.data_ptr()), C.c_void_p(sink.data_ptr()),
                        256, None)
        torch.cuda.synchronize()
    cyc = out.double().mean().item() / 2000 / 32
    print(f"{name:40s} {cyc:6.2f} cycles per (packed) value pair, one wave per SIMD", flush=True)
