"""Does a store cost its wave less when the other waves of the CU compute instead of storing?  (development probe;
build: tools/probes/build.sh)"""
import ctypes as C, os
import torch
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "_dbg", "libprobes.so"))
dev = torch.device("cuda:0")
NS, row_stride, cus = 32, 131072, 256
out = torch.zeros(cus * 4 * 2, dtype=torch.int64, device=dev)
sink = torch.zeros(1024, device=dev)
for seg_stride in (1024, 2048, 4096):
    wave_bytes = NS * row_stride + 16 * (seg_stride + 4160) + 4096
    buf = torch.empty(wave_bytes * cus * 4 // 4, dtype=torch.float32, device=dev)
    for pattern, pname in ((0, "16 x 64 B"), (1, "8 x 128 B"), (2, "dword: 2 x 128 B"), (3, "stem: 64 x 16 B @32")):
        for mask, name in ((15, "all four waves store"), (1, "wave 0 stores, 3 waves run MFMAs"), (3, "waves 0-1 store, 2 run MFMAs"),
                           (5, "waves 0 and 2 store, 2 run MFMAs")):
            for rep in range(3):
                rc = lib.store_stagger_probe(C.c_void_p(buf.data_ptr()), C.c_long(wave_bytes), mask, seg_stride, row_stride, 400,
                                             pattern, C.c_void_p(out.data_ptr()), C.c_void_p(sink.data_ptr()), cus, None)
                torch.cuda.synchronize()
                assert rc == 0
            o = out.view(cus, 4, 2).double()
            st = [w for w in range(4) if (mask >> w) & 1]
            mf = [w for w in range(4) if not (mask >> w) & 1]
            issue = o[:, st, 0].mean().item() / NS
            drain = o[:, st, 1].mean().item() / NS
            line = f"pixel stride {seg_stride:4d} B, {pname:9s} {name:36s} {issue:6.0f} cycles per store to issue, {drain:6.0f} with the drain"
            if mf:
                line += f";   MFMA waves: {o[:, mf, 0].mean().item() / (400 * 16):5.1f} cycles per MFMA"
            print(line, flush=True)
