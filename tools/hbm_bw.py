"""Achievable HBM bandwidth of this box (development aid): device-to-device copy, read-only reduction, write-only fill
of buffers far larger than the 256-MB Infinity Cache.  usage: python tools/hbm_bw.py"""
import time, torch
dev = torch.device("cuda:0")
n = 1 << 30                      # 4 GiB of float32
a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
b = torch.empty_like(a)
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps
t = timed(lambda: b.copy_(a)); print(f"copy  4 GiB -> 4 GiB: {t*1e3:.2f} ms = {2*4*n/t/1e12:.2f} TB/s (read + write)")
t = timed(lambda: a.sum());   print(f"sum   4 GiB        : {t*1e3:.2f} ms = {4*n/t/1e12:.2f} TB/s (read)")
t = timed(lambda: b.fill_(1.0)); print(f"fill  4 GiB        : {t*1e3:.2f} ms = {4*n/t/1e12:.2f} TB/s (write)")
t = timed(lambda: torch.add(a, b, out=b)); print(f"add   2 x 4 GiB -> 4 GiB: {t*1e3:.2f} ms = {3*4*n/t/1e12:.2f} TB/s (2 reads + 1 write)")
