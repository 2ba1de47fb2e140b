#!/usr/bin/env python3
"""Host<->device copy rates on this box: hipHostMalloc'd (torch pinned) vs hipHostRegister'd pageable memory."""
import ctypes as C, time, torch, numpy as np
n = 1 << 30
dev = torch.empty(n, dtype=torch.uint8, device="cuda")
pin = torch.empty(n, dtype=torch.uint8).pin_memory()
def rate(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return reps * n / (time.perf_counter() - t) / 1e9
print("H2D pinned (hipHostMalloc) GB/s", round(rate(lambda: dev.copy_(pin, non_blocking=True)), 1))
print("D2H pinned GB/s", round(rate(lambda: pin.copy_(dev, non_blocking=True)), 1))
pg = torch.from_numpy(np.ones(n, dtype=np.uint8))
t = time.perf_counter(); torch.cuda.cudart().cudaHostRegister(pg.data_ptr(), n, 0); print("hipHostRegister 1 GiB s", round(time.perf_counter() - t, 3))
print("H2D registered GB/s", round(rate(lambda: dev.copy_(pg, non_blocking=True)), 1))
s2 = torch.cuda.Stream()
def both():
    dev[: n // 2].copy_(pin[: n // 2], non_blocking=True)
    with torch.cuda.stream(s2):
        pin[n // 2:].copy_(dev[n // 2:], non_blocking=True)
r = rate(both)
print("H2D + D2H concurrently, GB/s each way", round(r / 2, 1))
