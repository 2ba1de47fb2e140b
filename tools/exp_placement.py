#!/usr/bin/env python3
"""Diagnostic: does the position of the gof's arena in HBM change the kernel time?  Same launches with dummy
allocations of different sizes made first (torch is only the allocator)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tmc2-rs_amd"))
import torch
from tmc2rs import recon, synth
ctx = recon.Context(0)
frames = [synth.longdress_frame(i) for i in range(32)] * 4
def measure(label):
    g = ctx.gof(frames, capacity=1_000_000)
    g.reconstruct(); g.sync()
    out = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(100):
            g.reconstruct()
        g.sync()
        out.append((time.perf_counter() - t0) / 100 * 1e3)
    import ctypes as C
    p = C.c_void_p(); 
    g.lib.vpcc_gof_device_outputs(g.h, 0, C.byref(p), None, None, None)
    print(f"{label:28s} xyz[0] at {p.value:#x}  ", " ".join("%.3f" % x for x in out))
    g.close()
measure("fresh")
keep = []
for gb in (1, 3, 7, 20, 64):
    # the arena cache of the context would hand the same arena back: allocate a blocker of a new size each time
    keep.append(torch.empty(gb << 30, dtype=torch.uint8, device="cuda:0"))
    ctx2 = recon.Context(0)          # fresh context: empty arena cache
    ctx, old = ctx2, ctx
    measure(f"behind {sum(k.numel() for k in keep) >> 30} GB of blockers")
