#!/usr/bin/env python3
"""Where k_plan_tiles' time goes: in-kernel stamps (s_memtime, 100 MHz) of workgroup 0 / thread 0 at the kernel's phase
boundaries.  Needs a library built with the stamps: make EXTRA=-DVPCC_PLAN_STAMPS (never the product's build)."""
import ctypes as C, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tmc2-rs_amd"))
import numpy as np
from tmc2rs import recon, synth
ctx = recon.Context(0)
frames = [synth.longdress_frame(i) for i in range(32)] * 4
g = ctx.gof(frames, capacity=1_000_000)
names = ["descriptor", "occupancy + patch table loaded", "barrier", "cover walk", "barrier", "count walk + wave scan", "barrier", "items written", "block_to_patch stored"]
acc = np.zeros(9)
runs = 20
for r in range(runs + 3):
    g.reconstruct()
    g.sync()
    st = (C.c_ulonglong * 16)()
    assert ctx.lib.vpcc_debug_plan_stamps(st) == 0
    t = np.array(st[:9], dtype=np.float64)
    if r >= 3:
        acc += (t - t[0]) * 0.01          # us
print("k_plan_tiles, workgroup 0 / thread 0, us after kernel entry (mean of %d launches):" % runs)
prev = 0.0
for n, v in zip(names, acc / runs):
    print(f"  {n:34s} at {v:7.2f}  (+{v - prev:6.2f})")
    prev = v
g.close(); ctx.close()
