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
names = {0: "kernel entry", 9: "descriptor in registers", 10: "occupancy loads issued", 11: "patch table in LDS", 1: "occupancy words in LDS", 2: "barrier",
         3: "cover walk", 4: "barrier", 5: "compaction walk + wave scan", 6: "barrier", 7: "items written", 8: "end"}
order = [0, 9, 10, 11, 1, 2, 3, 4, 5, 6, 7, 8]
runs = 20
acc = np.zeros((2, 16))
skew = 0.0
for r in range(runs + 3):
    g.reconstruct()
    g.sync()
    st = (C.c_ulonglong * 32)()
    assert ctx.lib.vpcc_debug_plan_stamps(st) == 0
    t = np.array(st[:32], dtype=np.float64).reshape(2, 16)
    if r >= 3:
        acc += (t - t[0, 0]) * 0.01          # us (s_memrealtime: 100 MHz)
print("k_plan_tiles, thread 0 of the first and of the last workgroup, us after the first workgroup's entry (mean of %d launches):" % runs)
for w in (0, 1):
    print(" workgroup", "0" if w == 0 else "last")
    prev = acc[w, 0] / runs
    for k in order:
        v = acc[w, k] / runs
        print(f"  {names[k]:34s} at {v:7.2f}  (+{v - prev:6.2f})")
        prev = v
g.close(); ctx.close()
