#!/usr/bin/env python3
"""Diagnostic: how many workgroups of one tile-kernel launch are alive over time (VPCC_TILES_VARIANT bit 8192 of
the diagnostic build records start / exit of every workgroup with the 100 MHz s_memrealtime clock)."""
import ctypes as C, os, sys
import numpy as np
os.environ["VPCC_DIAG_LIB"] = "1"
os.environ["VPCC_TILES_VARIANT"] = str(8192 | int(os.environ.get("VPCC_TILES_VARIANT", "0")))
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tmc2-rs_amd"))
from tmc2rs import recon, synth, _abi
ctx = recon.Context(0)
owlii = os.environ.get("VPCC_WORKLOAD") == "owlii"
cycles = int(os.environ.get("VPCC_CYCLES", "1"))            # 4: the 128-frame launch of bench.py (rounds)
frames = [(synth.owlii_frame if owlii else synth.longdress_frame)(i) for i in range(32)] * cycles
g = ctx.gof(frames, capacity=2_400_000 if owlii else 1_000_000)
lib = _abi.load_library()
for _ in range(3):
    g.reconstruct()
g.sync()
n = 8192
buf = (C.c_uint64 * (3 * n))()
lib.vpcc_debug_read_wg_times(buf, n)
a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 3).astype(np.int64)
a = a[a[:, 2] != 0]
t0 = a[:, 0].min()
start, steps, end = (a[:, 0] - t0) / 100.0, a[:, 1], (a[:, 2] - t0) / 100.0       # microseconds
worked = steps > 0
print(f"workgroups recorded {len(a)}, that processed groups {worked.sum()}, kernel span {end.max():.1f} us")
print(f"groups per working workgroup: mean {steps[worked].mean():.2f} min {steps[worked].min()} max {steps[worked].max()}")
bin_us = 5.0 if cycles == 1 else 20.0
edges = np.arange(0, end.max() + bin_us, bin_us)
print("time us : workgroups alive with work")
for lo in edges[:-1]:
    alive = ((start <= lo) & (end > lo) & worked).sum()
    print(f"{lo:6.0f} : {alive:5d} " + "#" * (alive // 16))
ew = np.sort(end[worked])
print("exit times of working workgroups, percentiles 0/10/50/90/100 us:", [round(float(np.percentile(ew, q)), 1) for q in (0, 10, 50, 90, 100)])
ids = np.flatnonzero(np.frombuffer(buf, dtype=np.uint64).reshape(n, 3)[:, 2] != 0)
xcd, lab = ids % 8, (ids // 8) % (4 if cycles == 1 else 8)
print("exit time (us) by groups processed:")
for k in sorted(set(steps[worked])):
    m = worked & (steps == k)
    print(f"  {k} groups: {m.sum():4d} workgroups, exit mean {end[m].mean():6.1f} min {end[m].min():6.1f} max {end[m].max():6.1f}")
print("last exit per XCD:", [round(float(end[worked & (xcd == x)].max()), 1) for x in range(8)])
print("median exit per XCD:", [round(float(np.median(end[worked & (xcd == x)])), 1) for x in range(8)])
print("last exit per frame slot / team of XCD 0:", [round(float(end[worked & (xcd == 0) & (lab == l)].max()), 1) for l in range(4 if cycles == 1 else 8)])
print("first exit per team of XCD 0:", [round(float(end[worked & (xcd == 0) & (lab == l)].min()), 1) for l in range(4 if cycles == 1 else 8)])
