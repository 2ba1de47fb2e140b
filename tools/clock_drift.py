#!/usr/bin/env python3
"""Does the kernel time drift over a long run (clock / power state)?  Prints ms per 128-frame launch for consecutive
chunks of launches.  usage: tools/clock_drift.py [seconds=6]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tmc2-rs_amd"))
from tmc2rs import recon, synth
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
ctx = recon.Context(0)
frames = [synth.longdress_frame(i) for i in range(32)]
g = ctx.gof(frames * 4, capacity=1_000_000)
g.reconstruct(); g.sync()
t_end = time.perf_counter() + secs
out = []
while time.perf_counter() < t_end:
    t0 = time.perf_counter()
    for _ in range(100):
        g.reconstruct()
    g.sync()
    out.append((time.perf_counter() - t0) / 100 * 1e3)
print(" ".join("%.3f" % x for x in out))
g.close(); ctx.close()
