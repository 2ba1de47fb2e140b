#!/usr/bin/env python3
"""Diagnostic: 128-frame launches on ONE context back to back vs on TWO contexts (two streams of one GPU) at the same
time.  If two launches in flight give more points per second than one, a launch leaves part of the chip idle (ramp,
drain, lock-step phases) that independent work can fill."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tmc2-rs_amd"))
from tmc2rs import recon, synth
frames = [synth.longdress_frame(i) for i in range(32)] * 4
ctxs = [recon.Context(0), recon.Context(0)]
gofs = [c.gof(frames, capacity=1_000_000) for c in ctxs]
for g in gofs:
    g.reconstruct(); g.sync()
pts = int(gofs[0].point_counts().sum())

def run(gs, launches):
    for _ in range(20):
        for g in gs: g.reconstruct()
    for g in gs: g.sync()
    t = time.perf_counter()
    for _ in range(launches):
        for g in gs: g.reconstruct()
    for g in gs: g.sync()
    dt = time.perf_counter() - t
    return dt / (launches * len(gs))

for rep in range(3):
    a = run(gofs[:1], 400)
    b = run(gofs, 200)
    print(f"one stream: {a*1e3:.4f} ms per 128-frame launch ({pts/a/1e9:.1f} Gpoints/s)   two streams: {b*1e3:.4f} ms per launch ({pts/b/1e9:.1f} Gpoints/s)")
