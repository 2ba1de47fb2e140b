#!/usr/bin/env python3
"""Diagnostic: is the slow mode of the 128-frame launch an artefact of the batch being four identical copies of 32
frames (identical work at addresses exactly 32 frame-slices apart)?  4 x 32 against 128 distinct frames, several
fresh arenas each."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tmc2-rs_amd"))
from tmc2rs import recon, synth
from concurrent.futures import ThreadPoolExecutor
with ThreadPoolExecutor(16) as ex:
    distinct = list(ex.map(synth.longdress_frame, range(128)))
copies = distinct[:32] * 4
def measure(frames):
    ctx = recon.Context(0)
    g = ctx.gof(frames, capacity=1_000_000)
    g.reconstruct(); g.sync()
    pts = int(g.point_counts().sum())
    for _ in range(40): g.reconstruct()
    g.sync()
    t0 = time.perf_counter()
    for _ in range(150): g.reconstruct()
    g.sync()
    dt = (time.perf_counter() - t0) / 150 * 1e3
    return dt, pts, g, ctx
keep = []
for rep in range(5):
    a = measure(copies); b = measure(distinct)
    keep += [a, b]                                   # arenas stay allocated: every pair lands somewhere else
    print(f"4 x 32 copies: {a[0]:.3f} ms ({a[1]/a[0]/1e6:.1f} Gpoints/s)    128 distinct: {b[0]:.3f} ms ({b[1]/b[0]/1e6:.1f} Gpoints/s)", flush=True)
