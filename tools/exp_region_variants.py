#!/usr/bin/env python3
"""Diagnostic (diagnostic library): which part of the tile kernel is sensitive to where the arena lies in VRAM?
Several arenas in one process, each timed under the ablation variants (32 no stores, 256 no attribute loads, 512 no
geometry loads, 1 no look-back wait)."""
import os, sys, time, ctypes
os.environ["VPCC_DIAG_LIB"] = "1"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tmc2-rs_amd"))
from tmc2rs import recon, synth
libc = ctypes.CDLL(None)
def setenv(k, v): libc.setenv(k.encode(), v.encode(), 1); os.environ[k] = v
frames = [synth.longdress_frame(i) for i in range(32)] * 4
ctx = recon.Context(0)
def t(g):
    for _ in range(30): g.reconstruct()
    g.sync()
    t0 = time.perf_counter()
    for _ in range(100): g.reconstruct()
    g.sync()
    return (time.perf_counter() - t0) / 100 * 1e3
gofs = [ctx.gof(frames, capacity=1_000_000) for _ in range(8)]
for g in gofs: g.reconstruct(); g.sync()
for v in (0, 32, 256, 512, 1, 0):
    setenv("VPCC_TILES_VARIANT", str(v))
    print(f"variant {v:4d}: " + " ".join("%.3f" % t(g) for g in gofs), flush=True)
for g in gofs: g.close()
ctx.close()
