#!/usr/bin/env python3
"""Diagnostic: ms per 128-frame launch of N arenas allocated one after the other in one process (library chosen with
VPCC_DIAG_LIB as for tools/ab_multi.sh)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tmc2-rs_amd"))
from tmc2rs import recon, synth
frames = [synth.longdress_frame(i) for i in range(32)] * 4
ctx = recon.Context(0)
def t(g):
    for _ in range(30): g.reconstruct()
    g.sync()
    t0 = time.perf_counter()
    for _ in range(100): g.reconstruct()
    g.sync()
    return (time.perf_counter() - t0) / 100 * 1e3
gofs = [ctx.gof(frames, capacity=1_000_000) for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8)]
for g in gofs: g.reconstruct(); g.sync()
for rep in range(2):
    print(os.environ.get("VPCC_DIAG_LIB", "product"), " ".join("%.3f" % t(g) for g in gofs), flush=True)
for g in gofs: g.close()
ctx.close()
