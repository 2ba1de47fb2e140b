#!/usr/bin/env python3
"""Diagnostic: ms per 128-frame S-longdress launch with the gof's four kinds of arrays (geometry planes, attribute
planes, positions, colours) at chosen offsets (GB) inside ONE slab (VPCC_DIAG_SLAB), against the default allocation."""
import os, sys, time, zlib
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
def crc(g):
    c = 0
    for i in (0, 77, 127):
        d = g.download(i)
        c = zlib.crc32(d["xyz"].tobytes(), c); c = zlib.crc32(d["rgb"].tobytes(), c)
    return c
ref = None
for spec in sys.argv[1:]:
    if spec == "default": os.environ.pop("VPCC_DIAG_SLAB", None)
    else: os.environ["VPCC_DIAG_SLAB"] = spec
    g = ctx.gof(frames, capacity=1_000_000)
    ms = t(g); c = crc(g)
    if ref is None: ref = c
    print("%-28s %.4f ms  %s" % (spec, ms, "same output" if c == ref else "OUTPUT DIFFERS"), flush=True)
    g.close()
ctx.close()
