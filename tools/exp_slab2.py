#!/usr/bin/env python3
"""Diagnostic: inside ONE 130-GB slab (VPCC_DIAG_SLAB), (1) where along the slab does the output have to lie for the launch to
be fast, with the planes at its start -> the 32-GB chunk boundaries in slab coordinates; (2) every assignment of
{attribute planes, positions, colours} to the chunk parity of the geometry planes (a) or the other one (b)."""
import os, sys, time, itertools
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tmc2-rs_amd"))
os.environ["VPCC_DIAG_SLAB_GB"] = "130"
from tmc2rs import recon, synth
frames = [synth.longdress_frame(i) for i in range(32)] * 4
ctx = recon.Context(0)
def run(spec):
    os.environ["VPCC_DIAG_SLAB"] = ",".join("%g" % x for x in spec)
    g = ctx.gof(frames, capacity=1_000_000)
    for _ in range(20): g.reconstruct()
    g.sync()
    t0 = time.perf_counter()
    for _ in range(60): g.reconstruct()
    g.sync()
    ms = (time.perf_counter() - t0) / 60 * 1e3
    g.close()
    return ms
sweep = []
for x in range(4, 128, 4):
    ms = run((0, 1, x, x + 1.2)); sweep.append((x, ms))
print("output at x GB (planes at 0):", " ".join("%d:%.3f" % s for s in sweep), flush=True)
lo = min(m for _, m in sweep); hi = max(m for _, m in sweep); mid = (lo + hi) / 2
fast = [x for x, m in sweep if m < mid]
b0 = fast[0]                                   # first offset in the other parity
print("fast from", b0, "GB on; levels %.3f / %.3f" % (lo, hi), flush=True)
# slots: parity a = the chunk of offset 0 (if b0 >= 8) else the chunk after the first b chunk
A = 0 if b0 >= 8 else b0 + 32 + 1
B = b0 + 1
sub = {"geo": 0.0, "attr": 1.0, "xyz": 2.5, "rgb": 3.5}
for pa, px, pr in itertools.product("ab", repeat=3):
    pos = lambda kind, par: (A if par == "a" else B) + sub[kind]
    spec = (pos("geo", "a"), pos("attr", pa), pos("xyz", px), pos("rgb", pr))
    print("geo a, attr %s, xyz %s, rgb %s : %.4f ms" % (pa, px, pr, run(spec)), flush=True)
# three chunks: a, b, and the next a chunk (A2)
A2 = B - 1 + 32 + 1
for name, spec in (("geo a0, attr b, xyz a1, rgb a1", (A, B + 1, A2 + 2.5, A2 + 3.5)),
                   ("geo a0, attr a1, xyz b, rgb b", (A, A2 + 1, B + 2.5, B + 3.5)),
                   ("geo a0, attr b, xyz a1, rgb b", (A, B + 1, A2 + 2.5, B + 3.5))):
    print(name, ": %.4f ms" % run(spec), flush=True)
ctx.close()
