#!/usr/bin/env python3
"""How representative is a bench step (the SAME 32-frame GOF reconstructed again and again, 0.8 GB of planes and
outputs) of a stream of different GOFs?  Times 32-frame launches that alternate between N resident GOFs (N x 0.8 GB
touched in turn) against launches on one GOF.  usage: tools/two_gofs.py [N=2]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tmc2-rs_amd"))
from tmc2rs import recon, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ctx = recon.Context(0)
gofs = [ctx.gof([synth.longdress_frame(32 * k + i) for i in range(32)], capacity=1_000_000) for k in range(n)]
for g in gofs:
    g.reconstruct(); g.sync()


def run(seq, reps=300):
    for g in seq[:8]:
        g.reconstruct()
    seq[0].sync()
    for g in gofs:
        g.sync()
    t0 = time.perf_counter()
    for r in range(reps):
        seq[r % len(seq)].reconstruct()
    for g in gofs:
        g.sync()
    return (time.perf_counter() - t0) / reps * 1e3


for trial in range(2):
    print("one GOF repeated      : %.4f ms per launch" % run([gofs[0]]))
    print("%d GOFs in turn        : %.4f ms per launch" % (n, run(gofs)))
for g in gofs:
    g.close()
ctx.close()
