#!/usr/bin/env python3
"""Diagnostic: the placement measurement of one tuned 128-frame gof, traced (VPCC_RUNTIME_TRACE=1), with the GPU's bus id."""
import os, sys, subprocess
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tmc2-rs_amd"))
from tmc2rs import recon, synth, _abi
frames = [synth.longdress_frame(i) for i in range(32)] * 4
ctx = recon.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
keep = []
for k in range(n):
    g = ctx.gof(frames, capacity=1_000_000, flags=_abi.VPCC_GOF_TUNE_PLACEMENT)
    g.reconstruct(); g.sync()
    print("gof", k, g.placement(), flush=True)
    keep.append(g)
for g in keep: g.close()
ctx.close()
