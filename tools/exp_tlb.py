#!/usr/bin/env python3
"""Diagnostic: N gofs of one process, each launched a few times — run under rocprofv3 --pmc to set address-translation
counters beside the launch durations (which differ by where the gof's memory lies)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tmc2-rs_amd"))
from tmc2rs import recon, synth
frames = [synth.longdress_frame(i) for i in range(32)] * 4
ctx = recon.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
gofs = [ctx.gof(frames, capacity=1_000_000) for _ in range(n)]
for rep in range(4):
    for g in gofs:
        g.reconstruct(); g.sync()
for g in gofs: g.close()
ctx.close()
