#!/usr/bin/env python3
"""Where a group of k_general_blocks spends its time: in-kernel stamps (s_memrealtime, 100 MHz) of thread 0 of every 397th
workgroup at the kernel's phase boundaries.  Needs a library built with the stamps: make EXTRA=-DVPCC_GEN_STAMPS (never the
product's build)."""
import ctypes as C, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tmc2-rs_amd"))
import numpy as np
from tmc2rs import recon, synth, _abi
ctx = recon.Context(0)
frames = [synth.longdress_frame(i) for i in range(32)] * 4
g = ctx.gof(frames, capacity=1_000_000, flags=_abi.VPCC_GOF_FORCE_GENERAL)
names = ["entry", "records and owners", "sample loads issued", "samples arrived", "ranks: barrier passed", "layer 0 colours", "layer 1 colours",
         "look-back done (wave 0)", "barrier passed", "staging pass 0", "staging pass 1", "", "", "", "", "end"]
SLOTS = 512
runs, acc, cnt = 10, np.zeros(16), 0
spans = []
for r in range(runs + 2):
    st = (C.c_ulonglong * (SLOTS * 16))()
    g.reconstruct(); g.sync()
    assert ctx.lib.vpcc_debug_general_stamps(st) == 0
    t = np.array(st[:], dtype=np.float64).reshape(SLOTS, 16)
    if r < 2:
        continue
    ok = (t[:, 0] > 0) & (t[:, 15] > t[:, 0])
    t = t[ok]
    for k in range(16):
        good = t[:, k] > 0
        t[~good, k] = np.nan
    acc += np.nansum((t - t[:, :1]) * 0.01, axis=0); cnt += len(t)
    spans.append(((t[:, 15] - t[:, 0]) * 0.01))
    kernel_span = (np.nanmax(t[:, 15]) - np.nanmin(t[:, 0])) * 0.01
spans = np.concatenate(spans)
print(f"k_general_blocks, thread 0 of {cnt // runs} sampled workgroups per launch, us after the workgroup's entry (mean of {runs} launches); kernel ~{kernel_span:.0f} us")
prev = 0.0
for k in range(16):
    if not names[k]:
        continue
    v = acc[k] / cnt
    print(f"  {names[k]:28s} at {v:7.2f}  (+{v - prev:6.2f})")
    prev = v
print("  lifetime of a group: median %.2f us, 10th / 90th percentile %.2f / %.2f, longest %.2f" % (np.median(spans), np.percentile(spans, 10), np.percentile(spans, 90), spans.max()))
g.close(); ctx.close()
