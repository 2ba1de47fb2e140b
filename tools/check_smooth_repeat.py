#!/usr/bin/env python3
"""Diagnostic: repeated reconstruct + smooth on one gof, alternating between "both filters in one pass" (grid 8 / 8)
and separate passes (4 / 16), with per-iteration counts of points that differ from the specification — what
tests/test_smoothing.py::test_repeated_smoothing_launches_leave_the_grid_clean asserts, in numbers."""
import sys, os
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/tmc2-rs_amd")
import numpy as np, cases, oracle_binding as ob
from tmc2rs import recon, _abi
fr = [cases.overlapping_3d_frame(i) for i in range(2)]
ctx = recon.Context(0)
g = ctx.gof(fr, flags=_abi.VPCC_GOF_WANT_PATCH_INDEX)
for it, (grid, cgrid) in enumerate(((8, 8), (4, 16), (8, 8), (8, 8), (8, 8))):
    g.reconstruct()
    before = [g.download(i, want_patch_index=True) for i in range(len(fr))]
    g.smooth(10, grid_size=grid, threshold=1, color_grid_size=cgrid, color_threshold_smoothing=5, color_threshold_difference=80)
    for i, b in enumerate(before):
        after = g.download(i)
        exp_xyz = ob.spec_smooth_geometry(b["xyz"], b["patch_index"], 10, grid, 1)
        exp_rgb = ob.spec_smooth_color(exp_xyz, b["rgb"], b["patch_index"], 10, cgrid, 5, 80)
        bx = np.any(after["xyz"] != exp_xyz, axis=1); br = np.any(after["rgb"] != exp_rgb, axis=1)
        moved = np.any(exp_xyz != b["xyz"], axis=1)
        print("iter", it, (grid, cgrid), "frame", i, "points", len(bx), "moved", int(moved.sum()), "xyz wrong", int(bx.sum()), "rgb wrong", int(br.sum()),
              "rgb wrong among moved", int((br & moved).sum()), flush=True)
        if br.sum():
            j = np.flatnonzero(br)[:5]
            print("   e.g.", j, after["rgb"][j].tolist(), exp_rgb[j].tolist(), b["rgb"][j].tolist(), exp_xyz[j].tolist())
g.close(); ctx.close()
