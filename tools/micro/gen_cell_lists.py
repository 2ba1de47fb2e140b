#!/usr/bin/env python3
"""Writes tools/micro/bin/cell_lists.{hdr,bin} for tools/micro/brick_layout: for 8 S-longdress frames (reconstructed by the CPU
oracle), the grid cells (grid 8, 10 bits) of every span of 1 024 points in order of first appearance — what k_smooth_stats lists."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "tmc2-rs_amd")); sys.path.insert(0, os.path.join(REPO, "tests"))
import oracle_binding as ob
from tmc2rs import synth
out, counts = [], []
for fi in range(8):
    st, ref = ob.reconstruct(synth.longdress_frame(fi))
    c = np.minimum(ob.xyz_array(ref).astype(np.int64) // 8, 127)
    key = (c[:, 2] * 128 + c[:, 1]) * 128 + c[:, 0]
    ents = []
    for s0 in range(0, len(key), 1024):
        k = key[s0:s0 + 1024]
        _, idx = np.unique(k, return_index=True)
        ents.append(k[np.sort(idx)])
    e = np.concatenate(ents).astype(np.uint32)
    counts.append(len(e)); out.append(e)
d = os.path.join(REPO, "tools", "micro", "bin")
os.makedirs(d, exist_ok=True)
np.array([len(counts)] + counts, dtype=np.uint32).tofile(os.path.join(d, "cell_lists.hdr"))
np.concatenate(out).tofile(os.path.join(d, "cell_lists.bin"))
print(counts)
