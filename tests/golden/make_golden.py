#!/usr/bin/env python3
"""Generates the golden fixtures of tests/golden/: inputs (decoded planes + patch tables) and the outputs
the CPU oracle (oracle/vpcc_oracle.c) produces for them.  The reference ships no fixtures for this path
and cannot be run here (Rust + libavcodec absent), so these vectors freeze the ORACLE's behaviour —
itself pinned by tests/test_oracle_kat.py and tests/pyref.py — across rounds.

    python tests/golden/make_golden.py        # rewrites tests/golden/*.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "tmc2-rs_amd"))
sys.path.insert(0, os.path.join(HERE, ".."))

import cases            # noqa: E402
import oracle_binding as ob   # noqa: E402

GOLDEN = ["small0", "small2_wide", "exotic_orientations", "relative_d1", "truncation_degenerate_axes",
          "block8_ragged", "gray_exact_boundaries", "single_map_extension"]


def pack(frame, ref):
    d = {k: np.asarray(frame[k]) for k in ("width", "height", "occupancy_resolution", "occupancy_precision")}
    for k in ("map_count", "absolute_d1", "attribute_count"):
        d[k] = np.asarray(frame.get(k, {"map_count": 2, "absolute_d1": 1, "attribute_count": 1}[k]))
    d["patches"] = np.asarray(frame["patches"])
    d["occupancy"] = np.ascontiguousarray(frame["occupancy"])
    for m, g in enumerate(frame["geometry"]):
        d[f"geometry{m}"] = np.ascontiguousarray(g)
    for m, a in enumerate(frame["attribute"]):
        for c, p in zip("yuv", a):
            d[f"attribute{m}{c}"] = np.ascontiguousarray(p)
    d["out_xyz"] = ob.xyz_array(ref)
    d["out_rgb"] = ob.rgb_array(ref)
    d["out_partition"] = ref["partition"].astype(np.uint32)
    d["out_block_to_patch"] = ref["block_to_patch"].astype(np.uint32)
    return d


def unpack(z):
    f = {k: int(z[k]) for k in ("width", "height", "occupancy_resolution", "occupancy_precision", "map_count",
                                "absolute_d1", "attribute_count")}
    f["flags"] = 0
    f["patches"] = z["patches"]
    f["occupancy"] = z["occupancy"]
    f["geometry"] = [z[f"geometry{m}"] for m in range(2) if f"geometry{m}" in z]
    f["attribute"] = [tuple(z[f"attribute{m}{c}"] for c in "yuv") for m in range(2) if f"attribute{m}y" in z]
    return f


if __name__ == "__main__":
    for name in GOLDEN:
        frame = cases.PARITY_CASES[name]()
        st, ref = ob.reconstruct(frame)
        assert st == 0
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **pack(frame, ref))
        print(name, ref["n"], "points")
