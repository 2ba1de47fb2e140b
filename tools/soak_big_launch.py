#!/usr/bin/env python3
"""ONE launch over many DISTINCT full-size frames (1280 x 1408, patch statistics varied per frame: cover 0.25 .. 0.6, patch sides up to 8 .. 40
blocks, Swap share 0 .. 1) — rounds of eight frames per XCD label, workgroups that help other frames, unequal frames — every frame against
the oracle.  Usage: tools/soak_big_launch.py [frames = 200] [seed = 0]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tmc2-rs_amd")); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
from tmc2rs import _abi, recon, synth
import oracle_binding as ob
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(0xB16 + seed)
t0 = time.time()
frames = [synth.make_frame(1280, 1408, 4, 16, seed=0xB1600000 + seed * 4099 + i, coord_bits=10, cover_target=float(rng.uniform(0.25, 0.6)),
                           max_side=int(rng.integers(8, 41)), swap_prob=float(rng.uniform(0, 1)), size_skew=float(rng.uniform(1, 8))) for i in range(n)]
print(f"{n} frames synthesised in {time.time() - t0:.0f} s", flush=True)
ctx = recon.Context(0)
ctx.reserve(32)
g = ctx.gof(frames, capacity=1_400_000, flags=_abi.VPCC_GOF_PROFILE)
for rep in range(3):                                   # the same launch three times: generation-tagged counters, nothing cleared in between
    g.reconstruct()
counts = g.point_counts()
print(f"launch: {g.kernel_times()}, points per frame {int(counts.min())} .. {int(counts.max())}", flush=True)
bad = 0
for i, f in enumerate(frames):
    st, ref = ob.reconstruct(f)
    got = g.download(i)
    ok = st == 0 and got["n"] == ref["n"] and np.array_equal(got["xyz"], ob.xyz_array(ref)) and np.array_equal(got["rgb"], ob.rgb_array(ref))
    if not ok:
        bad += 1
        print(f"MISMATCH frame {i}: {ref['n']} vs {got['n']} points", flush=True)
    if i % 50 == 49:
        print(f"{i + 1} frames checked, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print(f"soak: one launch over {n} distinct full-size frames, {bad} mismatches")
sys.exit(1 if bad else 0)
