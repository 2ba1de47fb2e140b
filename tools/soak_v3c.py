#!/usr/bin/env python3
"""Soak of the V3C path: random sequences are WRITTEN as V3C sample streams by the independent Python writer (tests/v3c_writer.py) together
with their "decoded" raw videos, read back by the C++ syntax parser + patch-table builder, streamed through the Decoder (1 .. 3 lanes) and
compared with the oracle on the frames the stream was written from.  Canvas sizes, block sizes 8 / 16 / 32, occupancy precisions,
one or two maps, with or without attribute, geometry-smoothing SEIs on some GOFs (applied, and checked against the specification).
Usage: tools/soak_v3c.py [streams = 100] [first seed = 0]"""
import os, sys, shutil, tempfile, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tmc2-rs_amd")); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
from tmc2rs import recon, synth
import oracle_binding as ob
import v3c_writer as W
n_streams = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(0x73C + seed0)
bad = frames_done = 0
t0 = time.time()
d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
try:
    for si in range(n_streams):
        R = int(rng.choice([16, 16, 16, 8, 32])); prec = int(rng.choice([1, 2, 4, 4]))
        unit = max(R, 2 * prec, 4)
        w = unit * int(rng.integers(2, max(3, 400 // unit))); h = unit * int(rng.integers(2, max(3, 300 // unit)))
        maps = int(rng.choice([2, 2, 2, 1])); attr = int(rng.choice([1, 1, 1, 0]))
        gofs, seis = [], []
        for g in range(int(rng.integers(1, 6))):
            fr = []
            for k in range(int(rng.integers(1, 7))):
                f = synth.make_frame(w, h, prec, R, seed=0x73C00000 + seed0 * 7919 + si * 64 + g * 8 + k, max_side=int(rng.integers(1, max(2, 96 // R + 1))),
                                     cover_target=float(rng.uniform(0.2, 0.9)), swap_prob=float(rng.uniform(0, 1)), dup_prob=float(rng.uniform(0, 0.6)),
                                     occupancy_values="random" if k % 2 else "one")
                p = f["patches"].copy()
                if (si + g) % 2:                                  # patches that overlap in 3-D: work for the geometry filter
                    p["u1"] = 100 + (np.arange(len(p)) % 5) * 3; p["v1"] = 100 + (np.arange(len(p)) % 7) * 2
                    p["d1"] = np.where(p["projection_mode"] == 0, 100, 300)
                f["patches"] = p
                f["map_count"] = maps; f["attribute_count"] = attr
                fr.append(f)
            gofs.append(fr)
            seis.append((int(rng.choice([4, 8, 16])), int(rng.choice([0, 2, 4]))) if rng.random() < 0.4 else None)
        smooth = rng.random() < 0.5                              # apply_geo_smoothing_type: the SEI's GOFs are smoothed
        paths = W.write_sequence(d, gofs, seis=seis)
        dec = recon.Decoder(paths["bin"], devices=(0,) * int(rng.choice([1, 1, 2, 3])), occupancy_yuv=paths["occ"], geometry_yuv=paths["geo"],
                            attribute_yuv=paths["attr"], occupancy_precision=prec)
        if smooth:
            dec.set_smoothing(geometry=True)
        dec.start()
        got = list(dec)
        err = dec.error()
        dec.close()
        exp = [(f, seis[gi] if smooth else None) for gi, g in enumerate(gofs) for f in g]
        ok = err == "" and len(got) == len(exp)
        for fr, (f, sei) in zip(got, exp) if ok else []:
            st, ref = ob.reconstruct(f)
            xyz = ob.xyz_array(ref)
            if sei is not None:
                xyz = ob.spec_smooth_geometry(xyz, ref["partition"].astype(np.uint16), 10, sei[0], sei[1])
            ok = ok and st == 0 and fr["n"] == ref["n"] and np.array_equal(fr["xyz"], xyz) and (not attr or np.array_equal(fr["rgb"], ob.rgb_array(ref)))
        frames_done += len(got)
        if not ok:
            bad += 1
            print(f"MISMATCH stream {si}: {w}x{h} R={R} prec={prec} maps={maps} attr={attr}, GOFs {[len(g) for g in gofs]}, SEIs {seis}, smoothing {smooth}: "
                  f"error {err!r}, {len(got)} of {len(exp)} frames", flush=True)
        if si % 10 == 9:
            print(f"{si + 1} streams, {frames_done} frames, {bad} bad, {time.time() - t0:.0f} s", flush=True)
finally:
    shutil.rmtree(d, ignore_errors=True)
print(f"soak: {n_streams} V3C streams, {frames_done} frames, {bad} bad")
sys.exit(1 if bad else 0)
