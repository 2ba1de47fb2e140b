#!/usr/bin/env python3
"""The largest canvas the ABI accepts (32768 x 32768, two maps): one synthetic frame through the HIP path and the CPU oracle, compared
point for point.  More than 2^32 / 6 points, so the byte offsets into the position array pass 4 GiB; planes of 2 GiB each.
Needs about 60 GB of host memory and a few minutes (the frame is synthesised with numpy).  Usage: tools/exp_max_canvas.py [side = 32768] [share of the canvas covered by patches = 0.7] [general]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tmc2-rs_amd")); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
from tmc2rs import _abi, recon, synth
import oracle_binding as ob
side = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
cover = float(sys.argv[2]) if len(sys.argv) > 2 else 0.7
import threading
def _beat():
    k = 0
    while True:
        time.sleep(60); k += 1
        print(f"... {k} min", flush=True)
threading.Thread(target=_beat, daemon=True).start()
t0 = time.perf_counter()
f = synth.make_frame(side, side, 4, 16, seed=0x51DE0000 + side, coord_bits=16, cover_target=cover, max_side=max(24, side // 80),
                     max_patches=2000, size_skew=2.0)
print(f"{side} x {side}: {len(f['patches'])} patches, synthesised in {time.perf_counter() - t0:.0f} s", flush=True)
t0 = time.perf_counter()
st, ref = ob.reconstruct(f)
assert st == 0
print(f"oracle: {ref['n']} points ({ref['n'] * 6 / 2**32:.2f} x 4 GiB of positions) in {time.perf_counter() - t0:.0f} s", flush=True)
ctx = recon.Context(0)
t0 = time.perf_counter()
g = ctx.gof([f], flags=_abi.VPCC_GOF_FORCE_GENERAL if 'general' in sys.argv[3:] else 0)
g.reconstruct()
got = g.download(0)
print(f"HIP: {got['n']} points, create + reconstruct + download {time.perf_counter() - t0:.1f} s, kernel(s) {g.kernel_times()}", flush=True)
ok = got["n"] == ref["n"]
if ok:
    rx, rc = ob.xyz_array(ref), ob.rgb_array(ref)
    step = 1 << 26
    for a in range(0, got["n"], step):                     # piecewise: no temporaries of the whole size
        ex, ec = np.array_equal(got["xyz"][a:a + step], rx[a:a + step]), np.array_equal(got["rgb"][a:a + step], rc[a:a + step])
        if not (ex and ec) and ok:
            d = np.nonzero((got["xyz"][a:a + step] != rx[a:a + step]).any(axis=1) | (got["rgb"][a:a + step] != rc[a:a + step]).any(axis=1))[0]
            print(f"first difference at point {a + int(d[0])} (2^32 / 6 = {2**32 // 6})")
        ok = ok and ex and ec
print("equal to the oracle:", ok)
g.close(); ctx.close()
sys.exit(0 if ok else 1)
