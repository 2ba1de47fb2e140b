#!/usr/bin/env python3
"""End-to-end (PCIe-inclusive) rate of the C++ Decoder pipeline: container in host memory -> pinned async
H2D ingest -> reconstruction -> D2H of every frame -> consumer, with GOF-level double buffering.
This is NOT bench.py's `value` (which times the hot path with planes resident in HBM); it is the figure
DESIGN.md quotes for host-buffer hand-over."""
import os, sys, time, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tmc2-rs_amd"))
from tmc2rs import container, recon, synth
n_gofs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
frames = [synth.longdress_frame(i) for i in range(32)]
d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
path = os.path.join(d, "e2e.vpccgof")
container.write_container(path, [frames] * n_gofs)
size = os.path.getsize(path)
try:
    for rep in range(2):
        dec = recon.Decoder(path)
        t0 = time.perf_counter()
        dec.start()                      # reads the container on the caller's thread (like the reference)
        t_read = time.perf_counter() - t0
        nf, npts, sec = dec.drain()
        t_first = dec.first_frame_seconds()
        dec.close()
        per_gof = (sec - t_first) / max(n_gofs - 1, 1)
        print(f"rep {rep}: start-up (contexts, page-locking {size/1e9:.1f} GB, first GOF) {t_first*1e3:.0f} ms; then "
              f"{per_gof*1e3:.2f} ms per 32-frame GOF = {32/per_gof:.0f} frames/s steady state")
        print(f"rep {rep}: {nf} frames, {npts/1e6:.1f} Mpoints in {sec:.3f} s -> {nf/sec:.0f} frames/s, {npts/sec/1e6:.0f} Mpoints/s "
              f"end-to-end ({size/1e9:.2f} GB container, file read {t_read:.2f} s not included); "
              f"H2D {size/sec/1e9:.1f} GB/s + D2H {npts*9/sec/1e9:.1f} GB/s")
finally:
    os.remove(path)
    os.rmdir(d)
