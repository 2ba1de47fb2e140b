#!/usr/bin/env python3
"""The Decoder's lanes on ONE GPU: devices=(0,) against (0, 0) and (0, 0, 0, 0) — every lane has its own thread, context, streams,
pool and page-locked result blocks and gets every G-th frame of a unit; the link is shared, so the rate cannot rise — it shows what the
lanes' host work and the smaller units (32 / 16 frames per lane and launch) cost.  Usage: tools/exp_two_lanes.py [gofs]"""
import os, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tmc2-rs_amd"))
from tmc2rs import container, recon, synth
n_gofs = int(sys.argv[1]) if len(sys.argv) > 1 else 17
frames = [synth.longdress_frame(i) for i in range(32)]
d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
path = os.path.join(d, "e2e.vpccgof")
container.write_container(path, [frames] * n_gofs)
try:
    for devices in ((0,), (0, 0), (0, 0, 0, 0), (0,)):
        dec = recon.Decoder(path, devices=devices)
        dec.start()
        nf, npts, sec = dec.drain()
        t_first = dec.first_frame_seconds()
        st = dec.stats()
        dec.close()
        print(f"{len(devices)} lane(s): first frame after {t_first*1e3:.0f} ms, {(nf-1)/(sec-t_first):.0f} frames/s after it, {nf/sec:.0f} over the whole run; "
              f"{st['launches']} launches, largest {st['max_frames_per_launch']} frames, host plan + enqueue {st['launch_seconds']/nf*1e6:.1f} us per frame (slowest lane)")
finally:
    os.remove(path)
    os.rmdir(d)
