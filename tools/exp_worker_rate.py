#!/usr/bin/env python3
"""How many frames per second the Decoder's worker thread can hand over when the frames cost next to nothing to move or to
reconstruct (64x64-canvas frames of a few thousand points): the ceiling its per-frame host work — posting a download to a lane,
waiting for it, the hand-over through the capacity-1 channel — puts on ANY number of GPUs.  Eight GPUs at 2 900 frames/s each
would need 23 000.  Usage: tools/exp_worker_rate.py [gofs]"""
import os, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tmc2-rs_amd"))
from tmc2rs import container, recon, synth
n_gofs = int(sys.argv[1]) if len(sys.argv) > 1 else 200
frames = [synth.small_frame(i) for i in range(32)]
d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
path = os.path.join(d, "small.vpccgof")
container.write_container(path, [frames] * n_gofs)
try:
    for devices in ((0,), (0, 0), (0, 0, 0, 0), (0,)):
        dec = recon.Decoder(path, devices=devices)
        dec.start()
        nf, npts, sec = dec.drain()
        t_first = dec.first_frame_seconds()
        st = dec.stats()
        dec.close()
        print(f"{len(devices)} lane(s): {nf} frames of {npts // nf} points, {(nf-1)/(sec-t_first):.0f} frames/s after the first, "
              f"{st['launches']} launches")
finally:
    os.remove(path)
    os.rmdir(d)
