#!/usr/bin/env python3
"""Cold start of the streaming Decoder: every run in a process of its own (nothing of the GPU touched before), the worker's
steps traced (VPCC_DECODER_TRACE=2).  Usage: tools/exp_cold_start.py [runs] [gofs]"""
import os, subprocess, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
gofs = int(sys.argv[2]) if len(sys.argv) > 2 else 17
child = r'''
import os, sys, time
sys.path.insert(0, os.path.join(%r, "tmc2-rs_amd"))
from tmc2rs import recon
t0 = time.perf_counter()
dec = recon.Decoder(sys.argv[1])
dec.start()
t1 = time.perf_counter()
nf, npts, sec = dec.drain()
print(f"start() {t1 - t0:.3f} s (reads the container); drain: first frame after {dec.first_frame_seconds()*1e3:.0f} ms, {nf} frames in {sec:.3f} s = {nf/sec:.0f} frames/s whole run, "
      f"{(nf-1)/(sec-dec.first_frame_seconds()):.0f} after the first frame; start() call to first frame {t1 - t0 + dec.first_frame_seconds():.3f} s", flush=True)
dec.close()
''' % REPO
sys.path.insert(0, os.path.join(REPO, "tmc2-rs_amd"))
from tmc2rs import container, synth
frames = [synth.longdress_frame(i) for i in range(32)]
d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
path = os.path.join(d, "cold.vpccgof")
container.write_container(path, [frames] * gofs)
try:
    for r in range(runs):
        env = dict(os.environ, VPCC_DECODER_TRACE="2")
        out = subprocess.run([sys.executable, "-c", child, path], capture_output=True, text=True, env=env)
        print(f"--- run {r} (a process of its own)")
        for l in out.stderr.splitlines():
            if any(k in l for k in ("contexts ready", "page-locked", "first unit", "first frame")):
                print("   ", l)
        print("   ", out.stdout.strip() or out.stderr[-400:])
finally:
    os.remove(path)
    os.rmdir(d)
