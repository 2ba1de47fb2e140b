#!/usr/bin/env python3
"""Where the Decoder's input buffer lies (NUMA), and what that does to the end-to-end rate on ONE GPU.
The Decoder interleaves its input over the NUMA nodes of its lanes' GPUs when they are more than one; on a box with one GPU
VPCC_DECODER_INTERLEAVE_NODES=0,1 forces it.  Prints the policy and the pages per node of the process's largest mapping
(/proc/self/numa_maps) right after start(), then the rate.  Usage: tools/exp_input_interleave.py [gofs]"""
import os, sys, tempfile, re
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tmc2-rs_amd"))
from tmc2rs import container, recon, synth
n_gofs = int(sys.argv[1]) if len(sys.argv) > 1 else 17
frames = [synth.longdress_frame(i) for i in range(32)]
d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
path = os.path.join(d, "e2e.vpccgof")
container.write_container(path, [frames] * n_gofs)
print("nodes:", sorted(x for x in os.listdir("/sys/devices/system/node") if x.startswith("node")))
try:
    for setting in (None, "0,1", None, "0,1"):
        if setting is None:
            os.environ.pop("VPCC_DECODER_INTERLEAVE_NODES", None)
        else:
            os.environ["VPCC_DECODER_INTERLEAVE_NODES"] = setting
        dec = recon.Decoder(path)
        dec.start()
        best = None
        for line in open("/proc/self/numa_maps"):
            pages = sum(int(m) for m in re.findall(r" N\d+=(\d+)", line))
            if best is None or pages > best[0]:
                best = (pages, line.strip())
        nf, npts, sec = dec.drain()
        t_first = dec.first_frame_seconds()
        dec.close()
        print(f"VPCC_DECODER_INTERLEAVE_NODES={setting}: {best[1][:200]}")
        print(f"    first frame after {t_first*1e3:.0f} ms, {(nf-1)/(sec-t_first):.0f} frames/s after it, {nf/sec:.0f} over the whole run")
finally:
    os.remove(path)
    os.rmdir(d)
