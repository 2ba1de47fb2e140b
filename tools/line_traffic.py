#!/usr/bin/env python3
"""Cache-line-granular lower bound of the plane reads of the bench GOF (CPU only, no GPU needed):
tmc2rs/traffic.py summed over the frames bench.py uses.  Compare with the memory-side read counters.

usage: tools/line_traffic.py [--workload longdress|owlii] [--frames 32]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tmc2-rs_amd"))
from tmc2rs import synth, traffic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="longdress")
ap.add_argument("--frames", type=int, default=32)
a = ap.parse_args()
make = synth.longdress_frame if a.workload == "longdress" else synth.owlii_frame
tot = traffic.gof_read_bytes([make(i) for i in range(a.frames)])
tot["workload"], tot["frames"] = a.workload, a.frames
print(json.dumps(tot, indent=1))
