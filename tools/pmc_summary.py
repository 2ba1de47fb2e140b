#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc CSV output (tools/pmc.sh) per kernel: mean counter value per dispatch."""
import csv, glob, os, sys, collections
root = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "k_recon"
for f in sorted(glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if kern in r.get("Kernel_Name", ""):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f"{os.path.relpath(f, root).split(os.sep)[0]:8s} {k:32s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
