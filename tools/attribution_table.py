#!/usr/bin/env python3
"""Turns the counter passes of tools/attribution.sh into a markdown table (bytes per launch of the tile kernel).
Read bytes are exact: 32*n32 + 64*n64 + 128*n128 of TCC_EA0_RDREQ_{32B,64B,128B}; write bytes: 64-B requests
x 64 + the rest x 32 (TCC_EA0_WRREQ / _64B)."""
import collections
import csv
import glob
import json
import os
import sys

root = sys.argv[1]


def means(name, kernel):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(root, name, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel in r.get("Kernel_Name", ""):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def read_bytes(m):
    n32, n64, n128 = m.get("TCC_EA0_RDREQ_32B_sum", 0), m.get("TCC_EA0_RDREQ_64B_sum", 0), m.get("TCC_EA0_RDREQ_128B_sum", 0)
    other = m.get("TCC_EA0_RDREQ_sum", 0) - n32 - n64 - n128
    return 32 * n32 + 64 * n64 + 128 * n128, other


rows = []
names = {0: "production kernel", 128: "no emit-phase geometry re-read", 256: "no attribute loads", 2048: "no chroma loads", 4096: "no luma loads",
         512: "no count-phase geometry loads", 640: "no geometry loads at all (128+512)",
         896: "occupancy only (128+256+512)", 32: "no output stores", 384: "count-phase geometry only (128+256)"}
out = {"variants": {}}
print("| variant | what | read MB | 32-B | 64-B | 128-B req | other req | write MB | L2 hit rate |")
print("|---|---|---|---|---|---|---|---|---|")
for d in sorted(glob.glob(os.path.join(root, "v*_rd")), key=lambda p: int(os.path.basename(p)[1:].split("_")[0])):
    v = int(os.path.basename(d)[1:].split("_")[0])
    m, n = means(os.path.basename(d), "k_recon_tiles")
    if not m:
        continue
    w, _ = means(f"v{v}_wr", "k_recon_tiles")
    rb, other = read_bytes(m)
    w64 = w.get("TCC_EA0_WRREQ_64B_sum", 0)
    wb = 64 * w64 + 32 * (w.get("TCC_EA0_WRREQ_sum", 0) - w64)
    hit, miss = w.get("TCC_HIT_sum", 0), w.get("TCC_MISS_sum", 0)
    hr = hit / (hit + miss) if hit + miss else float("nan")
    print(f"| {v} | {names.get(v, '')} | {rb/1e6:.1f} | {m.get('TCC_EA0_RDREQ_32B_sum',0):.0f} | {m.get('TCC_EA0_RDREQ_64B_sum',0):.0f} | "
          f"{m.get('TCC_EA0_RDREQ_128B_sum',0):.0f} | {other:.0f} | {wb/1e6:.1f} | {hr:.3f} |")
    out["variants"][str(v)] = {"what": names.get(v, ""), "read_bytes": rb, "write_bytes": wb, "l2_hit_rate": hr,
                               "rdreq": {k: m[k] for k in m}, "wrreq": {k: w[k] for k in w},
                               "launches_averaged": max(n.values()) if n else 0}
s, _ = means("v0_sect", "k_recon_tiles")
if s:
    print()
    print("production kernel, L2 side: " + ", ".join(f"{k} {v:.4g}" for k, v in sorted(s.items())))
    out["l2_side"] = s
# calibration on the sparse-read micro-benchmark: one kernel dispatch per configuration and repetition
cal = []
for f in glob.glob(os.path.join(root, "sparse_rd", "**", "*counter_collection.csv"), recursive=True):
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if "k_sparse" in r.get("Kernel_Name", ""):
            per[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    cal = [per[k] for k in sorted(per)]
fetch = []
for f in glob.glob(os.path.join(root, "sparse_fetch", "**", "*counter_collection.csv"), recursive=True):
    per = {}
    for r in csv.DictReader(open(f)):
        if "k_sparse" in r.get("Kernel_Name", "") and r["Counter_Name"] == "FETCH_SIZE":
            per[int(r["Dispatch_Id"])] = float(r["Counter_Value"])
    fetch = [per[k] for k in sorted(per)]
if cal:
    cfg = [(4, 0), (2, 0), (1, 0), (3, 0), (4, 1), (1, 1)]
    tiles, frames, planes = 80 * 88, 32, 5
    print()
    print("sparse_read calibration (3 repetitions each; requested = bytes the lanes load, lines = distinct 128-B lines x 128):")
    print("| K of 4 tiles per line | rows | requested MB | lines MB | counted read MB (3 reps) | FETCH_SIZE KB x 1024 (3 reps) |")
    print("|---|---|---|---|---|---|")
    out["calibration"] = []
    for i, (K, rs) in enumerate(cfg):
        req = tiles / 4 * K * frames * planes * 512 / (2 if rs else 1)
        lines = tiles / 4 * frames * planes * 16 / (2 if rs else 1) * 128
        got = [read_bytes(c)[0] for c in cal[3 * i:3 * i + 3]]
        fs = [x * 1024 for x in fetch[3 * i:3 * i + 3]]
        print(f"| {K} | {'even' if rs else 'all'} | {req/1e6:.1f} | {lines/1e6:.1f} | {', '.join(f'{g/1e6:.1f}' for g in got)} | "
              f"{', '.join(f'{g/1e6:.1f}' for g in fs)} |")
        out["calibration"].append({"K": K, "even_rows_only": bool(rs), "requested_bytes": req, "line_bytes": lines,
                                   "counted_read_bytes": got, "fetch_size_bytes": fs})
json.dump(out, open(os.path.join(root, "attribution.json"), "w"), indent=1)
