#!/usr/bin/env python3
"""Builds profiles/rNN/traffic*.json from a tools/profile_round.sh output directory.
Usage: tools/traffic_json.py <profile_round_outdir> <round> <kernel substring> "<workload text>" > profiles/rNN/traffic.json"""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tmc2-rs_amd"))
from tmc2rs import provenance
root, rnd, kernel, workload = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]


def means(sub, kern):
    """Per counter: the mean per dispatch of every kernel whose name contains `kern`, SUMMED over those kernels (one
    kernel for k_recon_tiles; the eight launches of a smoothing step for k_smooth)."""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    # (gpurun merges every call's output into the same directory: the newest file is this round's pass)
    for f in sorted(glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1:]:
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                acc[r["Counter_Name"]][r["Kernel_Name"]].append(float(r["Counter_Value"]))
    tot = {c: sum(sum(v) / len(v) for v in per.values()) for c, per in acc.items()}
    n = {c: min(len(v) for v in per.values()) for c, per in acc.items()}
    return tot, n


f, nf = means("pmc_FETCH_SIZE", kernel)
w, nw = means("pmc_WRITE_SIZE", kernel)
rd, nrd = means("pmc_rd", kernel)
wr, _ = means("pmc_wr", kernel)
exact_rd = 32 * rd.get("TCC_EA0_RDREQ_32B_sum", 0) + 64 * rd.get("TCC_EA0_RDREQ_64B_sum", 0) + 128 * rd.get("TCC_EA0_RDREQ_128B_sum", 0)
w64 = wr.get("TCC_EA0_WRREQ_64B_sum", 0)
exact_wr = 64 * w64 + 32 * (wr.get("TCC_EA0_WRREQ_sum", 0) - w64)
fetch_b, write_b = f.get("FETCH_SIZE", 0) * 1024, w.get("WRITE_SIZE", 0) * 1024
# calibration: sparse_read reads K of every 4 tiles of each 128-B line; whole lines arrive whatever K is
cal = []
per = collections.defaultdict(dict)
for p in sorted(glob.glob(os.path.join(root, "sparse_rd", "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1:]:
    for r in csv.DictReader(open(p)):
        if "k_sparse" in r["Kernel_Name"]:
            per[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
fs = {}
for p in sorted(glob.glob(os.path.join(root, "sparse_FETCH_SIZE", "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1:]:
    for r in csv.DictReader(open(p)):
        if "k_sparse" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            fs[int(r["Dispatch_Id"])] = float(r["Counter_Value"]) * 1024
cfg = [(4, 0), (2, 0), (1, 0), (3, 0), (4, 1), (1, 1)]
tiles, frames, planes = 80 * 88, 32, 5
ids, fids = sorted(per), sorted(fs)
for i, (K, rs) in enumerate(cfg):
    if 3 * i < len(ids):
        c = per[ids[3 * i]]
        cal.append({"tiles_read_of_4_per_line": K, "even_rows_only": bool(rs),
                    "requested_bytes": tiles / 4 * K * frames * planes * 512 / (2 if rs else 1),
                    "distinct_128B_line_bytes": tiles / 4 * frames * planes * 16 / (2 if rs else 1) * 128,
                    "counted_read_bytes": 32 * c.get("TCC_EA0_RDREQ_32B_sum", 0) + 64 * c.get("TCC_EA0_RDREQ_64B_sum", 0) + 128 * c.get("TCC_EA0_RDREQ_128B_sum", 0),
                    "FETCH_SIZE_bytes": fs.get(fids[3 * i]) if 3 * i < len(fids) else None})
print(json.dumps({
    "round": rnd, "kernel": kernel, "workload": workload,
    # what was measured: the library the passes loaded and the kernel sources + flags it was built from
    "library": provenance.library(), "kernel_source_sha16": provenance.kernel_source_sha16(), "family_source_sha16": provenance.kernel_source_sha16(kernel if kernel in provenance.FAMILY_SOURCES else None), "hipflags": provenance.hipflags(),
    "source": "rocprofv3 --kernel-trace --pmc, separate passes (tools/profile_round.sh); reads = 2 x FETCH_SIZE as "
              "MI355X_MICROARCH.md prescribes for gfx950, cross-checked by the exact request-size counters; writes = WRITE_SIZE",
    "launches_averaged": {"FETCH_SIZE": nf.get("FETCH_SIZE", 0), "WRITE_SIZE": nw.get("WRITE_SIZE", 0), "request_counters": max(nrd.values()) if nrd else 0},
    "FETCH_SIZE_bytes_raw": fetch_b, "fetch_correction": 2, "WRITE_SIZE_bytes": write_b,
    "exact_read_bytes": exact_rd, "exact_write_bytes": exact_wr,
    "read_requests": rd, "write_requests": wr,
    "l2_hit_rate": (wr.get("TCC_HIT_sum", 0) / (wr.get("TCC_HIT_sum", 0) + wr.get("TCC_MISS_sum", 1))) if wr else None,
    "hbm_read_bytes_per_launch": int(2 * fetch_b), "hbm_write_bytes_per_launch": int(write_b),
    "hbm_bytes_per_launch": int(2 * fetch_b + write_b),
    "calibration_sparse_read": cal}, indent=1))
