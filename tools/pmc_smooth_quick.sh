#!/bin/bash
# Two counter passes over the smoothing kernels (instruction counts, wave cycles): tools/pmc_smooth_quick.sh <outdir>
out=$1; mkdir -p "$out"; out=$(cd "$out" && pwd)
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
pass() { name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out/$name" -- python3 $R/bench.py --smooth --steps 3 --warmup 1 --ramp-ms 0 --min-seconds 0 --no-cpu-baseline --no-verify --no-end-to-end --no-other-configs --no-gpu-state > "$out/$name.log" 2>&1 || echo "pass $name failed"; }
pass a SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_INSTS_VMEM_WR
pass b SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void vpcc::", "")
        if "smooth" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k, "  ".join(f"{c.replace('SQ_', '')} {sum(v)/len(v):.4g}" for c, v in sorted(acc[k].items())))
PY
