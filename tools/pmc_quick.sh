#!/bin/bash
# Two counter passes (instruction counts; wave cycles and waits) over the kernels whose name contains <kernel substring>, on a
# short bench.py run: tools/pmc_quick.sh <outdir> <kernel substring> [bench args...]
out=$1; kern=$2; shift 2
mkdir -p "$out"; out=$(cd "$out" && pwd)
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
pass() { name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out/$name" -- python3 $R/bench.py --steps 3 --warmup 1 --ramp-ms 0 --min-seconds 0 --no-cpu-baseline --no-verify --no-end-to-end --no-other-configs --no-gpu-state --no-compare --no-fresh-gof $BENCH_ARGS > "$out/$name.log" 2>&1 || echo "pass $name failed"; }
BENCH_ARGS="$*"
pass a SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_INSTS_VMEM_WR
pass b SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
pass c SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_ACTIVE_INST_SCA
python3 - "$out" "$kern" <<'PY'
import csv, glob, sys, collections
out, kern = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void vpcc::", "").replace("vpcc::", "")
        if kern in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c, v in sorted(acc[k].items()):
        print(f"    {c.replace('SQ_', ''):22s} {sum(v)/len(v):14.5g}   (n={len(v)})")
PY
