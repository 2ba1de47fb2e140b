#!/bin/bash
# Memory-side counter passes (FETCH_SIZE, WRITE_SIZE, L2 hits / misses, requests to the fabric by size) over the kernels whose name
# contains <kernel substring>, on a short bench.py run: tools/pmc_mem.sh <outdir> <kernel substring> [bench args...]
out=$1; kern=$2; shift 2
mkdir -p "$out"; out=$(cd "$out" && pwd)
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
pass() { name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out/$name" -- python3 $R/bench.py --steps 3 --warmup 1 --ramp-ms 0 --min-seconds 0 --no-cpu-baseline --no-verify --no-end-to-end --no-other-configs --no-gpu-state --no-compare --no-fresh-gof $BENCH_ARGS > "$out/$name.log" 2>&1 || echo "pass $name failed"; }
BENCH_ARGS="$*"
pass f FETCH_SIZE
pass w WRITE_SIZE
pass h TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass r TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
pass t TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum
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
        print(f"    {c:28s} {sum(v)/len(v):14.5g}   (n={len(v)})")
PY
