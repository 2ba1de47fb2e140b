#!/bin/bash
# A/B of compile-time variants of the product library on the GPU box: for every EXTRA flag set rebuilds
# libvpcc_recon.so, runs bench.py (timing, output verified against the oracle) and one counter pass for
# the exact memory-side read/write bytes of the tile kernel.  The default build is restored at the end.
# Usage: tools/ab.sh <outdir> "<flags A>" "<flags B>" ...      ("" = default build; "env:A=1 B=2" = default build run
# with these environment variables, e.g. the VPCC_BENCH_* workload diagnostics of bench.py)
out=$1; shift
mkdir -p "$out"; out=$(cd "$out" && pwd)
R=$GRAFT_REPO_ROOT
export PATH=/opt/rocm/bin:$PATH
i=0
for flags in "$@"; do
  i=$((i+1))
  label=$flags; envs=""
  case "$flags" in env:*) envs="${flags#env:}"; flags="";; esac
  (cd "$R" && make -j8 product EXTRA="$flags" > "$out/build_$i.log" 2>&1) || { echo "[$i] build failed: $flags"; tail -5 "$out/build_$i.log"; continue; }
  (cd "$R" && export $envs BENCH_AB=1 && python3 bench.py --steps 200 --no-cpu-baseline --no-end-to-end --no-other-configs --no-compare $BENCH_ARGS > "$out/bench_$i.json" 2> "$out/bench_$i.err") || { echo "[$i] bench failed: $flags"; tail -3 "$out/bench_$i.err"; continue; }
  (cd /tmp && export $envs BENCH_AB=1 && TMPDIR=/tmp rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum \
     --output-format csv -d "$out/pmc_$i" -- python3 "$R/bench.py" --steps 3 --warmup 1 --ramp-ms 0 --min-seconds 0 --no-cpu-baseline --no-verify --no-end-to-end --no-other-configs --no-gpu-state --no-compare $BENCH_ARGS > "$out/pmc_$i.log" 2>&1) || echo "[$i] pmc failed"
  python3 - "$out" "$i" "$label" <<'PY'
import csv, glob, json, sys, collections
out, i, flags = sys.argv[1], sys.argv[2], sys.argv[3]
b = json.load(open(f"{out}/bench_{i}.json"))
acc = collections.defaultdict(list)
for f in glob.glob(f"{out}/pmc_{i}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_recon_tiles" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
rd = 128 * m.get("TCC_EA0_RDREQ_128B_sum", 0) + 64 * m.get("TCC_EA0_RDREQ_64B_sum", 0)
w64 = m.get("TCC_EA0_WRREQ_64B_sum", 0)
wr = 64 * w64 + 32 * (m.get("TCC_EA0_WRREQ_sum", 0) - w64)
r = b["roofline"]
print(f"[{i}] {flags or '(default)':48s} ms_per_step {b['ms_per_step']:.4f} kernel_ms {r['kernel_ms']:.4f}  read {rd/1e6:6.1f} MB write {wr/1e6:6.1f} MB "
      f"(floor {r['line_floor_bytes']/1e6:.0f})  verified {all(v.get('equals_oracle', v.get('entries_checked') == v.get('entries_equal_oracle')) for v in b['verified_frames'])}")
PY
done
(cd "$R" && make -j8 product EXTRA= > "$out/build_restore.log" 2>&1)
