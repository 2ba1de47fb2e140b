#!/bin/bash
# Round profile: kernel-trace stats + PMC traffic passes of `bench.py`, plus a FETCH_SIZE/WRITE_SIZE
# calibration on a micro-benchmark with a known byte count and the same 8-B/lane tile access pattern
# (MI355X_MICROARCH.md: FETCH_SIZE is uncalibrated for access widths other than 16 B/lane).
# Usage (on the GPU box): tools/profile_round.sh <outdir>
out=$1; mkdir -p "$out"; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 $R/bench.py --steps 10 --warmup 2 --ramp-ms 0 --no-cpu-baseline > "$out/stats.log" 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out/pmc_$c" -- python3 $R/bench.py --steps 3 --warmup 1 --ramp-ms 0 --no-cpu-baseline --no-verify --no-end-to-end --min-seconds 0 > "$out/pmc_$c.log" 2>&1
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out/cal_$c" -- $R/tools/micro/bin/tile_feat > "$out/cal_$c.log" 2>&1
done
