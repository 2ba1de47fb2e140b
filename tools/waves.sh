#!/bin/bash
# Rebuilds the library with different register budgets (waves per SIMD) and times each: run on the GPU box.
for w in "$@"; do
  touch tmc2-rs_amd/csrc/vpcc_tiles.hip
  make product EXTRA=-DVPCC_TILES_WAVES_PER_EU=$w >/dev/null 2>&1 || { echo "build failed for $w"; exit 1; }
  python bench.py --steps 100 --warmup 10 --ramp-ms 50 --no-cpu-baseline --profile-steps 3 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('waves_per_eu $w', d['ms_per_step'], d['roofline']['all_kernels_ms'])"
done
