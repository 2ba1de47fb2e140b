#!/bin/bash
# Sweep of VPCC_TILES_DEPTH (groups pipelined per workgroup): prints ms per step per depth.
export VPCC_DIAG_LIB=1
for d in "$@"; do
  VPCC_TILES_DEPTH=$d python bench.py --diag --steps 100 --warmup 10 --ramp-ms 50 --no-cpu-baseline --no-verify --no-end-to-end --no-other-configs --no-gpu-state --no-compare 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('depth $d', d['ms_per_step'], d['roofline']['all_kernels_ms'])"
done
