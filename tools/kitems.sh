#!/bin/bash
# Rebuilds with K items per wave and sweeps the pipeline depth: run on the GPU box.  Usage: tools/kitems.sh K depth...
k=$1; shift
touch tmc2-rs_amd/csrc/vpcc_device.hpp
make product EXTRA=-DVPCC_TILE_ITEMS_PER_WAVE=$k >/dev/null 2>&1 || { echo "build failed"; exit 1; }
python -m pytest tests/test_parity_gpu.py -x -q 2>&1 | tail -1
export VPCC_DIAG_LIB=1
for d in "$@"; do
  VPCC_TILES_DEPTH=$d python bench.py --steps 100 --warmup 10 --ramp-ms 50 --no-cpu-baseline --no-verify --no-end-to-end 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('K $k depth $d', d['ms_per_step'], d['roofline']['all_kernels_ms'])"
done
