#!/bin/bash
# Timing-only ablations of the tile kernel (VPCC_TILES_VARIANT bits, see vpcc_tiles.hip: 1 no look-back wait, 8 no colour,
# 32 no stores, 256 no attribute loads, 512 no geometry loads): prints ms per step and kernel ms per variant.
export VPCC_DIAG_LIB=1   # ablation switches exist in libvpcc_recon_diag.so only (make diag)
for v in "$@"; do
  VPCC_TILES_VARIANT=$v python bench.py --diag --steps 100 --warmup 10 --ramp-ms 50 --no-cpu-baseline --no-verify --no-end-to-end --no-other-configs --no-gpu-state --no-compare 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant $v', d['ms_per_step'], d['roofline']['all_kernels_ms'])"
done
