#!/bin/bash
# Timing-only ablations of the fused kernel (see vpcc_fused.hip): prints kernel ms per variant.
for v in "$@"; do
  VPCC_TILES_VARIANT=$v python bench.py --steps 100 --warmup 10 --ramp-ms 50 --no-cpu-baseline --profile-steps 3 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant $v', d['ms_per_step'], d['roofline']['all_kernels_ms'])"
done
