#!/bin/bash
# What kind of GPU did this call get?  Bus id, temperatures, a plain copy, the output pattern alone by region
# (tools/micro/store_shapes), the placement measurement of one tuned 128-frame gof, and the kernel under the ablations
# that tell reads from writes (diagnostic build).  Usage: tools/box_report.sh <outfile>
out=$1
R=$GRAFT_REPO_ROOT
{
  rocm-smi --showbus --showtemp --showclocks 2>/dev/null | grep -E "Bus|Temperature|sclk|mclk"
  echo "== store_shapes regions 40"
  timeout -k 10 120 $R/tools/micro/bin/store_shapes 40 | cut -c1-60
  echo "== placement"
  VPCC_RUNTIME_TRACE=1 timeout -k 10 200 python3 $R/tools/exp_place_curve.py 1 2>&1 | grep -E "placement|gof"
  echo "== placement, every round treated as flat (looks 16 GB further away, twice)"
  VPCC_PLACEMENT_FLAT=9 VPCC_PLACEMENT_BUDGET_MS=2000 VPCC_RUNTIME_TRACE=1 timeout -k 10 200 python3 $R/tools/exp_place_curve.py 1 2>&1 | grep -E "placement|gof"
  echo "== ablations (0 = product behaviour, 32 = no output stores, 256 = no attribute loads, 512 = no geometry loads)"
  for v in 0 32 256 512; do
    VPCC_DIAG_LIB=1 VPCC_TILES_VARIANT=$v timeout -k 10 200 python3 $R/bench.py --diag --steps 100 --no-cpu-baseline --no-verify --no-end-to-end --no-other-configs --no-compare 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d['gpu_state_under_load']
print('variant $v ms', d['ms_per_step'], 'placement', d['config']['placement'][0]['ms_as_allocated'], '->', d['config']['placement'][0]['ms_kept'], 'copy', g.get('copy_1GiB_GBps_read_plus_write'), 'mem C', g.get('memory_C'), g.get('pci_bus'))"
  done
} > $out 2>&1
