#!/bin/bash
# Rebuilds with different cache-policy masks (VPCC_TILES_NT) and reports time + L2-fabric traffic: run on the GPU box.
root=$GRAFT_REPO_ROOT
for m in "$@"; do
  cd $root; touch tmc2-rs_amd/csrc/vpcc_tiles.hip
  make product EXTRA=-DVPCC_TILES_NT=$m >/dev/null 2>&1 || { echo "build failed"; exit 1; }
  python -m pytest tests/test_parity_gpu.py -x -q 2>&1 | tail -1
  python bench.py --steps 100 --warmup 10 --ramp-ms 50 --no-cpu-baseline --profile-steps 3 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('NT $m', d['ms_per_step'], d['roofline']['all_kernels_ms'])"
  cd /tmp; export TMPDIR=/tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $root/gpurun_out/nt/m${m}_$c -- python3 $root/bench.py --steps 3 --warmup 1 --ramp-ms 0 --no-cpu-baseline --profile-steps 1 > /dev/null 2>&1
  done
  cd $root; python tools/pmc_summary.py gpurun_out/nt | grep "m${m}_"
done
