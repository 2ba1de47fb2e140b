#!/bin/bash
# Round evidence for the smoothing kernels (BASELINE config 4), run on the GPU box through gpurun:
#   stats/   rocprofv3 --kernel-trace --stats of `bench.py --smooth` (kernel durations)
#   pmc_*/   memory-side traffic of the k_smooth_* kernels, separate counter passes (FETCH_SIZE x 2 on gfx950,
#            WRITE_SIZE, exact request-size counters) — tools/traffic_json.py <out> NN k_smooth "..." sums the kernels
# Usage: tools/profile_smooth.sh <outdir>
out=$1; shift
mkdir -p "$out"; out=$(cd "$out" && pwd)
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 $R/bench.py --smooth --min-seconds 1.5 --no-cpu-baseline --no-end-to-end --no-other-configs --no-gpu-state --no-compare --no-fresh-gof > "$out/stats.log" 2>&1
pass() { name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out/pmc_$name" -- python3 $R/bench.py --smooth --steps 3 --warmup 1 --ramp-ms 0 --min-seconds 0 --no-cpu-baseline --no-verify --no-end-to-end --no-other-configs --no-gpu-state --no-compare --no-fresh-gof > "$out/pmc_$name.log" 2>&1 || echo "pass $name failed"; }
pass FETCH_SIZE FETCH_SIZE
pass WRITE_SIZE WRITE_SIZE
pass rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
pass wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum
pass at TCC_ATOMIC_sum TCC_REQ_sum
echo "profile_smooth done: $out"
