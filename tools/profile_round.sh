#!/bin/bash
# Round evidence for one bench configuration (run on the GPU box through gpurun):
#   stats/      rocprofv3 --kernel-trace --stats of the default-like bench command (kernel durations; the ~100 launches of the
#               pool's probe kernel, k_probe_outputs, are a kernel of their own in the table)
#   pmc_*/      memory-side traffic of the dominant kernel, separate counter passes as MI355X_MICROARCH.md
#               prescribes: FETCH_SIZE and WRITE_SIZE (the guide's counters; FETCH_SIZE x 2 on gfx950) and the
#               exact request-size counters TCC_EA0_RDREQ_{32B,64B,128B}, TCC_EA0_WRREQ(_64B)
#   sparse_*/   the same read counters on tools/micro/sparse_read (known bytes, part of every 128-B line used):
#               calibrates "reads happen in whole 128-B lines"
# Usage: tools/profile_round.sh <outdir> [bench args...]     e.g.  tools/profile_round.sh gpurun_out/prof --workload owlii
out=$1; shift
mkdir -p "$out"; out=$(cd "$out" && pwd)
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 $R/bench.py --min-seconds 1.5 --no-cpu-baseline --no-end-to-end --no-other-configs --no-gpu-state --no-compare --no-fresh-gof "$@" > "$out/stats.log" 2>&1
pass() { name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out/pmc_$name" -- python3 $R/bench.py --steps 3 --warmup 1 --ramp-ms 0 --min-seconds 0 --no-cpu-baseline --no-verify --no-end-to-end --no-other-configs --no-gpu-state --no-compare --no-fresh-gof $BENCH_PMC_ARGS > "$out/pmc_$name.log" 2>&1 || echo "pass $name failed"; }
BENCH_PMC_ARGS="$*"
pass FETCH_SIZE FETCH_SIZE
pass WRITE_SIZE WRITE_SIZE
pass rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
pass wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum
if [ -x "$R/tools/micro/bin/sparse_read" ]; then
  rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d "$out/sparse_rd" -- "$R/tools/micro/bin/sparse_read" > "$out/sparse_rd.log" 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/sparse_FETCH_SIZE" -- "$R/tools/micro/bin/sparse_read" > "$out/sparse_FETCH_SIZE.log" 2>&1
fi
echo "profile_round done: $out"
