#!/bin/bash
# VALU accounting of the tile kernel per ablation (diagnostic build): instruction counts by kind and the
# cycles the vector pipe spent executing them.  Usage: tools/pmc_valu.sh <outdir> <variant>...
out=$1; shift
mkdir -p "$out"; out=$(cd "$out" && pwd)
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export VPCC_DIAG_LIB=1
for v in "$@"; do
  export VPCC_TILES_VARIANT=$v
  for pass in "a SQ_INSTS_VALU SQ_INST_CYCLES_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT SQ_INSTS_SALU SQ_INST_CYCLES_SALU" \
              "b SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_LDS"; do
    set -- $pass; name=$1; shift
    rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out/v${v}_$name" -- \
      python3 "$root/bench.py" --diag --steps 3 --warmup 1 --ramp-ms 0 --no-cpu-baseline --no-verify --no-end-to-end --no-other-configs --no-gpu-state --no-compare --min-seconds 0 > "$out/v${v}_$name.log" 2>&1 || echo "variant $v pass $name failed"
  done
done
cd "$root" && python3 tools/pmc_summary.py "$out"
