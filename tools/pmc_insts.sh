#!/bin/bash
# Instruction counts of the tile kernel under ablation variants.  Usage: tools/pmc_insts.sh <outdir> <variant>...
out=$1; shift
mkdir -p "$out"; out=$(cd "$out" && pwd)
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export VPCC_DIAG_LIB=1   # ablation switches exist in libvpcc_recon_diag.so only (make diag)
for v in "$@"; do
  export VPCC_TILES_VARIANT=$v
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES --output-format csv -d "$out/v$v" -- \
    python3 "$root/bench.py" --diag --steps 3 --warmup 1 --ramp-ms 0 --no-cpu-baseline --no-verify --no-end-to-end --no-other-configs --no-gpu-state --no-compare --min-seconds 0 > "$out/v$v.log" 2>&1 || echo "variant $v failed"
done
cd "$root" && python tools/pmc_summary.py "$out"
