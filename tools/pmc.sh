#!/bin/bash
# PMC passes for the dominant kernel (run on the GPU box through gpurun).  One rocprofv3 run per
# counter group (SQ: 8 slots, TCC: FETCH_SIZE 3 / WRITE_SIZE 2), kernel-trace only, as
# MI355X_MICROARCH.md prescribes.  Usage: tools/pmc.sh <outdir> [bench args...]
out=$1; shift
mkdir -p "$out"; out=$(cd "$out" && pwd)
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out/$name" -- \
    python3 "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --ramp-ms 0 --no-cpu-baseline --no-verify --no-end-to-end --no-other-configs --no-gpu-state --no-compare --min-seconds 0 $BENCH_ARGS \
    > "$out/$name.log" 2>&1 || echo "pass $name failed"
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM
run sq3 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MFMA_MOPS_F64
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
