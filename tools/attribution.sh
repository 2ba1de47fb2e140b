#!/bin/bash
# Read/write traffic attribution of the tile kernel (run on the GPU box through gpurun):
#   * exact memory-side request counts by size (gfx950: TCC_EA0_RDREQ_{32B,64B,128B}), L2 hit/miss, sectors;
#   * one set of passes per ablation of the DIAGNOSTIC build (libvpcc_recon_diag.so, VPCC_TILES_VARIANT bits:
#     128 no emit-phase geometry re-read, 256 no attribute loads, 512 no count-phase geometry loads, 32 no stores);
#   * the same counters on tools/micro/sparse_read (known bytes, part of each 128-B line used).
# Usage: tools/attribution.sh <outdir> [variant ...]      (default variants: 0 128 256 512 640 896 32)
out=$1; shift
variants=${@:-0 128 256 512 640 896 32}
mkdir -p "$out"; out=$(cd "$out" && pwd)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export VPCC_DIAG_LIB=1
pass() {  # name counters...
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out/$name" -- \
    python3 "$R/bench.py" --diag --steps 3 --warmup 1 --ramp-ms 0 --no-cpu-baseline --profile-steps 1 --no-verify --no-end-to-end --no-other-configs --no-gpu-state --no-compare \
    > "$out/$name.log" 2>&1 || echo "pass $name failed"
}
for v in $variants; do
  export VPCC_TILES_VARIANT=$v
  pass v${v}_rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
  pass v${v}_wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum
  echo "variant $v done"
done
export VPCC_TILES_VARIANT=0
pass v0_sect TCC_READ_SECTORS_sum TCC_WRITE_SECTORS_sum TCC_REQ_sum TCC_EA0_RDREQ_DRAM_sum
unset VPCC_TILES_VARIANT
if [ -x "$R/tools/micro/bin/sparse_read" ]; then
  rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum \
    --output-format csv -d "$out/sparse_rd" -- "$R/tools/micro/bin/sparse_read" > "$out/sparse_rd.log" 2>&1 || echo "sparse failed"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/sparse_fetch" -- "$R/tools/micro/bin/sparse_read" > "$out/sparse_fetch.log" 2>&1 || echo "sparse fetch failed"
fi
python3 "$R/tools/attribution_table.py" "$out" > "$out/table.md" 2>&1
cat "$out/table.md"
