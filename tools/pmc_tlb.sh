#!/bin/bash
# Address-translation counters of the tile kernel over 8 gofs of one process.  Usage: tools/pmc_tlb.sh <outdir>
out=$1; mkdir -p "$out"; out=$(cd "$out" && pwd)
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export VPCC_OUTPUT_CANDIDATES=1
rocprofv3 --list-avail > "$out/avail.txt" 2>&1
grep -i -o "[A-Z0-9_]*UTCL[A-Za-z0-9_]*" "$out/avail.txt" | sort -u > "$out/utcl_names.txt"
run() { name=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out/$name" -- python3 "$root/tools/exp_tlb.py" 8 > "$out/$name.log" 2>&1 || echo "pass $name failed"; }
run t1 TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum
run t2 TCP_UTCL1_PERMISSION_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
