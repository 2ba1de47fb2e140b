#!/bin/bash
# Same-box A/B of two builds of the library: alternates bench.py between libvpcc_recon.so (A) and
# libvpcc_recon_diag.so (B, whatever was linked under that name) N times and prints min / median kernel ms of each.
# Run-to-run spread on one box is ~3 % (clocks), so single runs cannot rank changes of a per cent or two.
# Usage: tools/ab_libs.sh [N=8] [bench args...]
n=${1:-8}; shift
for i in $(seq $n); do
  for k in 0 1; do
    VPCC_DIAG_LIB=$k python3 bench.py --diag --steps 200 --no-cpu-baseline --no-end-to-end --no-other-configs --no-gpu-state --no-compare --no-verify "$@" 2>/dev/null \
      | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print($k, d['roofline']['kernel_ms'])"
  done
done | python3 -c "
import sys, statistics as st
v = {0: [], 1: []}
for l in sys.stdin:
    k, x = l.split(); v[int(k)].append(float(x))
for k, name in ((0, 'A libvpcc_recon.so     '), (1, 'B libvpcc_recon_diag.so')):
    print(name, 'min %.4f  median %.4f  max %.4f  n=%d' % (min(v[k]), st.median(v[k]), max(v[k]), len(v[k])))
"
