#!/bin/bash
# Same-box A/B/C... of several builds of the library.  Builds every flag set up front (tmc2-rs_amd/libvpcc_ab_<i>.so),
# then ALTERNATES short bench runs between them N times and prints min / median / max kernel ms per build: single
# runs on one box differ by up to 10 % (clocks, neighbours), so only interleaved medians can rank a few per cent.
# Usage: N=6 tools/ab_multi.sh "<flags A>" "<flags B>" ...      ("" = default build)       [BENCH_ARGS=...]
R=$GRAFT_REPO_ROOT; cd "$R"
export PATH=/opt/rocm/bin:$PATH
n=${N:-6}; i=0; libs=()
for flags in "$@"; do
  i=$((i+1))
  make -j8 product EXTRA="$flags" > /tmp/ab_multi_build_$i.log 2>&1 || { echo "[$i] build failed: $flags"; tail -5 /tmp/ab_multi_build_$i.log; exit 1; }
  cp tmc2-rs_amd/libvpcc_recon.so tmc2-rs_amd/libvpcc_ab_$i.so; libs+=("libvpcc_ab_$i.so")
done
make -j8 product EXTRA= > /tmp/ab_multi_restore.log 2>&1
for r in $(seq $n); do
  i=0
  for flags in "$@"; do
    i=$((i+1))
    VPCC_DIAG_LIB=libvpcc_ab_$i.so python3 bench.py --diag --steps 200 --no-cpu-baseline --no-end-to-end --no-other-configs --no-gpu-state --no-compare --no-verify $BENCH_ARGS 2>/dev/null \
      | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print($i, d['roofline']['kernel_ms'])"
  done
done > /tmp/ab_multi_runs.txt
python3 - "$@" <<'PY'
import sys, statistics as st, collections
flags = sys.argv[1:]
v = collections.defaultdict(list)
for l in open("/tmp/ab_multi_runs.txt"):
    k, x = l.split(); v[int(k)].append(float(x))
for i, f in enumerate(flags, 1):
    x = v[i]
    print(f"[{i}] {f or '(default)':44s} min {min(x):.4f}  median {st.median(x):.4f}  max {max(x):.4f}  n={len(x)}")
PY
rm -f tmc2-rs_amd/libvpcc_ab_*.so
