#!/bin/bash
# Same-box A/B of PREBUILT libraries (tmc2-rs_amd/<name>.so): alternates short bench runs N times, prints min / median / max kernel ms.
# Usage: N=3 tools/ab_prebuilt.sh libvpcc_recon.so libvpcc_ab_old.so ...        [BENCH_ARGS=...]
n=${N:-3}
for r in $(seq $n); do
  for lib in "$@"; do
    VPCC_DIAG_LIB=$lib python3 bench.py --diag --steps 200 --no-cpu-baseline --no-end-to-end --no-other-configs --no-gpu-state --no-compare $BENCH_ARGS 2>/tmp/ab_err.txt \
      | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); v=d['verified_frames'][-1] if d['verified_frames'] else {}; print('$lib', d['roofline']['kernel_ms'], v.get('entries_equal_oracle'))" || { echo "$lib FAILED"; tail -3 /tmp/ab_err.txt; }
  done
done > /tmp/ab_runs.txt
python3 - <<'PY'
import statistics as st, collections
v = collections.defaultdict(list); ok = {}
for l in open("/tmp/ab_runs.txt"):
    p = l.split()
    if len(p) >= 3 and p[1] != "FAILED": v[p[0]].append(float(p[1])); ok[p[0]] = p[2]
    else: print(l.strip())
for k, x in v.items():
    print(f"{k:28s} min {min(x):.4f}  median {st.median(x):.4f}  max {max(x):.4f}  n={len(x)}  equal_oracle={ok[k]}")
PY
