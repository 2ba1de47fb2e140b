#!/bin/bash
# A/B of compile-time variants of the smoothing kernels on the GPU box: rebuild, `bench.py --smooth`, kernel times.
# Usage: tools/ab_smooth.sh "<flags A>" "<flags B>" ...      ("" = default build; VERIFY=0 skips the spec check)
R=$GRAFT_REPO_ROOT
export PATH=/opt/rocm/bin:$PATH
i=0
for flags in "$@"; do
  i=$((i+1))
  (cd "$R" && make -j8 product EXTRA="$flags" > /tmp/ab_smooth_build_$i.log 2>&1) || { echo "[$i] build failed: $flags"; tail -5 /tmp/ab_smooth_build_$i.log; continue; }
  (cd "$R" && python3 bench.py --smooth --steps 100 --no-cpu-baseline --no-end-to-end --no-other-configs --no-gpu-state $([ "$VERIFY" = 0 ] && echo --no-verify) 2>/dev/null) | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
k = d['roofline']['all_kernels_ms']
sm = sum(v for n, v in k.items() if n.startswith('k_smooth'))
print('[$i] %-40s smoothing %.3f ms  ' % ('$flags' or '(default)', sm) + ' '.join('%s %.3f' % (n.replace('k_smooth_', '').replace('geometry', 'g').replace('color', 'c'), v) for n, v in k.items() if n.startswith('k_smooth')) + '  verified ' + str([v.get('equals_spec') for v in d['verified_frames']]))
" || echo "[$i] bench failed: $flags"
done
(cd "$R" && make -j8 product EXTRA= > /tmp/ab_smooth_restore.log 2>&1)
