#!/bin/bash
# same-box A/B of the C3 narrow form: one update at a time | pair look-ahead | + interleaved wave sums
cd "$GRAFT_REPO_ROOT" || exit 1
run() { timeout -k 10 300 python bench.py --workload c3 --no-cpu-baseline --steps 40 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$1', '%.4e attempts/s' % d['value'], '%.3f ms/sweep' % d['roofline']['avg_launch_ms'])"; }
for rep in 1 2 3; do
  run "one-at-a-time     "
  SGA_CSR_PAIR_AHEAD=1 run "pair look-ahead   "
  SGA_CSR_PAIR_AHEAD=2 run "pair + wave_sum2  "
done
