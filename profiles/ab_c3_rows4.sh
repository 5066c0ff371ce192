#!/bin/bash
# same-box A/B of the C3 narrow form: one update at a time (SGA_CSR_PAIR_AHEAD=0) | four | eight updates per
# step (sweep_csr_rows.hip; the default where it applies)
cd "$GRAFT_REPO_ROOT" || exit 1
run() { timeout -k 10 300 python bench.py --workload c3 --no-cpu-baseline --steps 40 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$1', '%.4e attempts/s' % d['value'], '%.3f ms/sweep' % d['roofline']['avg_launch_ms'], d['roofline'].get('kernel_instantiation',''))"; }
for rep in 1 2; do
  SGA_CSR_PAIR_AHEAD=0 run "one-at-a-time     "
  SGA_CSR_PAIR_AHEAD=4 run "four per step     "
  SGA_CSR_PAIR_AHEAD=8 run "eight per step    "
done
