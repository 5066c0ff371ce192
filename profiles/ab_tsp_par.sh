#!/bin/bash
# same-box A/B of the implicit TSP form: one update at a time (SGA_TSP_PARALLEL=0) | 2 | 4 | 8 updates per step
# (sweep_tsp_par_kernel; the default picks by the number of cities), C5 at 1000 and at 100 cities
cd "$GRAFT_REPO_ROOT" || exit 1
run() { tag=$1; shift; timeout -k 10 400 python bench.py --workload c5 --implicit --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$tag', '%.4e attempts/s' % d['value'], '%.3f ms/sweep' % d['roofline']['avg_launch_ms'], d['roofline'].get('kernel_instantiation',''))"; }
for par in 0 2 4 8; do
  SGA_TSP_PARALLEL=$par run "1000 cities, par=$par" --cities 1000 --replicas 256 --steps 2 --warmup 1
done
for par in 0 2 4 8; do
  SGA_TSP_PARALLEL=$par run " 100 cities, par=$par"
done
run "1000 cities, default" --cities 1000 --replicas 256 --steps 2 --warmup 1
run " 100 cities, default"
