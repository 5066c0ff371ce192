#!/bin/bash
# C4 with packed entries: rows requested 1 / 2 / 3 updates ahead (variant libraries of the 5-head-slot unit)
B=$GRAFT_REPO_ROOT/build
for rep in 1 2; do
for v in base ra2_5 ra3_5; do
  if [ $v = base ]; then lib=X=1; else lib=SGA_LIBRARY_PATH=$B/libsga_$v.so; fi
  env $lib python bench.py --workload c4 --no-cpu-baseline > gpurun_out/ab_$v.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/ab_$v.json')); print('$v', 'f32 %.2f ms' % d['ms_per_step'], 'packed %.2f ms' % d['variants']['packed_entries']['ms_per_step'], flush=True)"
done
done
