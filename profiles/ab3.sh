#!/bin/bash
# same-box A/B: in-tree library | previous commit (build/libsga_prev.so) | a named variant per workload
one() { n=$1; lib=$2; shift 2
  env $lib timeout -k 10 400 python bench.py "$@" --no-cpu-baseline > gpurun_out/ab_$n.json 2> gpurun_out/ab_$n.err
  python - $n <<PY
import json,sys
n=sys.argv[1]
try:
    d=json.load(open(f"gpurun_out/ab_{n}.json")); print(n, "%.3f ms/step"%d["ms_per_step"], "frac %.3f"%d["roofline"]["frac"], flush=True)
except Exception as e: print(n,"ERR",e, flush=True)
PY
}
B=$GRAFT_REPO_ROOT/build
K="--workload c5 --cities 1000 --replicas 256 --steps 2 --warmup 1"
for rep in 1 2; do
one c4_new X=1 --workload c4; one c4_prev SGA_LIBRARY_PATH=$B/libsga_prev.so --workload c4; one c4_vx SGA_LIBRARY_PATH=$B/libsga_vx5.so --workload c4
one c5k_new X=1 $K; one c5k_prev SGA_LIBRARY_PATH=$B/libsga_prev.so $K; one c5k_vx SGA_LIBRARY_PATH=$B/libsga_vx8.so $K
done
