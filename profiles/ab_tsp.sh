#!/bin/bash
# implicit TSP sweep: distance rows requested 1 / 2 / 3 updates ahead (variant libraries) x waves per replica
one() { n=$1; lib=$2; shift 2
  env $lib timeout -k 10 400 python bench.py "$@" --no-cpu-baseline > gpurun_out/ab_$n.json 2> gpurun_out/ab_$n.err
  python - $n <<PY
import json,sys
n=sys.argv[1]
try:
    d=json.load(open(f"gpurun_out/ab_{n}.json")); print(n, "%.3f ms/step"%d["ms_per_step"], d["config"]["geometry"][:75], flush=True)
except Exception as e: print(n,"ERR",e, flush=True)
PY
}
B=$GRAFT_REPO_ROOT/build
K="--workload c5 --implicit --cities 1000 --replicas 256 --steps 2 --warmup 1"
for w in 0 2; do
  one k_ra1_w$w SGA_LIBRARY_PATH=$B/libsga_ra1.so $K --waves $w
  one k_ra2_w$w X=1 $K --waves $w
  one k_ra3_w$w SGA_LIBRARY_PATH=$B/libsga_ra3.so $K --waves $w
done
one c_ra1 SGA_LIBRARY_PATH=$B/libsga_ra1.so --workload c5 --implicit
one c_ra2 X=1 --workload c5 --implicit
one c_ra3 SGA_LIBRARY_PATH=$B/libsga_ra3.so --workload c5 --implicit
