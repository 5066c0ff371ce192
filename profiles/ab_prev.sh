#!/bin/bash
# same-box A/B of the in-tree library against build/libsga_prev.so (the previous commit's build):
#   bash profiles/ab_prev.sh "<workload tags>"     tags: c3 c4 c5 c5k
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
P=SGA_LIBRARY_PATH=$GRAFT_REPO_ROOT/build/libsga_prev.so
for t in $1; do
  case $t in
    c3) args="--workload c3";; c4) args="--workload c4";; c5) args="--workload c5";;
    c5k) args="--workload c5 --cities 1000 --replicas 256 --steps 2 --warmup 1";;
  esac
  one ${t}_new X=1 $args; one ${t}_prev $P $args; one ${t}_new2 X=1 $args; one ${t}_prev2 $P $args
done
