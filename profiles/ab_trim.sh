#!/bin/bash
# same-box A/B against build/libsga_prev.so: lanes behind a row's last entry read the zero slot (new) or the padding (prev)
one() { n=$1; lib=$2; shift 2
  env $lib timeout -k 10 400 python bench.py "$@" --no-cpu-baseline --no-variants > gpurun_out/ab_$n.json 2> gpurun_out/ab_$n.err
  python - $n <<PY
import json,sys
n=sys.argv[1]
try:
    d=json.load(open(f"gpurun_out/ab_{n}.json")); print(n, "%.3f ms/step"%d["ms_per_step"], "frac %.3f"%d["roofline"]["frac"], flush=True)
except Exception as e: print(n,"ERR",e, flush=True)
PY
}
P=SGA_LIBRARY_PATH=$GRAFT_REPO_ROOT/build/libsga_prev.so
K="--cities 1000 --replicas 256 --steps 2 --warmup 1"
for rep in 1 2; do
one c4_new X=1 --workload c4; one c4_prev $P --workload c4
one c5_new X=1 --workload c5; one c5_prev $P --workload c5
one c5k_new X=1 --workload c5 $K; one c5k_prev $P --workload c5 $K
done
