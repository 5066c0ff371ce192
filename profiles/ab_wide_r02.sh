#!/bin/bash
# same-box A/B of the wide CSR forms after the zero-slot change: rows requested ahead (variant libraries from
# profiles/build_variant.sh: -DCSR_WIDE_ROWS_AHEAD=1|3 on the 5- and 8-head-slot units) and waves per replica
run() { # name env args
  n=$1; shift; e=$1; shift
  env $e timeout -k 10 400 python bench.py "$@" --no-cpu-baseline > gpurun_out/ab_$n.json 2> gpurun_out/ab_$n.err
  python - $n <<PY
import json,sys
n=sys.argv[1]
try:
    d=json.load(open(f"gpurun_out/ab_{n}.json")); print(n, "%.2f ms/step"%d["ms_per_step"], "frac %.3f"%d["roofline"]["frac"], d["config"]["geometry"][:60], flush=True)
except Exception as e: print(n,"ERR",e, flush=True)
PY
}
L=$GRAFT_REPO_ROOT/build
run c4_base X=1 --workload c4
run c4_ra3 SGA_LIBRARY_PATH=$L/libsga_ra3_5.so --workload c4
run c4_ra1 SGA_LIBRARY_PATH=$L/libsga_ra1_5.so --workload c4
run c4_w4 X=1 --workload c4 --waves 4
run c4_base2 X=1 --workload c4
run c5_base X=1 --workload c5
run c5_w2 X=1 --workload c5 --waves 2
K="--workload c5 --cities 1000 --replicas 256 --steps 2 --warmup 1"
run c5k_base X=1 $K
run c5k_ra3 SGA_LIBRARY_PATH=$L/libsga_ra3_8.so $K
run c5k_ra1 SGA_LIBRARY_PATH=$L/libsga_ra1_8.so $K
run c5k_w4 X=1 $K --waves 4
