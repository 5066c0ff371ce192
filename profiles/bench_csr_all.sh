#!/bin/bash
# one line per CSR workload (default forms): bash profiles/bench_csr_all.sh [extra bench args]
run() { n=$1; shift
  timeout -k 10 400 python bench.py "$@" --no-cpu-baseline > gpurun_out/b_$n.json 2> gpurun_out/b_$n.err
  python - $n <<PY
import json,sys
n=sys.argv[1]
try:
    d=json.load(open(f"gpurun_out/b_{n}.json")); print(n, "%.3f ms/step"%d["ms_per_step"], "%.4g attempts/s"%d["value"], "frac %.3f"%d["roofline"]["frac"], d["config"]["geometry"][:70], flush=True)
except Exception as e: print(n,"ERR",e, flush=True)
PY
}
run c3 --workload c3 "$@"
run c4 --workload c4 "$@"
run c5 --workload c5 "$@"
run c5k --workload c5 --cities 1000 --replicas 256 --steps 2 --warmup 1 "$@"
run c5i --workload c5 --implicit "$@"
run c5ki --workload c5 --implicit --cities 1000 --replicas 256 --steps 2 --warmup 1 "$@"
