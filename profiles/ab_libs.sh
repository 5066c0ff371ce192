#!/bin/bash
# same-box A/B of library variants (build/libsga_<name>.so): bash profiles/ab_libs.sh "<names>" <bench args...>
names=$1; shift
for n in $names; do
  SGA_LIBRARY_PATH=$GRAFT_REPO_ROOT/build/libsga_$n.so timeout -k 10 240 python bench.py "$@" --no-cpu-baseline > gpurun_out/ab_$n.json 2>gpurun_out/ab_$n.err
  python - "$n" <<PY
import json,sys
n=sys.argv[1]
try:
    d=json.load(open(f"gpurun_out/ab_{n}.json")); print(n, "%.4g attempts/s"%d["value"], "%.2f ms/step"%d["ms_per_step"], "frac %.3f"%d["roofline"]["frac"])
except Exception as e: print(n,"ERR",e)
PY
done
