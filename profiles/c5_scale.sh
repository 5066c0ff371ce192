#!/bin/bash
# C5 (TSP QUBO, CSR) at growing size on one GPU; pass extra bench flags per line below.
set -o pipefail
out=gpurun_out/c5_scale.txt; : > $out
run() { echo "== $*" >> $out; timeout -k 10 400 python bench.py "$@" --no-cpu-baseline 2>>gpurun_out/c5_scale.err | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('%.4g attempts/s  %.1f ms/sweep  %.0f GB/s  %s' % (d['value'], d['ms_per_step'], d['roofline']['achieved'], d['config']['geometry']))" >> $out || exit 1; }
while read -r line; do [ -z "$line" ] || run $line || break; done <<LIST
${C5_RUNS:-"--workload c5"}
LIST
cat $out
