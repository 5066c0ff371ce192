#!/bin/bash
# CSR workloads under forced forms: RUNS lines are "<ENV=..> <bench args>"
out=gpurun_out/c4_forms.txt; : > $out
run() { echo "== $*" >> $out; env $1 timeout -k 10 400 python bench.py ${@:2} --no-cpu-baseline 2>>gpurun_out/c4_forms.err | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('%.4g attempts/s  %.1f ms/sweep  %.0f GB/s  %s' % (d['value'], d['ms_per_step'], d['roofline']['achieved'], d['config']['geometry']))" >> $out; }
while read -r line; do [ -z "$line" ] || run $line; done <<LIST
${RUNS}
LIST
cat $out
