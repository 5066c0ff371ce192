#!/bin/bash
# bench.py at varying replica counts / waves per replica: bash profiles/residency.sh "<env> <args>" ...
out=gpurun_out/residency.txt; : > $out
run() { echo "== $*" >> $out; env $1 timeout -k 10 400 python bench.py ${@:2} --no-cpu-baseline --no-variants 2>>gpurun_out/residency.err | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('%.4g attempts/s  %.2f ms/sweep  frac %.3f  %s' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['geometry']))" >> $out; }
while read -r line; do [ -z "$line" ] || run $line; done <<LIST
${RUNS}
LIST
cat $out
