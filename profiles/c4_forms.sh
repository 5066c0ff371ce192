#!/bin/bash
out=gpurun_out/c4_forms.txt; : > $out
run() { echo "== $*" >> $out; env $1 timeout -k 10 400 python bench.py ${@:2} --no-cpu-baseline 2>>gpurun_out/c4_forms.err | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('%.4g attempts/s  %.1f ms/sweep  %.0f GB/s  %s' % (d['value'], d['ms_per_step'], d['roofline']['achieved'], d['config']['geometry']))" >> $out; }
run X=1 --workload c4
run SGA_FORCE_CSR_BIG=1 --workload c4
run SGA_FORCE_CSR_BIG=1 --workload c4 --waves 4
run SGA_FORCE_CSR_BIG=1 --workload c4 --waves 1
run X=1 --workload c4 --replicas 768
run SGA_FORCE_CSR_BIG=1 --workload c4 --replicas 2048
run X=1 --workload c4 --replicas 2048
cat $out
