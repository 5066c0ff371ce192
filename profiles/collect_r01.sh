#!/bin/bash
# rocprofv3 passes behind the committed summaries (run on the GPU box through gpurun; the CSVs
# land in gpurun_out/, profiles/summarize_rocprof.py condenses them afterwards):
#   --kernel-trace --stats        per-kernel durations
#   --kernel-trace --pmc X        one counter per pass (FETCH_SIZE, WRITE_SIZE), never with other traces
# Dense workloads: bench.py autotunes its launch geometry first; so that the traces hold the timed
# launches only (not the tuner's trial launches of other geometries), the tuner's choice is read
# from one plain run and passed to the profiled runs as --waves.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
one() {  # tag, bench args...
    tag=$1; shift
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${tag}_stats --output-format csv -- python3 bench.py "$@" --no-cpu-baseline --no-variants > gpurun_out/prof_${tag}_stats.log 2>&1 &&
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/prof_${tag}_fetch --output-format csv -- python3 bench.py "$@" --no-cpu-baseline --no-variants > gpurun_out/prof_${tag}_fetch.log 2>&1 &&
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/prof_${tag}_write --output-format csv -- python3 bench.py "$@" --no-cpu-baseline --no-variants > gpurun_out/prof_${tag}_write.log 2>&1 &&
    find gpurun_out/prof_${tag}_* -name "*_kernel_trace.csv" -delete
}
tuned() {  # bench args... -> the autotuned waves per replica
    python3 bench.py "$@" --no-cpu-baseline --no-variants 2>/dev/null | python3 -c "
import json, sys, re
print(re.search(r'waves_per_replica=(\d+)', json.loads(sys.stdin.readline())['config']['geometry']).group(1))"
}
w=$(tuned) && echo "c2a_f32: autotuned waves=$w" && one c2a_f32 --waves $w &&
w=$(tuned --storage i8) && echo "c2a_i8: autotuned waves=$w" && one c2a_i8 --storage i8 --waves $w &&
w=$(tuned --storage t2) && echo "c2a_t2: autotuned waves=$w" && one c2a_t2 --storage t2 --waves $w &&
one c3_csr --workload c3 &&
one c4_csr --workload c4 &&
one c5_csr --workload c5 &&
one c5_1000_csr --workload c5 --cities 1000 --replicas 256 --steps 2 --warmup 1
echo collected
