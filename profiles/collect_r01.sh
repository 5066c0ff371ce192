#!/bin/bash
# rocprofv3 passes behind the committed summaries (run on the GPU box through gpurun; the CSVs
# land in gpurun_out/, profiles/summarize_rocprof.py condenses them afterwards):
#   --kernel-trace --stats        per-kernel durations
#   --kernel-trace --pmc X        one counter per pass (FETCH_SIZE, WRITE_SIZE), never with other traces
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
one() {  # tag, bench args...
    tag=$1; shift
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${tag}_stats --output-format csv -- python3 bench.py "$@" --no-cpu-baseline --no-variants > gpurun_out/prof_${tag}_stats.log 2>&1 &&
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/prof_${tag}_fetch --output-format csv -- python3 bench.py "$@" --no-cpu-baseline --no-variants > gpurun_out/prof_${tag}_fetch.log 2>&1 &&
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/prof_${tag}_write --output-format csv -- python3 bench.py "$@" --no-cpu-baseline --no-variants > gpurun_out/prof_${tag}_write.log 2>&1 &&
    find gpurun_out/prof_${tag}_* -name "*_kernel_trace.csv" -delete
}
one c2a_f32 &&
one c2a_i8 --storage i8 &&
one c2a_t2 --storage t2 &&
one c3_csr --workload c3 &&
one c4_csr --workload c4 &&
one c5_csr --workload c5 &&
one c5_1000_csr --workload c5 --cities 1000 --replicas 256 --steps 2 --warmup 1
echo collected
