#!/bin/bash
# Round-2 rocprofv3 passes (run on the GPU box through gpurun; summaries via summarize_rocprof.py):
# the dense sweep kernel on coupling matrices far beyond every cache (4.3 GB and 2.3 GB), heuristic
# geometry as in bench.py's roofline_beyond_cache block.  One counter per --pmc pass, never with
# other trace domains; the program comes directly after `--`.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
one() {  # tag, bench args...
    tag=$1; shift
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${tag}_stats --output-format csv -- python3 bench.py "$@" --no-cpu-baseline --no-variants > gpurun_out/prof_${tag}_stats.log 2>&1 &&
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/prof_${tag}_fetch --output-format csv -- python3 bench.py "$@" --no-cpu-baseline --no-variants > gpurun_out/prof_${tag}_fetch.log 2>&1 &&
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/prof_${tag}_write --output-format csv -- python3 bench.py "$@" --no-cpu-baseline --no-variants > gpurun_out/prof_${tag}_write.log 2>&1 &&
    find gpurun_out/prof_${tag}_* -name "*_kernel_trace.csv" -delete &&
    python3 profiles/summarize_rocprof.py --stats gpurun_out/prof_${tag}_stats --fetch gpurun_out/prof_${tag}_fetch \
        --write gpurun_out/prof_${tag}_write --tag r02_${tag} --note "bench.py $*" &&
    cp profiles/r02_${tag}_* gpurun_out/ &&
    rm -rf gpurun_out/prof_${tag}_stats gpurun_out/prof_${tag}_fetch gpurun_out/prof_${tag}_write  # raw CSVs: tens of MB
}
for t in "$@"; do
  case $t in
    n32768) one dense_f32_n32768 --spins 32768 --no-autotune --steps 3 --warmup 1 ;;
    n24000) one dense_f32_n24000 --spins 24000 --no-autotune --steps 3 --warmup 1 ;;
    c2a) one c2a_f32 --no-autotune ;;
    c3) one c3_csr --workload c3 ;;
    c4) one c4_csr --workload c4 ;;
    c5) one c5_csr --workload c5 ;;
    c5_1000) one c5_1000_csr --workload c5 --cities 1000 --replicas 256 --steps 2 --warmup 1 ;;
  esac || exit 1
done
echo collected
