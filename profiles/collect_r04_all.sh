#!/bin/bash
# the round's record: bench lines of every workload (gpurun), then the rocprofv3 passes (collect_r04.sh)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
b() { tag=$1; shift; timeout -k 10 600 python bench.py "$@" > gpurun_out/r04_bench_$tag.json 2> gpurun_out/r04_bench_$tag.err; echo "bench $tag rc=$?"; }
b c2a_f32 --steps 20 --warmup 5
b c2a_f32_force_dist --force-dist --no-variants --no-cpu-baseline
b c3_csr --workload c3
b c4_csr --workload c4
b c5_csr --workload c5
b c5_1000_csr --workload c5 --cities 1000 --replicas 256 --steps 2 --warmup 1
b c5_implicit --workload c5 --implicit
b c5_1000_implicit --workload c5 --implicit --cities 1000 --replicas 256 --steps 2 --warmup 1
PICK_FROM=gpurun_out/r04_bench_c2a_f32.json timeout -k 10 1000 bash profiles/collect_r04.sh headline c3 c5i c4cached > gpurun_out/r04_collect_all.log 2>&1; echo "collect rc=$?"
