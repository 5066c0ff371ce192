#!/bin/bash
# the round's record: bench lines of every workload (gpurun); the rocprofv3 passes are collect_r05.sh
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
b() { tag=$1; shift; timeout -k 10 900 python bench.py "$@" > gpurun_out/r05_bench_$tag.json 2> gpurun_out/r05_bench_$tag.err; echo "bench $tag rc=$?"; }
b c2a_f32 --steps 20 --warmup 5
b 2rank_gloo --gpus 2 --backend gloo --share-device --steps 20 --warmup 5
b c2a_f32_force_dist --force-dist --no-variants --no-cpu-baseline --configs c4,c5,c5_1000
b c3_csr --workload c3
b c4_csr --workload c4
b c5_csr --workload c5
b c5_1000_csr --workload c5 --cities 1000 --replicas 256 --steps 2 --warmup 1
b c5_implicit --workload c5 --implicit
b c5_1000_implicit --workload c5 --implicit --cities 1000 --replicas 256 --steps 3 --warmup 2
