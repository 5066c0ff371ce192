#!/bin/bash
# Round-5 rocprofv3 passes (run on the GPU box through gpurun; summaries via summarize_rocprof.py).
#   headline  bench.py C2a fp32 with --waves = the autotuner's pick of the committed line: stats + FETCH_SIZE + WRITE_SIZE
#   beyond    the same kernel on a 4.3 GB matrix beyond every cache (n = 32768): stats + FETCH_SIZE + WRITE_SIZE
#   c3        bench.py --workload c3: stats, traffic, the instruction counters its roofline block reads and the wait /
#             active-cycle breakdown (where the idle issue slots go: r05_experiments.md)
#   c5ki      configs[4] at 1000 cities, couplings implicit (sweep_tsp_par_kernel): stats, traffic, instruction + wait counters
#   c5k       the same instance with the 32 GB of CSR written out: stats + traffic
#   c4 | c5   the graded lines of C4 and C5 at 100 cities: stats + traffic
# One counter set per --pmc pass, never with other trace domains; the program comes directly after `--`.
# A counter set the box does not know is skipped (logged), the rest of the tag still summarised.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
R=r05
ISSUE="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD;SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SMEM;SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU;SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU;SQ_INSTS_VALU_MFMA_I8 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_WAVES;GRBM_GUI_ACTIVE"
passes() {  # tag, note, extra counter sets ("" | "A B C;D E"), program args...
    tag=$1; note=$2; extra=$3; shift 3
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${tag}_stats --output-format csv -- "$@" > gpurun_out/prof_${tag}_stats.log 2>&1 &&
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/prof_${tag}_fetch --output-format csv -- "$@" > gpurun_out/prof_${tag}_fetch.log 2>&1 &&
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/prof_${tag}_write --output-format csv -- "$@" > gpurun_out/prof_${tag}_write.log 2>&1 || return 1
    ex=""; i=0
    IFS=';' read -ra sets <<< "$extra"
    for set in "${sets[@]}"; do
        [ -z "$set" ] && continue
        i=$((i+1))
        if rocprofv3 --kernel-trace --pmc $set -d gpurun_out/prof_${tag}_x$i --output-format csv -- "$@" > gpurun_out/prof_${tag}_x$i.log 2>&1 &&
           [ -n "$(find gpurun_out/prof_${tag}_x$i -name '*_counter_collection.csv' | head -1)" ]; then
            ex="$ex --extra gpurun_out/prof_${tag}_x$i"
        else
            echo "counter set skipped on this box: $set (gpurun_out/prof_${tag}_x$i.log)"; tail -3 gpurun_out/prof_${tag}_x$i.log
        fi
    done
    find gpurun_out/prof_${tag}_* -name "*_kernel_trace.csv" -delete
    python3 profiles/summarize_rocprof.py --stats gpurun_out/prof_${tag}_stats --fetch gpurun_out/prof_${tag}_fetch \
        --write gpurun_out/prof_${tag}_write $ex --tag ${R}_${tag} --note "$note" &&
    cp profiles/${R}_${tag}_* gpurun_out/ &&
    rm -rf gpurun_out/prof_${tag}_stats gpurun_out/prof_${tag}_fetch gpurun_out/prof_${tag}_write gpurun_out/prof_${tag}_x*[0-9]
}
for t in "$@"; do
  case $t in
    avail) rocprofv3 -L > gpurun_out/${R}_counters_avail.txt 2>&1; grep -c . gpurun_out/${R}_counters_avail.txt ;;
    headline)
      W=$(python3 -c "import json,re; d=json.load(open('${PICK_FROM:-gpurun_out/${R}_bench_c2a_f32.json}')); print(re.search(r'waves_per_replica=(\d+)', d['config']['geometry']).group(1))")
      echo "autotuner's pick of the committed line: $W waves per replica"
      passes c2a_f32 "bench.py --waves $W --no-variants --no-cpu-baseline (the autotuner's pick of the committed line)" "" \
             python3 bench.py --waves $W --no-variants --no-cpu-baseline ;;
    beyond) passes dense_f32_n32768 "bench.py --spins 32768 --no-autotune --steps 3 --warmup 1 --no-variants --no-cpu-baseline (4.3 GB of fp32 couplings: beyond every cache)" "" \
             python3 bench.py --spins 32768 --no-autotune --steps 3 --warmup 1 --no-variants --no-cpu-baseline ;;
    c3) passes c3_csr "bench.py --workload c3 --no-cpu-baseline (several updates per step: sweep_csr_rows_kernel)" "$ISSUE" \
               python3 bench.py --workload c3 --no-cpu-baseline ;;
    c5ki) passes c5_1000_implicit "bench.py --workload c5 --cities 1000 --replicas 256 --implicit --steps 3 --warmup 2 --no-cpu-baseline (sweep_tsp_par_kernel)" "$ISSUE" \
               python3 bench.py --workload c5 --cities 1000 --replicas 256 --implicit --steps 3 --warmup 2 --no-cpu-baseline ;;
    c5i) passes c5_100_implicit "bench.py --workload c5 --implicit --no-cpu-baseline (sweep_tsp_par_kernel)" "$ISSUE" \
               python3 bench.py --workload c5 --implicit --no-cpu-baseline ;;
    c4) passes c4_csr "bench.py --workload c4 --no-variants --no-cpu-baseline (sweep_csr_kernel, one row per proposal)" "" \
               python3 bench.py --workload c4 --no-variants --no-cpu-baseline ;;
    c5) passes c5_csr "bench.py --workload c5 --no-variants --no-cpu-baseline (100 cities, CSR)" "" \
               python3 bench.py --workload c5 --no-variants --no-cpu-baseline ;;
    c5k) passes c5_1000_csr "bench.py --workload c5 --cities 1000 --replicas 256 --steps 2 --warmup 1 --no-variants --no-cpu-baseline" "" \
               python3 bench.py --workload c5 --cities 1000 --replicas 256 --steps 2 --warmup 1 --no-variants --no-cpu-baseline ;;
  esac || exit 1
done
echo collected
