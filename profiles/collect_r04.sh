#!/bin/bash
# Round-4 rocprofv3 passes (run on the GPU box through gpurun; summaries via summarize_rocprof.py).
#   headline  bench.py C2a fp32 with --waves = the autotuner's pick of the committed line; stats + FETCH_SIZE + WRITE_SIZE
#   c3        bench.py --workload c3: stats, traffic AND the instruction counters its roofline block reads (VALU / SALU / LDS)
#   c4 | c5 | c5k  the graded one-row-per-proposal lines of C4, C5 at 100 and at 1000 cities: stats + FETCH_SIZE + WRITE_SIZE
#   c4cached  the cached-field sweep over CSR couplings on C4 (profiles/r04_c4_cached.py c4)
#   cached    the dense cached-field variant (ON): sweep_clfb_kernel, then sweep_clf_kernel (profiles/r04_cached_profile_run.py)
#   mixed     per-replica routing: both kernels of a mixed launch (profiles/r04_mixed_profile_run.py); kernel trace kept
#   energy    the all-replica field pass after the XCD-aware tile mapping: C2 fp32, n = 32768 fp32
# One counter set per --pmc pass, never with other trace domains; the program comes directly after `--`.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
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
        rocprofv3 --kernel-trace --pmc $set -d gpurun_out/prof_${tag}_x$i --output-format csv -- "$@" > gpurun_out/prof_${tag}_x$i.log 2>&1 || return 1
        ex="$ex --extra gpurun_out/prof_${tag}_x$i"
    done
    [ "$KEEP_TRACE" = 1 ] && cp $(find gpurun_out/prof_${tag}_stats -name "*_kernel_trace.csv" | head -1) gpurun_out/r04_${tag}_kernel_trace.csv
    find gpurun_out/prof_${tag}_* -name "*_kernel_trace.csv" -delete
    python3 profiles/summarize_rocprof.py --stats gpurun_out/prof_${tag}_stats --fetch gpurun_out/prof_${tag}_fetch \
        --write gpurun_out/prof_${tag}_write $ex --tag r04_${tag} --note "$note" &&
    cp profiles/r04_${tag}_* gpurun_out/ &&
    rm -rf gpurun_out/prof_${tag}_stats gpurun_out/prof_${tag}_fetch gpurun_out/prof_${tag}_write gpurun_out/prof_${tag}_x*
}
for t in "$@"; do
  case $t in
    headline)
      W=$(python3 -c "import json,re; d=json.load(open('${PICK_FROM:-gpurun_out/r04_bench_c2a_f32.json}')); print(re.search(r'waves_per_replica=(\d+)', d['config']['geometry']).group(1))")
      echo "autotuner's pick of the committed line: $W waves per replica"
      passes c2a_f32 "bench.py --waves $W --no-variants --no-cpu-baseline (the autotuner's pick of the committed line)" "" \
             python3 bench.py --waves $W --no-variants --no-cpu-baseline ;;
    c3) passes c3_csr "bench.py --workload c3 --no-cpu-baseline (several updates per step: sweep_csr_rows_kernel)" \
               "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD;SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SMEM" \
               python3 bench.py --workload c3 --no-cpu-baseline ;;
    c5i) passes c5_100_implicit "bench.py --workload c5 --implicit --no-cpu-baseline (sweep_tsp_par_kernel)" \
               "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
               python3 bench.py --workload c5 --implicit --no-cpu-baseline ;;
    c4) passes c4_csr "bench.py --workload c4 --no-variants --no-cpu-baseline (sweep_csr_kernel, one row per proposal)" "" \
               python3 bench.py --workload c4 --no-variants --no-cpu-baseline ;;
    c5) passes c5_csr "bench.py --workload c5 --no-variants --no-cpu-baseline (100 cities, CSR)" "" \
               python3 bench.py --workload c5 --no-variants --no-cpu-baseline ;;
    c5k) passes c5_1000_csr "bench.py --workload c5 --cities 1000 --replicas 256 --steps 2 --warmup 1 --no-variants --no-cpu-baseline" "" \
               python3 bench.py --workload c5 --cities 1000 --replicas 256 --steps 2 --warmup 1 --no-variants --no-cpu-baseline ;;
    c4cached) passes c4_cached "profiles/r04_c4_cached.py c4 (cache off / on / auto, 50 sweeps each)" "" python3 profiles/r04_c4_cached.py c4 ;;
    cached) passes c2a_cached "profiles/r04_cached_profile_run.py (field cache ON, int8 rows, 120 sweeps in launches of 10)" \
               "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD;SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SMEM" \
               python3 profiles/r04_cached_profile_run.py ;;
    mixed) KEEP_TRACE=1 passes c2a_mixed "profiles/r04_mixed_profile_run.py (int8 couplings, ladder 400 -> 0.1, routed by replica)" "" \
               python3 profiles/r04_mixed_profile_run.py ;;
    energy)
      N=10000 STORAGE=f32 passes energy_c2_f32 "profiles/energy_profile_run.py N=10000 f32" "" python3 profiles/energy_profile_run.py
      export STORAGE=f32 N=32768; passes energy_n32768_f32 "profiles/energy_profile_run.py N=32768 f32" "" python3 profiles/energy_profile_run.py
      unset STORAGE N ;;
  esac || exit 1
done
echo collected
