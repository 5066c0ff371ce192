#!/bin/bash
# Round-3 rocprofv3 passes (run on the GPU box through gpurun; summaries via summarize_rocprof.py).
#   headline  bench.py C2a fp32 with --waves = the autotuner's pick ON THIS BOX (read from a first plain
#             run), so that the traced instantiation is the timed one; stats + FETCH_SIZE + WRITE_SIZE
#   c3        bench.py --workload c3 (CSR, degree ~32, 4096 replicas)
#   cached    the cached-local-field variant on the same instance (profiles/clf_profile_run.py)
#   energy    the all-replica field pass: C2 fp32 / int8, n = 32768 fp32, and the per-replica kernel (A/B)
# One counter per --pmc pass, never with other trace domains; the program comes directly after `--`.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
passes() {  # tag, note, program args...
    tag=$1; note=$2; shift 2
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${tag}_stats --output-format csv -- "$@" > gpurun_out/prof_${tag}_stats.log 2>&1 &&
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/prof_${tag}_fetch --output-format csv -- "$@" > gpurun_out/prof_${tag}_fetch.log 2>&1 &&
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/prof_${tag}_write --output-format csv -- "$@" > gpurun_out/prof_${tag}_write.log 2>&1 &&
    find gpurun_out/prof_${tag}_* -name "*_kernel_trace.csv" -delete &&
    python3 profiles/summarize_rocprof.py --stats gpurun_out/prof_${tag}_stats --fetch gpurun_out/prof_${tag}_fetch \
        --write gpurun_out/prof_${tag}_write --tag r03_${tag} --note "$note" &&
    cp profiles/r03_${tag}_* gpurun_out/ &&
    rm -rf gpurun_out/prof_${tag}_stats gpurun_out/prof_${tag}_fetch gpurun_out/prof_${tag}_write
}
for t in "$@"; do
  case $t in
    headline)
      # the pick of the committed bench line (collect_r03_all.sh passes it), else of a plain run here
      if [ -z "$PICK_FROM" ]; then
        python3 bench.py --no-variants --no-cpu-baseline > gpurun_out/r03_pick.json 2> gpurun_out/r03_pick.err || exit 1
        PICK_FROM=gpurun_out/r03_pick.json
      fi
      W=$(python3 -c "import json,re; d=json.load(open('$PICK_FROM')); print(re.search(r'waves_per_replica=(\d+)', d['config']['geometry']).group(1))")
      echo "autotuner's pick on this box: $W waves per replica"
      passes c2a_f32 "bench.py --waves $W --no-variants --no-cpu-baseline (the autotuner's pick on this box)" \
             python3 bench.py --waves $W --no-variants --no-cpu-baseline ;;
    c3) passes c3_csr "bench.py --workload c3 --no-cpu-baseline (several updates per step: sweep_csr_rows_kernel)" \
               python3 bench.py --workload c3 --no-cpu-baseline ;;
    cached) passes c2a_cached "profiles/clf_profile_run.py (3 + 20 sweeps, exchange every 10)" python3 profiles/clf_profile_run.py ;;
    energy)
      N=10000 STORAGE=f32 passes energy_c2_f32 "profiles/energy_profile_run.py N=10000 f32" python3 profiles/energy_profile_run.py
      export STORAGE=i8; passes energy_c2_i8 "profiles/energy_profile_run.py N=10000 i8" python3 profiles/energy_profile_run.py
      export STORAGE=f32 N=32768; passes energy_n32768_f32 "profiles/energy_profile_run.py N=32768 f32" python3 profiles/energy_profile_run.py
      export N=10000 SGA_NO_MFMA_ENERGY=1; passes energy_c2_f32_per_replica "profiles/energy_profile_run.py N=10000 f32, SGA_NO_MFMA_ENERGY=1 (round-2 kernel)" python3 profiles/energy_profile_run.py
      unset SGA_NO_MFMA_ENERGY STORAGE N ;;
  esac || exit 1
done
echo collected
