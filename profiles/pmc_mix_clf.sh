#!/bin/bash
# instruction mix of the cached-field sweep kernel over the first sweep(s) of the C2a instance:
#   bash profiles/pmc_mix_clf.sh <tag> [WARMUP STEPS]      (profiles/clf_profile_run.py is what runs)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; export WARMUP=${2:-0} STEPS=${3:-1}
rm -rf gpurun_out/mix_${tag}_*
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD -d gpurun_out/mix_${tag}_1 --output-format csv -- python3 profiles/clf_profile_run.py > gpurun_out/mix_${tag}_1.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SMEM -d gpurun_out/mix_${tag}_2 --output-format csv -- python3 profiles/clf_profile_run.py > gpurun_out/mix_${tag}_2.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_I8 SQ_WAVES -d gpurun_out/mix_${tag}_3 --output-format csv -- python3 profiles/clf_profile_run.py > gpurun_out/mix_${tag}_3.log 2>&1
python3 - "$tag" <<'PY'
import csv, glob, collections, sys
tag = sys.argv[1]
for d in (1, 2, 3):
    for f in glob.glob(f"gpurun_out/mix_{tag}_{d}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "sweep_" in r["Kernel_Name"]:
                acc[(r["Kernel_Name"][:64], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(acc.items()):
            print(k, c, "per launch %.4g" % (sum(v) / len(v)), "launches", len(v))
PY
grep -h "acceptance" gpurun_out/mix_${tag}_1.log
rm -rf gpurun_out/mix_${tag}_1 gpurun_out/mix_${tag}_2 gpurun_out/mix_${tag}_3
