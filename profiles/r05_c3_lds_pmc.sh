#!/bin/bash
# C3: is the LDS pipe (random int8 spin gathers: bank conflicts) a co-bottleneck of the several-updates-per-step kernel?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS" "SQ_ACTIVE_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_BUSY_CU_CYCLES"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set -d gpurun_out/prof_lds_$tag --output-format csv -- python3 bench.py --workload c3 --no-cpu-baseline > gpurun_out/prof_lds_$tag.log 2>&1 || { tail -3 gpurun_out/prof_lds_$tag.log; continue; }
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/prof_lds_*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sweep_csr_rows" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("gpurun_out/r05_c3_lds_pmc.txt", "w") as out:
    for k, v in sorted(agg.items()):
        line = f"{k} = {sum(v)/len(v):.5g} per launch ({len(v)} launches)"
        print(line); out.write(line + "\n")
PY
rm -rf gpurun_out/prof_lds_*
