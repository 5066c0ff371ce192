#!/bin/bash
# What do the SQ VALU counters say about instructions whose issue cost is known?  (profiles/probes/valu_issue_probe.hip:
# v_add / v_fma / v_and hold a SIMD for 2 cycles, VOP3 / DPP / compare / dot4 forms for 4.)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES -d gpurun_out/prof_probe \
    --output-format csv -- ./profiles/probes/valu_issue_probe > gpurun_out/r05_probe_pmc.log 2>&1 || { tail -5 gpurun_out/r05_probe_pmc.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/prof_probe/**/*_counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[(r["Kernel_Name"], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("gpurun_out/r05_probe_pmc.txt", "w") as out:
    for (k, g), c in agg.items():
        line = f"{k[:40]} grid={g} " + " ".join(f"{n}={sum(v)/len(v):.4g}" for n, v in sorted(c.items()))
        print(line)
        out.write(line + "\n")
PY
rm -rf gpurun_out/prof_probe
