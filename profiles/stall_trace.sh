#!/bin/bash
# Where does the 42 ms go?  Fresh `bench.py --workload c3` processes under rocprofv3 --kernel-trace until one shows
# the gap; its dispatch timeline (every kernel of the process, begin/end) is searched for idle stretches > 5 ms.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for i in $(seq 1 ${1:-16}); do
  rm -rf gpurun_out/kt
  rocprofv3 --kernel-trace -d gpurun_out/kt --output-format csv -- python3 bench.py --workload c3 --no-cpu-baseline > gpurun_out/kt.json 2> gpurun_out/kt.err
  python3 - $i <<'PY'
import json, sys, glob, csv
d = json.load(open("gpurun_out/kt.json"))
wall, kern = d["wall_ms_total"], d["kernel_ms_total"]
stalled = wall - kern > 10
print(sys.argv[1], "wall %.1f kernel %.1f" % (wall, kern), "STALLED" if stalled else "", flush=True)
if stalled:
    f = glob.glob("gpurun_out/kt/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Queue_Id", "?")) for r in csv.DictReader(open(f))))
    out = open("gpurun_out/r02_stall_timeline.txt", "w")
    t_end = rows[0][1]
    for k, (b, e, name, q) in enumerate(rows):
        gap = (b - t_end) / 1e6
        if gap > 5.0:
            out.write("idle %.1f ms before dispatch %d\n" % (gap, k))
            for j in range(max(0, k - 4), min(len(rows), k + 4)):
                bb, ee, nn, qq = rows[j]
                out.write("   %4d  start %+10.3f ms  dur %8.3f ms  queue %s  %s\n" % (j, (bb - b) / 1e6, (ee - bb) / 1e6, qq, nn))
        t_end = max(t_end, e)
    out.close()
    print(open("gpurun_out/r02_stall_timeline.txt").read(), flush=True)
    sys.exit(3)
PY
  [ $? -eq 3 ] && break
done
rm -rf gpurun_out/kt
