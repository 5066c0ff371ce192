#!/bin/bash
# The short-kernel stall (profiles/r02_experiments.md 13): N fresh `bench.py --workload c3` processes; for each
# the wall time of the 20 timed steps, the sum of their kernel events, and the host-side gaps between step
# ends that exceed 5 ms (the two exchange rounds wait for 7 and 10 queued sweeps: 28 and 40 ms are normal).
# Environment passes through (SGA_BENCH_NOEVENTS=1: no HIP timing events in the timed region).
for i in $(seq 1 ${1:-24}); do
  SGA_BENCH_DEBUG=1 python bench.py --workload c3 --no-cpu-baseline > gpurun_out/st.json 2> gpurun_out/st.err
  python - $i <<PY
import json,sys
d=json.load(open("gpurun_out/st.json"))
marks=[l for l in open("gpurun_out/st.err") if l.startswith("step end marks")]
m=[float(x) for x in marks[0].split(":")[1].split("|")[0].split()] if marks else []
gaps=[round(b-a,1) for a,b in zip([0]+m[:-1],m)]
print(sys.argv[1], "wall %.1f kernel %.1f"%(d["wall_ms_total"], d["kernel_ms_total"]), "host gaps>5ms:", [(i,g) for i,g in enumerate(gaps) if g>5], flush=True)
PY
done
