#!/bin/bash
# C3 (and C2b-as-CSR) A/B between the in-tree library and a variant build (SGA_LIBRARY_PATH), interleaved, three times.
#   usage: ab_r05_c3_lib.sh build/libsga_<variant>.so
cd "$GRAFT_REPO_ROOT" || exit 1
lib=$1
out=gpurun_out/r05_ab_c3_$(basename "$lib" .so).txt
: > "$out"
for i in 1 2 3; do
  for arm in current "$lib"; do
    if [ "$arm" = current ]; then line=$(python3 bench.py --workload c3 --no-cpu-baseline 2>/dev/null)
    else line=$(SGA_LIBRARY_PATH=$arm python3 bench.py --workload c3 --no-cpu-baseline 2>/dev/null); fi
    echo "$arm $(echo "$line" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e attempts/s  %.3f ms/step  %.3f ms/launch  %s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['kernel_instantiation']))")" | tee -a "$out"
  done
done
