#!/bin/bash
# C3 A/B of round 5 (profiles/r05_experiments.md): the same bench line with an engine option set through its
# environment default, interleaved, three times.   usage: ab_r05_c3.sh VAR=VALUE [VAR=VALUE ...]
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r05_ab_c3_$(echo "$*" | tr ' =' '__').txt
: > "$out"
for i in 1 2 3; do
  for arm in base "$@"; do
    if [ "$arm" = base ]; then
      line=$(python3 bench.py --workload c3 --no-cpu-baseline 2>/dev/null)
    else
      line=$(env "$arm" python3 bench.py --workload c3 --no-cpu-baseline 2>/dev/null)
    fi
    echo "$arm $(echo "$line" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4e attempts/s  %.3f ms/step  %.3f ms/launch  %s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['kernel_instantiation']))")" | tee -a "$out"
  done
done
