#!/bin/bash
# the cached-field variants (dense C2a, CSR C4) and the implicit TSP form: in-tree library against a variant build, interleaved
#   usage: ab_r05_variants.sh build/libsga_<variant>.so
cd "$GRAFT_REPO_ROOT" || exit 1
lib=$1
out=gpurun_out/r05_ab_variants_$(basename "$lib" .so).txt
: > "$out"
for i in 1 2; do
  for arm in current "$lib"; do
    if [ "$arm" = current ]; then unset SGA_LIBRARY_PATH; else export SGA_LIBRARY_PATH=$arm; fi
    a=$(python3 bench.py --no-configs --no-beyond-cache --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); v=d['variants']['cached_local_fields']; print('C2a cached sweeps 5-25 %.4e  after 100 %.4e  one per launch %.4e' % (v['value'], v['after_100_sweeps']['value'], v['after_100_sweeps']['one_sweep_per_launch']['value']))")
    b=$(python3 bench.py --workload c4 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); v=d['variants']['cached_local_fields']; print('C4 cached %.4e' % v['value'])")
    c=$(python3 bench.py --workload c5 --cities 1000 --replicas 256 --implicit --steps 3 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('c5_1000 implicit %.4e' % d['value'])")
    echo "$arm | $a | $b | $c" | tee -a "$out"
  done
done
