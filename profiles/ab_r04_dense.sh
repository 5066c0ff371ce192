#!/bin/bash
# Round 4 A/B: does the replica-list indirection (r = rep_list ? rep_list[blockIdx.x] : blockIdx.x) cost the graded
# dense fp32 kernel anything?  build/libsga_prevdense.so = this tree with sweep_dense_f32.hip compiled against round
# 3's sweep_dense_impl.h.  Same box, alternating, fixed geometry (9 waves x 5 chunks).
for rep in 1 2 3; do
  for lib in "" build/libsga_prevdense.so; do
    SGA_LIBRARY_PATH=$([ -n "$lib" ] && echo $PWD/$lib) python bench.py --steps 20 --warmup 3 --waves 9 --no-variants --no-cpu-baseline 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('${lib:-current}', round(d['ms_per_step'],3), 'ms/step', round(d['roofline']['avg_launch_ms'],3), 'ms/launch', d['roofline']['kernel_instantiation'])"
  done
done
