#!/bin/bash
# C3: what does the several-updates-per-step kernel pay for one more instruction of each class per step?
# (build/libsga_sens_*.so: sweep_csr_rows.hip rebuilt with -DROWS_SENS_<CLASS>=N; all replicas cold and the bench ladder)
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r05_c3_sensitivity.txt
: > "$out"
for i in 1 2; do
  for lib in current build/libsga_sens_valu2.so build/libsga_sens_valu4.so build/libsga_sens_salu.so build/libsga_sens_lds.so; do
    if [ "$lib" = current ]; then r=$(python3 profiles/r05_c3_regimes.py 2>/dev/null | grep -E "ladder|T = 0.5"); else r=$(SGA_LIBRARY_PATH=$lib python3 profiles/r05_c3_regimes.py 2>/dev/null | grep -E "ladder|T = 0.5"); fi
    echo "$lib" | tee -a "$out"; echo "$r" | cut -c1-80 | tee -a "$out"
  done
done
