#!/bin/bash
# Same-box A/B of the cached-local-field sweep: in-tree libsga.so against build/libsga_prev.so
# (profiles/build_variant.sh prev sweep_clf ""), C2a instance, int8 / fp32 / auto storage.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for rep in 1 2; do
  SGA_LIBRARY_PATH=$PWD/build/libsga_prev.so python profiles/clf_timing.py > gpurun_out/ab_clf_prev$rep.log 2>&1 || exit 1
  python profiles/clf_timing.py > gpurun_out/ab_clf_new$rep.log 2>&1 || exit 1
done
for rep in 1 2; do
  grep -h "sweeps\|cache=" gpurun_out/ab_clf_prev$rep.log | cut -c1-60 | paste - <(grep -h "sweeps\|cache=" gpurun_out/ab_clf_new$rep.log | awk '{print $4, $5}')
done > gpurun_out/ab_clf.txt
cat gpurun_out/ab_clf.txt
