cd "$GRAFT_REPO_ROOT"
timeout -k 10 420 python profiles/fuzz_parity.py 380 301 > gpurun_out/fuzz_r03_a.log 2>&1; tail -3 gpurun_out/fuzz_r03_a.log
FUZZ_BIG=1 timeout -k 10 320 python profiles/fuzz_parity.py 280 302 > gpurun_out/fuzz_r03_b.log 2>&1; tail -3 gpurun_out/fuzz_r03_b.log
grep -c "MISMATCH\|ERROR" gpurun_out/fuzz_r03_a.log gpurun_out/fuzz_r03_b.log
