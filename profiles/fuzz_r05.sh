#!/bin/bash
# round-4 fuzz record: fuzz_parity.py with the field cache ON / AUTO, the several-accepts-per-round form, the matrix-core energies and the CSR pair
# look-ahead in the mix (usage: bash profiles/fuzz_r05.sh <seconds> <seed> [big seconds])
cd "$GRAFT_REPO_ROOT" || exit 1
secs=${1:-380}; seed=${2:-301}; big=${3:-280}
timeout -k 10 $((secs + 60)) python profiles/fuzz_parity.py $secs $seed > gpurun_out/fuzz_r05_a.log 2>&1; tail -1 gpurun_out/fuzz_r05_a.log
FUZZ_BIG=1 timeout -k 10 $((big + 60)) python profiles/fuzz_parity.py $big $((seed + 1)) > gpurun_out/fuzz_r05_b.log 2>&1; tail -1 gpurun_out/fuzz_r05_b.log
echo "mismatches / errors: $(cat gpurun_out/fuzz_r05_a.log gpurun_out/fuzz_r05_b.log | grep -c 'MISMATCH\|ERROR')"
