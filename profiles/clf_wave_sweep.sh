#!/bin/bash
# waves per replica of the cached-field sweep (SGA_CLF_WAVES) on the C2a instance, int8 rows
cd "$GRAFT_REPO_ROOT" || exit 1
for w in 1 2 3 4 6 8; do echo "== SGA_CLF_WAVES=$w"; SGA_CLF_WAVES=$w STORAGES=i8 timeout -k 10 120 python profiles/clf_timing.py 2>&1 | grep -E "sweeps" ; done
