for w in 1 3 4 6 8; do echo "== SGA_CLF_WAVES=$w"; SGA_CLF_WAVES=$w STORAGES=i8 timeout -k 10 120 python profiles/clf_timing.py 2>&1 | grep -E "sweeps|describe" ; done
