for v in "" clfb_l4 clfb_l6 clfb_l8 clfb_l12 ""; do
  if [ -n "$v" ]; then export SGA_LIBRARY_PATH=build/libsga_$v.so; else unset SGA_LIBRARY_PATH; fi
  echo "== ${v:-default build}"; timeout -k 10 200 python profiles/r04_clfb_timing.py 0 2>&1 | grep "batched w=0    10" | cut -c1-330
done
