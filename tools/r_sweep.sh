#!/bin/bash
# tools/r_sweep.sh [LEN...]: reads per sub-tile (CAMMIQ_FAST_R = 8 / 4 / automatic) x read length on configs[1]'s index
cd "$(dirname "$0")/.."
for L in ${@:-100 150 250}; do
  for R in 8 4 auto; do
    if [ $R = auto ]; then unset CAMMIQ_FAST_R; else export CAMMIQ_FAST_R=$R; fi
    echo -n "len $L R $R: "
    python tools/kexp.py --timeout 200 --config 1 --steps 6 --extra "--read-len $L" tree | tail -1
  done
done
