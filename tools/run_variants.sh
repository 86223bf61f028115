#!/bin/bash
# tools/run_variants.sh NAME...  -> one bench line (kernel ms, Mreads/s) per variant library
cd "$(dirname "$0")/.."
for v in "$@"; do
  lib=$PWD/cammiq_amd/libcq_$v.so
  [ "$v" = main ] && lib=$PWD/cammiq_amd/libcammiq_hip.so
  CAMMIQ_LIB=$lib CAMMIQ_STAMPS_FILE=gpurun_out/stamps_$v.txt timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/bench_$v.json 2> gpurun_out/bench_$v.err || { echo "$v FAILED"; tail -5 gpurun_out/bench_$v.err; exit 1; }
  python - "$v" <<'PY'
import json,sys
v=sys.argv[1]
d=json.load(open(f"gpurun_out/bench_{v}.json"))
print(f"{v:10s} kernel_ms {d['roofline']['kernel_ms']:.3f}  step Mreads/s {d['value']:.0f}  nundet {d['outcome']['nundet']} cnt {d['outcome']['cnt_u_sum']}")
PY
done
