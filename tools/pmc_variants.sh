#!/bin/bash
# tools/pmc_variants.sh NAME...  -- instruction counts and wait cycles per launch of knock-out / experiment builds
# (variants/libcammiq_NAME.so; "tree" = the in-tree library), one rocprofv3 --pmc pass each; also the un-profiled
# kernel time of each.  BENCH_ARGS selects the workload.  Output: gpurun_out/pmc_variants.txt
root="$(cd "$(dirname "$0")/.." && pwd)"
o=$root/gpurun_out/pmc_variants.txt
: > "$o"
for v in "$@"; do
  if [ "$v" = tree ]; then unset CAMMIQ_LIB; else export CAMMIQ_LIB=$root/variants/libcammiq_$v.so; fi
  echo "== $v" >> "$o"
  python3 "$root/bench.py" ${BENCH_ARGS:-} --steps 6 --warmup 2 --no-cpu-baseline --no-host-fed 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kernel_ms', j['roofline']['kernel_ms'], 'slow_ms', j['roofline']['slow_path_kernel_ms'])" >> "$o" 2>&1
  "$root/tools/pmc.sh" "v_$v" SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY >> "$o" 2>&1
done
cat "$o"
