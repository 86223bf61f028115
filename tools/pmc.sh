#!/bin/bash
# tools/pmc.sh TAG COUNTER...   one rocprofv3 --pmc pass of a short bench run; prints per-launch means
# of the fast classify kernel.  (Counters only: never combined with sys/hip/hsa tracing.)
# BENCH_ARGS="--config 1" selects the workload (default: bench.py's own default, configs[2]).
set -uo pipefail
root="$(cd "$(dirname "$0")/.." && pwd)"
tag=$1; shift
out=$root/gpurun_out/pmc_$tag
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "$@" --output-format csv -d "$out" -- python3 "$root/bench.py" ${BENCH_ARGS:-} --steps 2 --warmup 1 --no-cpu-baseline --no-host-fed --no-calibrate > "$out/bench.json" 2> "$out/err.log" || { rc=$?; echo "pass $tag ($*) FAILED rc=$rc; last lines of err.log:"; tail -40 "$out/err.log"; exit $rc; }
python3 - "$out" <<'PY'
import csv,glob,sys,collections,re
out=sys.argv[1]
FAST=re.compile(r"classify_kernel(<\d+, \d+, false|ILi\d+ELi\d+ELb0E)")   # the fast classify kernel: SLOW = false is its third template argument
acc=collections.defaultdict(list)
for f in glob.glob(out+"/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if FAST.search(r["Kernel_Name"]):
            acc[r["Counter_Name"]].append((r["Dispatch_Id"],float(r["Counter_Value"])))
for k,v in sorted(acc.items()):
    d=collections.defaultdict(float)
    for i,x in v: d[i]+=x
    vals=list(d.values())
    print(f"{k:32s} launches {len(vals)}  mean/launch {sum(vals)/len(vals):.4g}")
PY
