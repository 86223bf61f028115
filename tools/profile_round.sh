#!/bin/bash
# tools/profile_round.sh  -- the rocprof evidence bench.py's roofline object cites (run on the GPU box):
#   1. rocprofv3 --kernel-trace --stats of a default-length bench run      -> gpurun_out/prof/kernel_stats.csv
#   2. rocprofv3 --pmc FETCH_SIZE, then --pmc WRITE_SIZE (separate passes) -> gpurun_out/prof/traffic.json
# Copy the results into profiles/ afterwards (see profiles/README.md).
set -euo pipefail
root="$(cd "$(dirname "$0")/.." && pwd)"
out=$root/gpurun_out/prof
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 "$root/bench.py" --steps 10 --warmup 2 --no-cpu-baseline > "$out/bench_under_rocprof.json" 2> "$out/trace.err"
cp "$(ls "$out"/trace/*/*kernel_stats.csv | head -1)" "$out/kernel_stats.csv"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d "$out/pmc_$c" -- python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$out/bench_$c.json" 2> "$out/pmc_$c.err"
done
python3 - "$out" <<'PY'
import csv,glob,json,sys,collections
out=sys.argv[1]; res={}
for c in ("FETCH_SIZE","WRITE_SIZE"):
    d=collections.defaultdict(float)
    for f in glob.glob(f"{out}/pmc_{c}/**/*counter_collection.csv",recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"]==c and ("Lb0" in r["Kernel_Name"] or "false" in r["Kernel_Name"]):
                d[r["Dispatch_Id"]]+=float(r["Counter_Value"])
    v=list(d.values())
    res[c+"_KB_per_launch"]=sum(v)/len(v); res[c+"_launches"]=len(v)
res["hbm_bytes_per_launch"]=(res["FETCH_SIZE_KB_per_launch"]+res["WRITE_SIZE_KB_per_launch"])*1024
json.dump(res,open(out+"/traffic.json","w"),indent=1); print(json.dumps(res))
PY
head -5 "$out/kernel_stats.csv"; cat "$out/bench_under_rocprof.json"
