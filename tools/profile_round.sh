#!/bin/bash
# tools/profile_round.sh TAG [bench.py args...]  -- the rocprof evidence bench.py's roofline object cites
# (run on the GPU box; TAG names the workload, e.g. "cfg2" for the default run, "cfg1" with `--config 1`).
# Everything comes from ONE lease, so that the un-profiled line, the --stats mean and the clock the chip held can
# be compared (boxes of the pool differ by up to ~12 % in kernel time):
#   0. the un-profiled bench line (HIP-event kernel ms)         -> gpurun_out/prof_TAG/bench_unprofiled.json
#      + the in-kernel clock of a diagnostic -DCQ_STAMPS=1 build (variants/libcammiq_stamps.so, built here by
#        tools/build_variant.sh stamps -DCQ_STAMPS=1): shader cycles / 100 MHz ticks over the waves' main
#        loops (MI355X_MICROARCH.md, DVFS item 6)               -> gpurun_out/prof_TAG/clock.txt
#   1. rocprofv3 --kernel-trace --stats of a bench run          -> gpurun_out/prof_TAG/kernel_stats.csv
#   2. rocprofv3 --pmc FETCH_SIZE, then --pmc WRITE_SIZE (separate passes: FETCH_SIZE takes 3 of the 4 TCC
#      slots, WRITE_SIZE 2 -- MI355X_MICROARCH.md "rocprofv3 PMC slots") -> gpurun_out/prof_TAG/traffic.json
#   3. tools/gather_bench at the workload's table size          -> gpurun_out/prof_TAG/gather.txt
# Copy the results into profiles/ afterwards (see profiles/README.md); tools/merge_traffic.py folds
# traffic.json + gather.txt into profiles/traffic_rNN.json under the workload's key (--round rNN).
set -euo pipefail
root="$(cd "$(dirname "$0")/.." && pwd)"
tag=$1; shift
out=$root/gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
python3 "$root/bench.py" "$@" --steps 6 --warmup 2 --no-cpu-baseline --no-host-fed > "$out/bench_unprofiled.json" 2> "$out/bench_unprofiled.err" || { tail -20 "$out/bench_unprofiled.err"; exit 1; }
if [ -f "$root/variants/libcammiq_stamps.so" ]; then
  CAMMIQ_LIB="$root/variants/libcammiq_stamps.so" CAMMIQ_STAMPS_FILE="$out/clock.txt" python3 "$root/bench.py" "$@" --steps 6 --warmup 2 --no-cpu-baseline --no-host-fed > "$out/bench_stamps.json" 2> "$out/bench_stamps.err" || tail -5 "$out/bench_stamps.err"
fi
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 "$root/bench.py" "$@" --steps 6 --warmup 2 --no-cpu-baseline --no-host-fed > "$out/bench_under_rocprof.json" 2> "$out/trace.err"
cp "$(ls "$out"/trace/*/*kernel_stats.csv | head -1)" "$out/kernel_stats.csv"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d "$out/pmc_$c" -- python3 "$root/bench.py" "$@" --steps 3 --warmup 1 --no-cpu-baseline --no-host-fed > "$out/bench_$c.json" 2> "$out/pmc_$c.err"
done
python3 - "$out" <<'PY'
import csv,glob,json,sys,collections,re,os
out=sys.argv[1]; res={}
FAST=re.compile(r"classify_kernel(<\d+, \d+, false|ILi\d+ELi\d+ELb0E)")   # demangled or mangled: classify_kernel<R, CAP, SLOW = false, ...>
line=json.loads(open(out+"/bench_under_rocprof.json").read().strip().splitlines()[-1])
cfg=line["config"]
for c in ("FETCH_SIZE","WRITE_SIZE"):
    d=collections.defaultdict(float)
    for f in sorted(glob.glob(f"{out}/pmc_{c}/**/*counter_collection.csv",recursive=True), key=os.path.getmtime)[-1:]:   # this pass's file (a merged local copy may hold older ones)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"]==c and FAST.search(r["Kernel_Name"]):   # the fast classify kernel only (SLOW = false: the third template argument)
                d[r["Dispatch_Id"]]+=float(r["Counter_Value"])
    v=list(d.values())
    res[c+"_KB_per_launch"]=sum(v)/len(v); res[c+"_launches"]=len(v)
# gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies a wide coalesced streaming read at half its
# bytes.  The only such stream here is the packed rows (16 B per lane, read once): add the missing half.  The
# rest of the kernel's reads are random 64-byte bucket / 16-byte node reads, calibrated at x1.0 in round 1
# (profiles/r01_v1_traffic.json: 1.5e9 known bucket reads x 64 B).
rows=cfg["reads_per_gpu"]*((cfg["read_len"]+15)//16*4+1)   # rows of ceil(len/16) words + one length byte per read
res["row_stream_bytes"]=rows
res["hbm_bytes_per_launch"]=(res["FETCH_SIZE_KB_per_launch"]+res["WRITE_SIZE_KB_per_launch"])*1024+0.5*rows
res["workload_key"]=cfg["workload_key"]; res["table_GB"]=cfg["table_GB"]
res["kernel_ms_under_stats"]=line["roofline"]["kernel_ms"]
res["kernel"]=line["roofline"].get("kernel")
# the --stats mean of the fast classify kernel itself (what a reader recomputes the fraction from)
for r in csv.DictReader(open(out+"/kernel_stats.csv")):
    if FAST.search(r["Name"]):
        res["kernel_ms_rocprof_stats"]=float(r["AverageNs"])/1e6; res["rocprof_stats_calls"]=int(r["Calls"]); res["rocprof_stats_name"]=r["Name"]
up=out+"/bench_unprofiled.json"
if os.path.exists(up):
    try:
        u=json.loads(open(up).read().strip().splitlines()[-1]); res["kernel_ms_unprofiled_same_lease"]=u["roofline"]["kernel_ms"]; res["value_unprofiled_same_lease"]=u["value"]
    except Exception: pass
ck=out+"/clock.txt"
if os.path.exists(ck):
    for l in open(ck):
        if l.startswith("in_kernel_clock_MHz"): res["in_kernel_clock_MHz"]=float(l.split()[1])
json.dump(res,open(out+"/traffic.json","w"),indent=1); print(json.dumps(res))
PY
mib=$(python3 -c "import json;print(int(json.load(open('$out/traffic.json'))['table_GB']*1e9/1048576))")
"$root/tools/gather_bench" "$mib" > "$out/gather.txt" 2>&1 || true
head -5 "$out/kernel_stats.csv"; cat "$out/bench_under_rocprof.json"; cat "$out/gather.txt"
