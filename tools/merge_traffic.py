#!/usr/bin/env python3
"""tools/merge_traffic.py [--round rNN] gpurun_out/prof_TAG [...]: fold the PMC traffic, the rocprofv3 --stats mean of
the classify kernel, the in-kernel clock and the gather ceiling measured by tools/profile_round.sh (one lease) into
profiles/traffic_rNN.json (default r03), keyed by the workload bench.py ran (config.workload_key); also copies the
lease's kernel_stats.csv / bench lines into profiles/ under that round's names."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import shutil
args = sys.argv[1:]
rnd = "r03"
if args and args[0] == "--round":
    rnd, args = args[1], args[2:]
dst = os.path.join(ROOT, "profiles", f"traffic_{rnd}.json")
db = json.load(open(dst)) if os.path.exists(dst) else {}
for d in args:
    d = d.rstrip("/")
    tag = os.path.basename(d).replace("prof_", "")
    t = json.load(open(os.path.join(d, "traffic.json")))
    ent = {k: t[k] for k in ("FETCH_SIZE_KB_per_launch", "WRITE_SIZE_KB_per_launch", "FETCH_SIZE_launches",
                             "WRITE_SIZE_launches", "row_stream_bytes", "hbm_bytes_per_launch", "table_GB")}
    ent["fetch_bytes_per_launch"] = t["FETCH_SIZE_KB_per_launch"] * 1024
    ent["kernel_ms_under_stats"] = t.get("kernel_ms_under_stats")
    for k in ("kernel", "kernel_ms_rocprof_stats", "rocprof_stats_calls", "rocprof_stats_name", "kernel_ms_unprofiled_same_lease",
              "value_unprofiled_same_lease", "in_kernel_clock_MHz"):
        if k in t:
            ent[k] = t[k]
    ent["kernel_stats_file"] = f"profiles/{rnd}_{tag}_kernel_stats.csv"
    for src, name in (("kernel_stats.csv", "kernel_stats.csv"), ("bench_under_rocprof.json", "bench_under_rocprof.json"),
                      ("bench_unprofiled.json", "bench_unprofiled_same_lease.json"), ("clock.txt", "in_kernel_clock.txt"),
                      ("gather.txt", "gather_ceiling.txt")):
        if os.path.exists(os.path.join(d, src)):
            shutil.copy(os.path.join(d, src), os.path.join(ROOT, "profiles", f"{rnd}_{tag}_{name}"))
    ent["note"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes of bench.py (tools/profile_round.sh); "
                   "KB x 1024; + half of the coalesced row stream, which gfx950 tallies at 1/2 (MI355X_MICROARCH.md); "
                   "random bucket reads calibrated x1.0 (profiles/r01_v1_traffic.json)")
    gp = os.path.join(d, "gather.txt")
    best = 0.0
    if os.path.exists(gp):
        for line in open(gp):
            m = re.search(r"W=16 B.*?:\s*([\d.]+) G accesses/s", line)
            if m:
                best = max(best, float(m.group(1)))
    if best:
        ent["gather_ceiling_lines_per_s"] = best * 1e9
        ent["gather_ceiling_note"] = (f"tools/gather_bench {int(t['table_GB'] * 1e9 / 1048576)}: random 16-byte loads (the probe "
                                      f"loop's access width) from a table of this workload's size, best of 4/6/8 blocks per CU: "
                                      f"{best:.1f} G 64-B lines/s")
    db[t["workload_key"]] = ent
json.dump(db, open(dst, "w"), indent=1, sort_keys=True)
print(json.dumps(db, indent=1))
