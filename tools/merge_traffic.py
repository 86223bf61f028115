#!/usr/bin/env python3
"""tools/merge_traffic.py gpurun_out/prof_TAG [...]: fold the PMC traffic + gather ceiling measured by
tools/profile_round.sh into profiles/traffic_r02.json, keyed by the workload bench.py ran (config.workload_key)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(ROOT, "profiles", "traffic_r02.json")
db = json.load(open(dst)) if os.path.exists(dst) else {}
for d in sys.argv[1:]:
    t = json.load(open(os.path.join(d, "traffic.json")))
    ent = {k: t[k] for k in ("FETCH_SIZE_KB_per_launch", "WRITE_SIZE_KB_per_launch", "FETCH_SIZE_launches",
                             "WRITE_SIZE_launches", "row_stream_bytes", "hbm_bytes_per_launch", "table_GB")}
    ent["fetch_bytes_per_launch"] = t["FETCH_SIZE_KB_per_launch"] * 1024
    ent["kernel_ms_under_stats"] = t.get("kernel_ms_under_stats")
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
