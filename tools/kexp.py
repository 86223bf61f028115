#!/usr/bin/env python3
"""tools/kexp.py [--config N] NAME...: same-box A/B of kernel variants (variants/libcammiq_NAME.so, built by
tools/build_variant.sh; "tree" = the in-tree library).  Runs bench.py once per variant with CAMMIQ_LIB set and
prints kernel ms / Mreads/s per variant; each run carries bench.py's own counter sanity checks, and the
counters of every variant must equal those of the first one."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--config", default="1")
ap.add_argument("--steps", default="10")
ap.add_argument("--extra", default="", help="extra bench.py arguments, one string")
ap.add_argument("names", nargs="+")
a = ap.parse_args()
first = None
for name in a.names:
    env = dict(os.environ)
    if name != "tree":
        env["CAMMIQ_LIB"] = os.path.join(ROOT, "variants", f"libcammiq_{name}.so")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", a.config, "--steps", a.steps, "--warmup", "2",
           "--no-cpu-baseline", "--no-host-fed"] + a.extra.split()
    r = subprocess.run(cmd, env=env, capture_output=True, text=True)
    if r.returncode != 0:
        print(f"{name:24s} FAILED rc={r.returncode}: {r.stderr[-400:]}", flush=True)
        continue
    j = json.loads(r.stdout.strip().splitlines()[-1])
    oc = {k: j["outcome"][k] for k in ("nundet", "nconf", "cnt_u_sum", "cnt_d_sum", "rcount_sum")}
    if first is None:
        first = oc
    same = "same counters" if oc == first else f"COUNTERS DIFFER {oc} vs {first}"
    print(f"{name:24s} kernel {j['roofline']['kernel_ms']:8.4f} ms  slow {j['roofline']['slow_path_kernel_ms']:.4f} ms  "
          f"step {j['ms_per_step']:8.4f} ms  value {j['value']:9.2f} Mreads/s  {same}", flush=True)
