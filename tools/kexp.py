#!/usr/bin/env python3
"""tools/kexp.py [--config N] NAME...: same-box A/B of kernel variants (variants/libcammiq_NAME.so, built by
tools/build_variant.sh; "tree" = the in-tree library).  Runs bench.py once per variant with CAMMIQ_LIB set and
prints kernel ms / Mreads/s per variant; each run carries bench.py's own counter sanity checks, and the
counters of every variant must equal those of the first one."""
import argparse
import json
import os
import signal
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--config", default="1")
ap.add_argument("--steps", default="10")
ap.add_argument("--extra", default="", help="extra bench.py arguments, one string (write --extra=\"--flag ...\" when it starts with a dash)")
ap.add_argument("--timeout", type=float, default=600.0,
                help="seconds per variant; the bench child runs in its own session and its whole group is killed")
ap.add_argument("names", nargs="+")
a = ap.parse_args()
first = None
for name in a.names:
    env = dict(os.environ)
    lib_name, *assigns = name.split(":")     # NAME[:VAR=VALUE...] -- same library, different environment knobs
    for kv in assigns:
        k, _, v = kv.partition("=")
        env[k] = v
    name_full, name = name, lib_name
    if name != "tree":
        env["CAMMIQ_LIB"] = os.path.join(ROOT, "variants", f"libcammiq_{name}.so")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", a.config, "--steps", a.steps, "--warmup", "2",
           "--no-cpu-baseline", "--no-host-fed"] + a.extra.split()
    # own session: on a timeout the whole group goes (the child that holds the GPU must not outlive this point
    # and keep the device while the next variant starts)
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, err = p.communicate(timeout=a.timeout)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)
        p.communicate()
        print(f"{name_full:40s} FAILED: no result within {a.timeout:.0f} s (process group killed)", flush=True)
        continue
    r = subprocess.CompletedProcess(cmd, p.returncode, out, err)
    if r.returncode != 0:
        print(f"{name_full:40s} FAILED rc={r.returncode}: {r.stderr[-400:]}", flush=True)
        continue
    j = json.loads(r.stdout.strip().splitlines()[-1])
    oc = {k: j["outcome"][k] for k in ("nundet", "nconf", "cnt_u_sum", "cnt_d_sum", "rcount_sum")}
    if first is None:
        first = oc
    same = "same counters" if oc == first else f"COUNTERS DIFFER {oc} vs {first}"
    print(f"{name_full:40s} kernel {j['roofline']['kernel_ms']:8.4f} ms  slow {j['roofline']['slow_path_kernel_ms']:.4f} ms  "
          f"step {j['ms_per_step']:8.4f} ms  value {j['value']:9.2f} Mreads/s  {same}", flush=True)
