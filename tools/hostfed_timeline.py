#!/usr/bin/env python3
"""tools/hostfed_timeline.py DIR: where the time of one host-fed query of a traced bench run (tools/hostfed_timeline.sh)
went -- the H2D chunk copies, the classify / widen kernels between them and the D2H of rcount behind them (this
runtime copies device -> pinned host with a shader, `__amd_rocclr_copyBuffer`, so it shows up as a kernel)."""
import csv
import glob
import sys

d = sys.argv[1]
kern, h2d = [], []
for f in glob.glob(d + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        kern.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
for f in glob.glob(d + "/trace/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "HOST_TO_DEVICE" in r["Direction"]:
            h2d.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
kern.sort(); h2d.sort()
big = [c for c in h2d if c[1] - c[0] > 300_000]            # the row chunks (the 2 MB of lengths take 0.08 ms)
# runs of chunk copies less than 3 ms apart = one query each; take the last run of at least 4
runs, cur = [], [big[0]]
for c in big[1:]:
    if c[0] - cur[-1][1] < 3_000_000:
        cur.append(c)
    else:
        runs.append(cur); cur = [c]
runs.append(cur)
q = [r for r in runs if len(r) >= 4][-1]
t0 = q[0][0]
d2h = [k for k in kern if "copyBuffer" in k[2] and k[0] > q[-1][0] and k[1] - k[0] > 300_000][:2]
end = d2h[-1][1] if d2h else q[-1][1]
ks = [k for k in kern if t0 <= k[0] <= end]
ms = lambda x: x / 1e6
busy = lambda xs: ms(sum(x[1] - x[0] for x in xs))
med = lambda xs: sorted(xs)[len(xs) // 2]
print(f"query window {ms(end - t0):.3f} ms, {len(q)} chunks")
print(f"H2D rows: {med([ms(c[1] - c[0]) for c in q]):.3f} ms per chunk (median), busy {busy(q):.3f} ms, last ends +{ms(q[-1][1] - t0):.3f}; "
      f"start-to-start period {med([ms(b[0] - a[0]) for a, b in zip(q, q[1:])]):.3f} ms")
for name in ("classify_kernel<8", "classify_kernel<4", "classify_kernel<1", "widen_rows", "fillBuffer"):
    sel = [k for k in ks if name in k[2]]
    if sel:
        print(f"{name:18s} {len(sel):3d} launches, {med([ms(k[1] - k[0]) for k in sel]):.3f} ms each (median), busy {busy(sel):.3f} ms, "
              f"first +{ms(sel[0][0] - t0):.3f}, last ends +{ms(sel[-1][1] - t0):.3f}")
fast = [k for k in ks if "classify_kernel<8" in k[2] or "classify_kernel<4" in k[2]]
if len(fast) > 2:
    print(f"classify start-to-start period {med([ms(b[0] - a[0]) for a, b in zip(fast, fast[1:])]):.3f} ms")
for k in d2h:
    print(f"D2H (copyBuffer)  +{ms(k[0] - t0):.3f} .. +{ms(k[1] - t0):.3f} ms ({ms(k[1] - k[0]):.3f} ms)")
