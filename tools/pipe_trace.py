#!/usr/bin/env python3
"""tools/pipe_trace.py FILE -- summary of one host-fed query traced by the library itself (CAMMIQ_PIPE_TRACE=FILE:
HIP events with timing on the copy / widen / compute streams + the host clock, cq_api.cpp PipeTrace).  Under rocprofv3's
tracer the kernels of different streams no longer overlap (the bracket grows from 26 to 32 ms), so the library's own
events are the timeline of record for the host-fed bracket."""
import sys


def summarise(path):
    """Timeline of one traced host-fed query (CAMMIQ_PIPE_TRACE): device marks are ms after the query's first device event."""
    dev, host = {}, {}
    for ln in open(path):
        kind, name, chunk, t = ln.split()
        (dev if kind == "dev" else host).setdefault(name, {})[int(chunk)] = float(t)
    med = lambda xs: sorted(xs)[len(xs) // 2] if xs else float("nan")
    nch = len(dev.get("h2d_begin", {}))
    cp = [dev["h2d_end"][c] - dev["h2d_begin"][c] for c in range(nch)]
    kn = [dev["kernel_end"][c] - dev["kernel_begin"][c] for c in range(nch)]
    print(f"chunks {nch}; H2D per chunk median {med(cp):.3f} ms, busy {sum(cp):.3f} ms, first begins +{dev['h2d_begin'][0]:.3f}, last ends +{dev['h2d_end'][nch - 1]:.3f}; "
          f"start-to-start median {med([dev['h2d_begin'][c + 1] - dev['h2d_begin'][c] for c in range(nch - 1)]):.3f}")
    gaps = [dev["h2d_begin"][c + 1] - dev["h2d_end"][c] for c in range(nch - 1)]
    print(f"copy-queue gaps: total {sum(gaps):.3f} ms, median {med(gaps):.3f}, max {max(gaps):.3f} (after chunk {gaps.index(max(gaps))})")
    print(f"classify (fast + slow-path kernel) per chunk median {med(kn):.3f} ms, busy {sum(kn):.3f} ms, first begins +{dev['kernel_begin'][0]:.3f}, last ends +{dev['kernel_end'][nch - 1]:.3f}; "
          f"start-to-start median {med([dev['kernel_begin'][c + 1] - dev['kernel_begin'][c] for c in range(nch - 1)]):.3f}")
    if "widen_begin" in dev:
        wd = [dev["widen_end"][c] - dev["widen_begin"][c] for c in range(nch)]
        lag = [dev["kernel_begin"][c] - dev["h2d_end"][c] for c in range(nch)]
        print(f"widen per chunk median {med(wd):.3f} ms; kernel begins {med(lag):.3f} ms (median) after its chunk's copy ended, max {max(lag):.3f}")
    if "narrow_begin" in dev:
        nb, ne = dev["narrow_begin"][0], dev["narrow_end"][0]
        print(f"tail: last kernel ends +{dev['kernel_end'][nch - 1]:.3f}; narrow kernel (its lanes write rcount's bytes into page-locked host memory) "
              f"+{nb:.3f} .. +{ne:.3f} ({ne - nb:.3f} ms)")
    h0 = host.get("slot_wait", {}).get(0, 0.0)
    for k in ("kernels_done", "pieces_copied", "widened", "query_done"):
        if k in host:
            print(f"host: {k} at {list(host[k].values())[0] - h0:.3f} ms after the first chunk was taken up")
    waits = [host["slot_free"][c] - host["slot_wait"][c] for c in range(nch)]
    print(f"host waited for a free slot {sum(waits):.3f} ms in all (max {max(waits):.3f} at chunk {waits.index(max(waits))})")
    print("per chunk: h2d_begin h2d_end kernel_begin kernel_end")
    for c in range(nch):
        print(f"  {c:3d} {dev['h2d_begin'][c]:8.3f} {dev['h2d_end'][c]:8.3f} {dev['kernel_begin'][c]:8.3f} {dev['kernel_end'][c]:8.3f}")



if __name__ == "__main__":
    summarise(sys.argv[1])
