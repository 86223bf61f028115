#!/bin/bash
# tools/hostfed_timeline.sh [bench args]: kernel + memory-copy timeline of bench.py's host-fed leg (rocprofv3
# --kernel-trace --memory-copy-trace, no counters), summarised per stream by tools/hostfed_timeline.py.
set -euo pipefail
root="$(cd "$(dirname "$0")/.." && pwd)"
out=$root/gpurun_out/hostfed_tl
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$out/trace" -- python3 "$root/bench.py" "$@" --steps 1 --warmup 0 --no-cpu-baseline --multi-leg off > "$out/bench.json" 2> "$out/trace.err"
python3 "$root/tools/hostfed_timeline.py" "$out"
