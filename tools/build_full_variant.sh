#!/bin/bash
# tools/build_full_variant.sh NAME [extra compiler flags...]
# Builds variants/libcammiq_NAME.so with EVERY source of the library compiled with the extra flags -- for experiments
# that change what host layout and kernel share (cq_device.h), e.g. -DCQ_RUN_SLOTS=4.  (tools/build_variant.sh
# only recompiles the kernels.)  Same-box A/B through CAMMIQ_LIB / tools/kexp.py.
set -euo pipefail
root="$(cd "$(dirname "$0")/.." && pwd)"
name=$1; shift
b=$(mktemp -d)
cd "$root/cammiq_amd/csrc"
for f in cq_decode cq_layout cq_pack cq_cache; do
  g++ -O3 -std=c++17 -fPIC "$@" -c $f.cpp -o "$b/$f.o" &
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c cq_api.cpp -o "$b/cq_api.o" &
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c cq_kernels.hip -o "$b/cq_kernels.o" &
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c cq_layout_gpu.hip -o "$b/cq_layout_gpu.o" &
wait
mkdir -p "$root/variants"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o "$root/variants/libcammiq_$name.so" "$b"/*.o -lpthread -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
rm -rf "$b"
echo "built variants/libcammiq_$name.so"
