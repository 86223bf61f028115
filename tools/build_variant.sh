#!/bin/bash
# tools/build_variant.sh NAME [KERNEL_SOURCE] [extra hipcc flags...]
# Builds variants/libcammiq_NAME.so: the current library with cq_kernels.hip replaced by KERNEL_SOURCE
# (default: the tree's own) and/or extra -D flags -- for same-box A/B timing through CAMMIQ_LIB (tools/kexp.py).
# variants/*.so are git-ignored but travel to the GPU box.
set -euo pipefail
root="$(cd "$(dirname "$0")/.." && pwd)"
name=$1; shift
src=$root/cammiq_amd/csrc/cq_kernels.hip
if [ $# -gt 0 ] && [ -f "$1" ]; then src=$1; shift; fi
b=$(mktemp -d)
cp "$src" "$b/cq_kernels.hip"
cd "$root/cammiq_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"$root/cammiq_amd/csrc" "$@" -c "$b/cq_kernels.hip" -o "$b/cq_kernels.o"
mkdir -p "$root/variants"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o "$root/variants/libcammiq_$name.so" build/cq_decode.o build/cq_layout.o build/cq_pack.o build/cq_cache.o build/cq_api.o build/cq_layout_gpu.o "$b/cq_kernels.o" -lpthread -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
rm -rf "$b"
echo "built variants/libcammiq_$name.so"
