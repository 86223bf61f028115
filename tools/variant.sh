#!/bin/bash
# Build an experimental variant of the library: tools/variant.sh NAME "-DFOO=1 ..."
# -> cammiq_amd/libcq_NAME.so (select with CAMMIQ_LIB=...).  Fails loudly; never reuses objects.
set -euo pipefail
cd "$(dirname "$0")/../cammiq_amd/csrc"
name=$1; shift
flags="${*:-}"
make -s build/cq_decode.o build/cq_layout.o build/cq_pack.o build/cq_cache.o build/cq_api.o
rm -f build/k_$name.o ../libcq_$name.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wextra -Wno-unused-parameter $flags -c cq_kernels.hip -o build/k_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../libcq_$name.so build/cq_decode.o build/cq_layout.o build/cq_pack.o build/cq_cache.o build/cq_api.o build/k_$name.o -lpthread
echo built libcq_$name.so
