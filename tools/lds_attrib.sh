#!/bin/bash
# tools/lds_attrib.sh -- LDS bank conflicts by phase: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of the full kernel and of
# the knock-out builds (exp9: no probe loop; exp8: probe loop without bucket loads / compares; exp1: no exact lookups,
# hence no hits and an empty decision phase).  Output: gpurun_out/lds_attrib.txt
root="$(cd "$(dirname "$0")/.." && pwd)"
o=$root/gpurun_out/lds_attrib.txt
: > "$o"
for v in tree exp9 exp8 exp1 hitpad; do
  if [ "$v" = tree ]; then unset CAMMIQ_LIB; else export CAMMIQ_LIB=$root/variants/libcammiq_$v.so; fi
  echo "== $v" >> "$o"
  "$root/tools/pmc.sh" "l_$v" SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS >> "$o" 2>&1
done
cat "$o"
