#!/bin/bash
# tools/pmc_sweep.sh  -- where does a wave's time go?  Separate rocprofv3 --pmc passes over the fast classify
# kernel (BENCH_ARGS selects the workload), results appended to gpurun_out/pmc_sweep.txt.
root="$(cd "$(dirname "$0")/.." && pwd)"
o=$root/gpurun_out/pmc_sweep.txt
: > "$o"
# a failed pass is recorded with its return code and rocprofv3's own message; the sweep goes on
p() { echo "== $*" >> "$o"; "$root/tools/pmc.sh" "$@" >> "$o" 2>&1 || echo "pass $1 FAILED rc=$?" >> "$o"; }
p s1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS
p s2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM
p s3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAVES
p t1 TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum
p t2 TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_PERMISSION_MISS_sum
# TA block: at most two raw counters per pass and no derived *_avr metric (round 2's pass with TA_BUSY_avr + three
# more TA counters over-subscribed the block's slots: rocprofv3 aborted with error 38 and took the sweep down)
p t3a TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
p t3b TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum
p t4 TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_ATOMIC_sum
p t5 TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum TCC_TAG_STALL_sum
p t6 TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_NC_READ_REQ_sum
cat "$o"
