#!/bin/bash
# Deeper PMC passes on the fused kernel (on the GPU box): instruction-cache, per-pipe busy / active cycles, LDS.
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-r02deep}
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/${tag}_counters.txt 2>&1
run() { n=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/${tag}_$n -- python3 $R/scripts/pmc_run.py > $R/gpurun_out/${tag}_$n.log 2>&1; }
run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_IFETCH
run b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU
run c SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_DCACHE_REQ SQC_DCACHE_MISSES SQ_INSTS_SMEM SQ_INSTS_BRANCH
run d SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_INSTS_LDS SQ_LDS_ATOMIC_RETURN SQ_INSTS_VALU_MFMA_MOPS_F32
run e SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVES_EQ_64
run f TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum
run g GRBM_GUI_ACTIVE GRBM_COUNT
cd $R
python3 scripts/pmc_summary.py gpurun_out/${tag}_a gpurun_out/${tag}_b gpurun_out/${tag}_c gpurun_out/${tag}_d gpurun_out/${tag}_e gpurun_out/${tag}_f gpurun_out/${tag}_g > gpurun_out/${tag}_summary.txt 2>&1
grep -A60 "k_fwd_bwd<128, 16, true, false, 0" gpurun_out/${tag}_summary.txt | head -70
grep -i "icache\|ifetch" gpurun_out/${tag}_counters.txt | head -20
tail -3 gpurun_out/${tag}_c.log
