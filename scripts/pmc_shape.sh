#!/bin/bash
# PMC passes on the fused kernel at one shape (on the GPU box): scripts/pmc_shape.sh tag shape-args...
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
run() { n=$1; shift; rocprofv3 --pmc $PMC --output-format csv -d $R/gpurun_out/${tag}_$n -- python3 $R/scripts/pmc_run.py "$@" > $R/gpurun_out/${tag}_$n.log 2>&1; }
PMC="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_IFETCH" run a "$@"
PMC="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU" run b "$@"
PMC="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_LDS" run e "$@"
PMC="SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQC_ICACHE_REQ SQC_ICACHE_MISSES" run c "$@"
cd $R
python3 scripts/pmc_summary.py gpurun_out/${tag}_a gpurun_out/${tag}_b gpurun_out/${tag}_e gpurun_out/${tag}_c > gpurun_out/${tag}_summary.txt 2>&1
grep -A40 "k_fwd_bwd" gpurun_out/${tag}_summary.txt | head -50
