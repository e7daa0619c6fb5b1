#!/bin/bash
# Regenerate the round's profile artefacts on the GPU box (run through gpurun from the repo root):
#   kernel-trace stats of the default bench command, PMC passes (separate runs, counters only),
#   the bench line itself.  Outputs land in gpurun_out/; copy the summaries into profiles/.
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_stats -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/${tag}_bench_under_rocprof.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/${tag}_$c -- python3 $R/scripts/pmc_run.py > /dev/null 2>&1
  # the same in the precisions BASELINE.json configs[2] names: bf16 tables, bf16 tables + bf16 matrix operands
  rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/${tag}_${c}_bf16t -- python3 $R/scripts/pmc_run.py 4096 td=bf16 > /dev/null 2>&1
  rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/${tag}_${c}_bf16mm -- python3 $R/scripts/pmc_run.py 4096 td=bf16 mm=bf16 > /dev/null 2>&1
  # ... and at the two shapes whose step is not the bench's: C5 (tables beyond every cache) and Movies-TV with 90-entry windows
  rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/${tag}_${c}_c5 -- python3 $R/scripts/pmc_run.py 4096 d=256 Ls=90 U=10000000 I=5000000 C=10000 > /dev/null 2>&1
  rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/${tag}_${c}_mtv -- python3 $R/scripts/pmc_run.py 4096 d=128 Ls=90 U=35896 I=28589 C=15 > /dev/null 2>&1
done
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/${tag}_sq1 -- python3 $R/scripts/pmc_run.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/${tag}_sq2 -- python3 $R/scripts/pmc_run.py > /dev/null 2>&1
cd $R
python3 scripts/pmc_summary.py gpurun_out/${tag}_FETCH_SIZE gpurun_out/${tag}_WRITE_SIZE gpurun_out/${tag}_sq1 gpurun_out/${tag}_sq2 > gpurun_out/${tag}_pmc_summary.txt 2>&1
python3 scripts/traffic_json.py ${tag#r} gpurun_out/${tag} > gpurun_out/${tag}_traffic.json 2>&1
python3 bench.py > gpurun_out/${tag}_bench.log 2>&1
tail -1 gpurun_out/${tag}_bench.log > gpurun_out/${tag}_bench_line.json
python3 scripts/kstats.py gpurun_out/${tag}_stats 8
grep -A3 "k_fwd_bwd\|k_apply" gpurun_out/${tag}_pmc_summary.txt | grep -E "k_fwd|k_apply|FETCH|WRITE" | head
cut -c1-400 gpurun_out/${tag}_bench_line.json
