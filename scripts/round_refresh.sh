#!/bin/bash
# One gpurun call that regenerates everything profiles/ holds for a round: the full GPU test suite, the profile set of
# scripts/refresh_profiles.sh, the in-kernel stamps (fp32 and bf16), the sharded static step's kernel stats.
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-r03}
cd $R
timeout 2400 python3 -m pytest tests -m gpu -x -q > gpurun_out/${tag}_pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/${tag}_pytest_gpu.log
timeout 1200 bash scripts/refresh_profiles.sh $tag
cd $R
TLSAN_LIB_PATH=$R/ab_diag/stamps.so timeout 300 python3 scripts/stamps.py > gpurun_out/${tag}_stamps.txt 2>&1
TLSAN_LIB_PATH=$R/ab_diag/stamps.so MM=bf16 TD=bf16 timeout 300 python3 scripts/stamps.py > gpurun_out/${tag}_stamps_bf16.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_shard_static -- python3 $R/scripts/shard_static_prof.py > $R/gpurun_out/${tag}_shard_static.log 2>&1
cd $R
python3 scripts/kstats.py gpurun_out/${tag}_shard_static 14 > gpurun_out/${tag}_sharded_static_kernel_stats.txt 2>&1
head -20 gpurun_out/${tag}_sharded_static_kernel_stats.txt
head -12 gpurun_out/${tag}_stamps.txt
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
