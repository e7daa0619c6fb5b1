#!/bin/bash
# index build alone, counters vs counting sort, at the two large-table shapes
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for shape in "d=128 Ls=10" "d=256 Ls=90"; do
for m in 1073741824 65536; do
  TLSAN_ISORT_MIN=$m rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04_idx_$m -- python3 $R/scripts/index_bench.py $shape B=4096 U=10000000 I=5000000 C=10000 > $R/gpurun_out/r04_idx_$m.log 2>&1
  echo "--- $shape TLSAN_ISORT_MIN=$m: $(grep 'index build' $R/gpurun_out/r04_idx_$m.log)"; python3 $R/scripts/kstats.py $R/gpurun_out/r04_idx_$m 6 | grep -v "k_apply\|elementwise\|k_gidx\|reduce_double"
  rm -rf $R/gpurun_out/r04_idx_$m
done
done
