#!/bin/bash
# On the GPU box: the item-side counting sort -- its tests, then the large-table shapes with and without it.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_item_index_from_a_counting_sort or test_user_index_from_a_sort" 2>&1 | tail -15
for m in 1073741824 65536; do
  echo "--- TLSAN_ISORT_MIN=$m"
  TLSAN_ISORT_MIN=$m python3 scripts/shape_bench.py d=128 Ls=10 B=4096 U=10000000 I=5000000 C=10000
  TLSAN_ISORT_MIN=$m python3 scripts/shape_bench.py d=256 Ls=90 B=4096 U=10000000 I=5000000 C=10000
done
cd /tmp && export TMPDIR=/tmp
for m in 1073741824 65536; do
  TLSAN_ISORT_MIN=$m rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04_isort_k$m -- python3 $R/scripts/shape_bench.py d=256 Ls=90 B=4096 U=10000000 I=5000000 C=10000 > $R/gpurun_out/r04_isort_k$m.log 2>&1
  echo "--- kernel stats C5 shape, TLSAN_ISORT_MIN=$m"; python3 $R/scripts/kstats.py $R/gpurun_out/r04_isort_k$m 12
done
