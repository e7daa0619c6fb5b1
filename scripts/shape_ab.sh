#!/bin/bash
# On the GPU box: scripts/shape_bench.py over every ab_libs/*.so for a list of shapes (one process per run).
#   scripts/shape_ab.sh "d=128 Ls=10 B=4096 sess=amazon" "d=128 Ls=90 B=4096" ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
for shape in "$@"; do
  for so in ab_libs/*.so; do
    echo -n "$(basename $so .so)  "
    TLSAN_LIB_PATH=$so timeout 300 python scripts/shape_bench.py $shape 2>&1 | tail -1
  done
done
done
