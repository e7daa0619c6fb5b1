#!/bin/bash
# On the GPU box: interleaved rounds of scripts/shape_bench.py over every ab_run/*.so (TLSAN_LIB_PATH): scripts/abshape.sh rounds shape-args...
cd ${GRAFT_REPO_ROOT:-/root/repo}
rounds=${1:-3}; shift
for i in $(seq $rounds); do
  for so in ab_run/*.so; do
    printf '%-10s ' "$(basename $so .so)"
    TLSAN_LIB_PATH=$PWD/$so timeout 300 python scripts/shape_bench.py "$@" 2>&1 | grep us/step | tail -1
  done
done
