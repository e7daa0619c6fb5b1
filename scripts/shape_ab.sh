#!/bin/bash
# On the GPU box: one shape of scripts/shape_bench.py over every ab_run/*.so, interleaved: scripts/shape_ab.sh rounds shape-args...
cd ${GRAFT_REPO_ROOT:-/root/repo}
rounds=$1; shift
for i in $(seq $rounds); do
  for so in ab_run/*.so; do
    n=$(basename $so .so)
    echo -n "$n: "; TLSAN_LIB_PATH=$so timeout 300 python scripts/shape_bench.py "$@" 2>&1 | tail -1
  done
done
