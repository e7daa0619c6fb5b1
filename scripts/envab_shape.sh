#!/bin/bash
# On the GPU box: interleaved rounds of scripts/shape_bench.py over settings of ONE environment variable, library as built:
#   scripts/envab_shape.sh VAR "v1 v2 ..." rounds shape-args...
cd ${GRAFT_REPO_ROOT:-/root/repo}
var=$1; vals=$2; rounds=${3:-3}; shift 3
for i in $(seq $rounds); do
  for v in $vals; do
    printf '%-14s ' "$var=$v"
    env $var=$v timeout 300 python scripts/shape_bench.py "$@" 2>&1 | tail -1
  done
done
