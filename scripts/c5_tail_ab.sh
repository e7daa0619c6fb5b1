#!/bin/bash
# On the GPU box: interleaved rounds of the C5 shape over TLSAN_SPEC_UFIRST / TLSAN_SPEC_ITEM_BLOCKS (k_finalize_update's
# user-row workgroups first / item-row workgroups launched at most):  CFGS="1:6144 0:0 ..." scripts/c5_tail_ab.sh rounds [shape-args]
cd ${GRAFT_REPO_ROOT:-/root/repo}
rounds=${1:-3}; shift
[ $# -eq 0 ] && set -- d=256 Ls=90 B=4096 U=10000000 I=5000000 C=10000
for r in $(seq $rounds); do
  for cfg in ${CFGS:-1:6144 0:0}; do
    IFS=: read uf cap <<< "$cfg"
    printf 'ufirst=%s cap=%-5s ' $uf $cap
    TLSAN_SPEC_UFIRST=$uf TLSAN_SPEC_ITEM_BLOCKS=$cap python scripts/shape_bench.py "$@" 2>&1 | tail -1
  done
done
