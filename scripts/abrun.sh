#!/bin/bash
# On the GPU box: interleaved rounds of the bench over every ab_run/*.so (fp32 leg).  The in-tree library is never
# overwritten: each run loads its variant through TLSAN_LIB_PATH, and bench.py records which library it measured.
#   scripts/abrun.sh [rounds] [bench args...]
cd ${GRAFT_REPO_ROOT:-/root/repo}
rounds=${1:-3}; shift
for i in $(seq $rounds); do
  for so in ab_run/*.so; do
    n=$(basename $so .so)
    TLSAN_LIB_PATH=$PWD/$so timeout 300 python bench.py --no-cpu-baseline --accuracy-steps 0 --also-bf16 0 "$@" 2>&1 | grep '"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-10s step %.2f us  k_fwd_bwd %.2f us  loss %s' % ('$n', d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3, d['final_loss']))"
  done
done
