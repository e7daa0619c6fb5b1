#!/bin/bash
# On the GPU box: interleaved bench rounds over settings of ONE environment variable with the library as built:
#   scripts/envab.sh VAR "v1 v2 ..." [rounds] [bench args...]
cd ${GRAFT_REPO_ROOT:-/root/repo}
var=$1; vals=$2; rounds=${3:-3}; shift 3
for i in $(seq $rounds); do
  for v in $vals; do
    env $var=$v timeout 300 python bench.py --no-cpu-baseline --accuracy-steps 0 --also-bf16 0 "$@" 2>&1 | grep '"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-10s step %.2f us  k_fwd_bwd %.2f us  loss %s' % ('$var=$v', d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3, d['final_loss']))"
  done
done
