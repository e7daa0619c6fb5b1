#!/bin/bash
# On the GPU box: the default bench N times in fresh processes (is the step time the same in every process?)
cd ${GRAFT_REPO_ROOT:-/root/repo}
n=${1:-10}; shift
for i in $(seq $n); do
  timeout 300 python bench.py --no-cpu-baseline --accuracy-steps 0 --also-bf16 0 "$@" > /tmp/br.out 2> /tmp/br.err
  grep '"metric"' /tmp/br.out | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('run $i: step %.2f us  k_fwd_bwd %.2f us' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3))"
  grep "concurrent_streams" /tmp/br.err | head -8
done
