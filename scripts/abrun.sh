#!/bin/bash
# On the GPU box: interleaved rounds of the bench over every ab_run/*.so (one process per run; same device).
#   scripts/abrun.sh [rounds] [bench args...]
cd ${GRAFT_REPO_ROOT:-/root/repo}
rounds=${1:-3}; shift
cp tlsan_amd/libtlsan_hip.so /tmp/orig.so
for i in $(seq $rounds); do
  for so in ab_run/*.so; do
    n=$(basename $so .so)
    cp $so tlsan_amd/libtlsan_hip.so
    timeout 300 python bench.py --no-cpu-baseline --accuracy-steps 0 --also-bf16 0 "$@" 2>&1 | grep '"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-14s step %.2f us  k_fwd_bwd %.2f us  loss %s' % ('$n', d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3, d['final_loss']))"
  done
done
cp /tmp/orig.so tlsan_amd/libtlsan_hip.so
