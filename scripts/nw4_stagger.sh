#!/bin/bash
# 8-sample workgroups with every second one started late (TLSAN_STAGGER = mode << 8 | units of 8 k cycles)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() {
  TLSAN_NW4=$1 TLSAN_STAGGER=$2 timeout 300 python3 bench.py --no-cpu-baseline --accuracy-steps 0 --also-bf16 0 2>&1 | grep '"metric"' | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('NW4=$1 STAGGER=$2 step %.2f us  k_fwd_bwd(events) %.2f us  loss %s' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3, d['final_loss']))"
}
run 0 0
run 2 0
for mode in 0 256 512; do for n in 1 2 3 4; do run 2 $((mode + n)); done; done
run 2 0
