#!/bin/bash
# On the GPU box: the train step at batches of 4096 / 8192 / 16384 sequences (d = 128, Electronics-scale tables), fp32 and
# bf16 tables + operands, single-GPU Model and the sharded step at one rank -- the data the weak-scaling operating point of
# the 8-GPU step is chosen from.   scripts/batch_sweep.sh > gpurun_out/rNN_batch_sweep.txt
cd ${GRAFT_REPO_ROOT:-/root/repo}
for B in 4096 8192 16384; do
  for prec in "" "td=bf16 mm=bf16"; do
    timeout 300 python scripts/shape_bench.py d=128 Ls=10 B=$B $prec 2>&1 | tail -1
  done
  echo -n "one-rank sharded step (static rows, plans two ahead), B=$B: "
  AHEAD=2 timeout 300 python bench.py --force-sharded --no-cpu-baseline --accuracy-steps 0 --batch $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f us/step, %.2f M seq/s' % (d['ms_per_step']*1e3, d['value']/1e6))"
done
