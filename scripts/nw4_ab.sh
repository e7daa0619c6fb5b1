#!/bin/bash
# On the GPU box: 16-sample (TLSAN_NW4=0) against 8-sample workgroups (TLSAN_NW4=2) of the fused kernel at d = 128:
# parity tests under NW4=2, interleaved bench runs, the small-batch shapes, kernel-trace stats of both.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
tag=${1:-r04_nw4}
TLSAN_NW4=2 timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not readme and not full_scale and not many_uses" > gpurun_out/${tag}_pytest.log 2>&1; echo "pytest(NW4=2) rc=$?"; tail -3 gpurun_out/${tag}_pytest.log
for i in 1 2; do
  for m in 0 2; do
    TLSAN_NW4=$m timeout 300 python3 bench.py --no-cpu-baseline --accuracy-steps 0 --also-bf16 0 2>&1 | grep '"metric"' | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('NW4=$m step %.2f us  k_fwd_bwd(events) %.2f us  loss %s' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3, d['final_loss']))"
  done
done
for m in 0 1 2; do
  echo "--- NW4=$m"
  TLSAN_NW4=$m python3 scripts/shape_bench.py d=128 Ls=10 B=1024 U=1659 I=1583 C=53
  TLSAN_NW4=$m python3 scripts/shape_bench.py d=128 Ls=10 B=2048
  TLSAN_NW4=$m python3 scripts/shape_bench.py d=128 Ls=10 B=256
done
cd /tmp && export TMPDIR=/tmp
for m in 0 2; do
  TLSAN_NW4=$m rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_k$m -- python3 $R/bench.py --no-cpu-baseline --accuracy-steps 0 --also-bf16 0 > $R/gpurun_out/${tag}_k$m.log 2>&1
  echo "--- kernel stats NW4=$m"; python3 $R/scripts/kstats.py $R/gpurun_out/${tag}_k$m 7
done
