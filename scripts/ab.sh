#!/bin/bash
# A/B of the bench on ONE GPU box (step times differ by ~1 us between boxes):
#   scripts/ab.sh <commit> [bench args...]   (run HERE: it builds a worktree of <commit> in ./ab_old and
#   sends both trees through one gpurun call, alternating three times)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
c=$1; shift
cd "$R"
rm -rf ab_old && git worktree add -f ab_old "$c" -q
(cd ab_old && python -c "from tlsan_amd import build; build.build(verbose=False)" > /dev/null)
/usr/local/graft/bin/gpurun --timeout 1800 -- "for i in 1 2 3; do for d in ab_old .; do (cd \$d && timeout 300 python bench.py --no-cpu-baseline --accuracy-steps 0 --also-bf16 0 $* 2>&1 | grep '\"metric\"' | python -c \"import json,sys; d=json.loads(sys.stdin.read()); print('\$d', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])\"); done; done" 2>&1 | tail -7
git worktree remove --force ab_old; git worktree prune
