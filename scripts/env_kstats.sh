#!/bin/bash
# On the GPU box: kernel-trace stats of one shape of scripts/shape_bench.py under settings of one environment variable:
#   scripts/env_kstats.sh VAR "v1 v2" shape-args...
R=${GRAFT_REPO_ROOT:-/root/repo}
var=$1; vals=$2; shift 2
cd /tmp && export TMPDIR=/tmp
for v in $vals; do
  export $var=$v
  rm -rf /tmp/ks_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$v -- python3 $R/scripts/shape_bench.py "$@" > /tmp/ks_$v.log 2>&1
  echo "== $var=$v: $(grep 'us/step' /tmp/ks_$v.log | tail -1)"
  python3 $R/scripts/kstats.py /tmp/ks_$v 7
done
