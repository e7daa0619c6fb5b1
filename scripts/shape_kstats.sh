#!/bin/bash
# On the GPU box: kernel-trace stats of one shape of scripts/shape_bench.py for every ab_run/*.so: scripts/shape_kstats.sh shape-args...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for so in $R/ab_run/*.so; do
  n=$(basename $so .so)
  export TLSAN_LIB_PATH=$so
  rm -rf /tmp/ks_$n
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$n -- python3 $R/scripts/shape_bench.py "$@" > /tmp/ks_$n.log 2>&1
  echo "== $n: $(tail -1 /tmp/ks_$n.log)"
  python3 $R/scripts/kstats.py /tmp/ks_$n 6
done
