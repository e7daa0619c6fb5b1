#!/bin/bash
# kernel-trace stats of a bench run on the GPU box: scripts/kprof.sh <name> [bench args...]
# writes gpurun_out/<name>/ (csv) and prints the top kernels
R=${GRAFT_REPO_ROOT:-/root/repo}
name=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$name -- python3 $R/bench.py --no-cpu-baseline --accuracy-steps 0 "$@" > $R/gpurun_out/$name.log 2>&1
cd $R && python3 scripts/kstats.py gpurun_out/$name 9; tail -1 gpurun_out/$name.log | cut -c1-200
