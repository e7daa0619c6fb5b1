#!/bin/bash
# On the GPU box: kernel-trace stats of one shape of scripts/shape_bench.py with the library as built (or TLSAN_LIB_PATH):
#   scripts/kstats_shape.sh tag shape-args...     -> prints the step time and the top kernels
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$tag -- python3 $R/scripts/shape_bench.py "$@" > /tmp/ks_$tag.log 2>&1
echo "== $tag: $(grep us/step /tmp/ks_$tag.log | tail -1)"
python3 $R/scripts/kstats.py /tmp/ks_$tag ${KSTATS_TOP:-8}
