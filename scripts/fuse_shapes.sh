#!/bin/bash
# On the GPU box: the lazy update as trailing blocks of the row-sum launch (TLSAN_TAIL_FUSE=1, the default) against a launch
# of its own (=0), interleaved per shape:   scripts/fuse_shapes.sh > gpurun_out/rNN_tail_fuse_shapes.txt
cd ${GRAFT_REPO_ROOT:-/root/repo}
while read -r shape; do
  [ -z "$shape" ] && continue
  for rep in 1 2; do for f in 0 1; do
    echo -n "fuse=$f  "; TLSAN_TAIL_FUSE=$f timeout 600 python scripts/shape_bench.py $shape 2>&1 | tail -1
  done; done
done <<'SHAPES'
d=64 Ls=10 B=32 U=2010 I=1723 C=226
d=128 Ls=10 B=1024 U=1659 I=1583 C=53
d=128 Ls=10 B=256 U=1659 I=1583 C=53
d=128 Ls=10 B=4096
d=64 Ls=10 B=4096
d=64 Ls=10 B=8192
d=128 Ls=10 B=8192
d=256 Ls=10 B=4096
d=128 Ls=10 B=4096 U=10000000 I=5000000 C=10000
d=256 Ls=90 B=4096 U=10000000 I=5000000 C=10000
SHAPES
