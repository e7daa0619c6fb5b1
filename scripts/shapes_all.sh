#!/bin/bash
# On the GPU box: the train step at the other shapes of BASELINE.json's configs (parity cases, not bench lines).
#   scripts/shapes_all.sh > gpurun_out/rNN_shapes.txt
cd ${GRAFT_REPO_ROOT:-/root/repo}
while read -r shape; do
  [ -z "$shape" ] && continue
  echo "# scripts/shape_bench.py $shape"
  timeout 600 python scripts/shape_bench.py $shape 2>&1 | tail -1
done <<'SHAPES'
d=64 Ls=10 B=32 U=2010 I=1723 C=226
d=128 Ls=10 B=1024 U=1659 I=1583 C=53
d=128 Ls=10 B=4096
d=128 Ls=10 B=4096 sess=amazon
d=64 Ls=10 B=4096
d=64 Ls=10 B=8192
d=128 Ls=10 B=4096 U=35896 I=28589 C=15
d=128 Ls=90 B=4096 U=35896 I=28589 C=15
d=128 Ls=90 B=4096 U=35896 I=28589 C=15 sess=amazon
d=256 Ls=10 B=4096
d=256 Ls=10 B=4096 sess=amazon
d=256 Ls=90 B=4096
d=128 Ls=10 B=4096 U=10000000 I=5000000 C=10000
d=256 Ls=90 B=4096 U=10000000 I=5000000 C=10000
d=128 Ls=10 B=4096 td=bf16 mm=bf16
d=128 Ls=90 B=4096 U=35896 I=28589 C=15 td=bf16 mm=bf16
d=256 Ls=10 B=4096 td=bf16 mm=bf16
d=256 Ls=90 B=4096 td=bf16 mm=bf16
d=256 Ls=90 B=4096 U=10000000 I=5000000 C=10000 td=bf16 mm=bf16
SHAPES
