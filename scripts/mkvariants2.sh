#!/bin/bash
# Build variants of libtlsan_hip.so whose k_fwd_bwd translation units (d = 64 / 128 / 128w4) get extra compiler flags
# (HERE; hipcc cross-compiles):   scripts/mkvariants2.sh name:"flags" name2:"flags" ...
# -> ab_run/<name>.so (git-ignored, shipped by gpurun; emptied when the experiment is over).  `head:` builds the last
# commit's tree instead of the working tree.  Run them on ONE box with scripts/abrun.sh.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
cd "$R"
mkdir -p ab_run
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  if [ "$name" = "head" ]; then
    rm -rf /tmp/headtree && git worktree add -f /tmp/headtree HEAD -q
    (cd /tmp/headtree && python -c "from tlsan_amd import build; build.build()" > /dev/null)
    cp /tmp/headtree/tlsan_amd/libtlsan_hip.so ab_run/head.so
    git worktree remove --force /tmp/headtree; git worktree prune
    echo "built ab_run/head.so [git HEAD]"; continue
  fi
  js=$(python3 -c "
import json,sys
f=sys.argv[1].split()
print(json.dumps({'tlsan_attn_d64.hip': f, 'tlsan_attn_d128.hip': f, 'tlsan_attn_d128w4.hip': f}))" "$flags")
  rm -f tlsan_amd/csrc/build/tlsan_attn_d64.o tlsan_amd/csrc/build/tlsan_attn_d128.o tlsan_amd/csrc/build/tlsan_attn_d128w4.o
  TLSAN_SOURCE_FLAGS="$js" python -c "from tlsan_amd import build; build.build()" > /dev/null
  cp tlsan_amd/libtlsan_hip.so ab_run/$name.so
  echo "built ab_run/$name.so  [$flags]"
done
# leave the tree's own library as the default build
rm -f tlsan_amd/csrc/build/tlsan_attn_d64.o tlsan_amd/csrc/build/tlsan_attn_d128.o tlsan_amd/csrc/build/tlsan_attn_d128w4.o
python -c "from tlsan_amd import build; build.build()" > /dev/null
