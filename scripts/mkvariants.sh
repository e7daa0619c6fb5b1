#!/bin/bash
# Build variants of libtlsan_hip.so that differ in compile-time toggles of the fused kernel (HERE; hipcc cross-compiles):
#   scripts/mkvariants.sh stamps:"-DTLSAN_STAMPS=1" name2:"-DOTHER=2 ..." ...
# -> ab_run/<name>.so (git-ignored, shipped by gpurun).  `base` (no flags) is always built.  Run them on ONE box with
#   scripts/abrun.sh (interleaved rounds in one gpurun call).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
cd "$R"
mkdir -p ab_run
build() {  # name flags
  rm -f tlsan_amd/csrc/build/*.o
  TLSAN_HIPCC_EXTRA="$2" python -c "from tlsan_amd import build; build.build()" > /dev/null
  cp tlsan_amd/libtlsan_hip.so ab_run/$1.so
  echo "built ab_run/$1.so  [$2]"
}
for spec in "$@"; do build "${spec%%:*}" "${spec#*:}"; done
build base ""
