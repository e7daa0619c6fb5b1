#!/bin/bash
# (superseded by scripts/abrun3.sh, which it now runs: fp32 AND the bf16 leg, every variant through TLSAN_LIB_PATH)
exec "$(dirname "$0")/abrun3.sh" "$@"
