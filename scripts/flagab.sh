#!/bin/bash
# usage: flagab.sh "<extra flags>"   (on the GPU box)
cd $GRAFT_REPO_ROOT
run() { timeout 300 python bench.py --no-cpu-baseline --accuracy-steps 0 --also-bf16 0 2>&1 | grep '"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"; }
cp tlsan_amd/libtlsan_hip.so /tmp/base.so
rm -f tlsan_amd/csrc/build/tlsan_attn_d128.o
TLSAN_HIPCC_EXTRA="$1" python -c "from tlsan_amd import build; build.build()" 2>&1 | tail -2
cp tlsan_amd/libtlsan_hip.so /tmp/exp.so
for i in 1 2 3; do cp /tmp/base.so tlsan_amd/libtlsan_hip.so; run base; cp /tmp/exp.so tlsan_amd/libtlsan_hip.so; run exp; done
