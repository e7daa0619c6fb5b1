#!/bin/bash
# HERE, after scripts/round_refresh.sh rNN ran through gpurun: copy the summaries the judge reads from gpurun_out/ (scratch)
# into profiles/ (tracked).   scripts/copy_profiles.sh r06
t=${1:-r06}; cd "$(dirname "$0")/.."; G=gpurun_out; P=profiles
f=$(find $G/${t}_stats -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $P/${t}_kernel_stats.csv
for n in bench_line.json pmc_summary.txt pmc_c5_summary.txt pmc_d256_summary.txt pmc_streamed_summary.txt stamps.txt stamps_bf16.txt stamps_streamed.txt \
         shapes.txt sharded_static_kernel_stats.txt streamed_kernel_stats.txt sharded_bench_line.json eval_bench.txt batch_sweep.txt; do
  [ -s $G/${t}_$n ] && cp $G/${t}_$n $P/${t}_$n
done
[ -s $G/${t}_pytest_gpu.log ] && tail -8 $G/${t}_pytest_gpu.log > $P/${t}_pytest_gpu.txt
[ -s $G/${t}_traffic.json ] && python -c "import json,sys; json.load(open('$G/${t}_traffic.json'))" && cp $G/${t}_traffic.json $P/traffic.json
ls -la $P | grep ${t}_ | wc -l
