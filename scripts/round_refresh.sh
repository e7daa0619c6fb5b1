#!/bin/bash
# One gpurun call that regenerates everything profiles/ holds for a round: the full GPU test suite, the profile set of
# scripts/refresh_profiles.sh (bench under a kernel trace, PMC passes in fp32 / bf16 tables / bf16 operands, traffic.json),
# the in-kernel stamps (fp32, bf16, streamed), the other shapes, PMC + register reports of the d = 256 kernels, the
# sharded static step (kernel stats, bench line).  Needs ab_run/stamps.so (scripts/mkvariants.sh stamps:"-DTLSAN_STAMPS=1").
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-r05}
cd $R
timeout 3000 python3 -m pytest tests -m gpu -x -q > gpurun_out/${tag}_pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/${tag}_pytest_gpu.log
timeout 1500 bash scripts/refresh_profiles.sh $tag
cd $R
TLSAN_LIB_PATH=$R/ab_run/stamps.so timeout 300 python3 scripts/stamps.py > gpurun_out/${tag}_stamps.txt 2>&1
TLSAN_LIB_PATH=$R/ab_run/stamps.so MM=bf16 TD=bf16 timeout 300 python3 scripts/stamps.py > gpurun_out/${tag}_stamps_bf16.txt 2>&1
TLSAN_LIB_PATH=$R/ab_run/stamps.so timeout 300 python3 scripts/stamps.py d=128 Ls=90 U=35896 I=28589 C=15 > gpurun_out/${tag}_stamps_streamed.txt 2>&1
timeout 1500 bash scripts/shapes_all.sh > gpurun_out/${tag}_shapes.txt 2>&1
timeout 600 bash scripts/pmc_shape.sh ${tag}_pmc_d256 d=256 Ls=10 > /dev/null 2>&1
timeout 600 bash scripts/pmc_shape.sh ${tag}_pmc_c5 d=256 Ls=90 U=10000000 I=5000000 C=10000 > /dev/null 2>&1
timeout 600 bash scripts/pmc_shape.sh ${tag}_pmc_streamed d=128 Ls=90 U=35896 I=28589 C=15 > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_shard_static -- python3 $R/scripts/shard_static_prof.py > $R/gpurun_out/${tag}_shard_static.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_streamed_stats -- python3 $R/scripts/shape_bench.py d=128 Ls=90 B=4096 U=35896 I=28589 C=15 > $R/gpurun_out/${tag}_streamed_stats.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_c5_stats -- python3 $R/scripts/shape_bench.py d=256 Ls=90 B=4096 U=10000000 I=5000000 C=10000 > $R/gpurun_out/${tag}_c5_stats.log 2>&1
cd $R
python3 scripts/kstats.py gpurun_out/${tag}_shard_static 14 > gpurun_out/${tag}_sharded_static_kernel_stats.txt 2>&1
{ echo "# Movies-TV shape, Ls = 90"; python3 scripts/kstats.py gpurun_out/${tag}_streamed_stats 8; echo "# C5 shape"; python3 scripts/kstats.py gpurun_out/${tag}_c5_stats 10; } > gpurun_out/${tag}_streamed_kernel_stats.txt 2>&1
AHEAD=2 timeout 300 python3 bench.py --force-sharded --no-cpu-baseline --accuracy-steps 0 2>/dev/null | tail -1 > gpurun_out/${tag}_sharded_bench_line.json
timeout 300 python3 scripts/eval_bench.py > gpurun_out/${tag}_eval_bench.txt 2>&1
head -12 gpurun_out/${tag}_sharded_static_kernel_stats.txt
head -12 gpurun_out/${tag}_stamps.txt
cut -c1-300 gpurun_out/${tag}_sharded_bench_line.json
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
# (the traffic nodes of the C5 / Movies-TV shapes take their kernel durations from the traces above)
python3 scripts/traffic_json.py ${tag#r} gpurun_out/${tag} > gpurun_out/${tag}_traffic.json 2>&1
timeout 900 bash scripts/batch_sweep.sh > gpurun_out/${tag}_batch_sweep.txt 2>&1
