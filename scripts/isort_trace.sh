R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for m in 1073741824 65536; do
  TLSAN_ISORT_MIN=$m rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04_isort_d128_k$m -- python3 $R/scripts/shape_bench.py d=128 Ls=10 B=4096 U=10000000 I=5000000 C=10000 > $R/gpurun_out/r04_isort_d128_k$m.log 2>&1
  echo "--- kernel stats 10M/5M/10k d=128, TLSAN_ISORT_MIN=$m"; python3 $R/scripts/kstats.py $R/gpurun_out/r04_isort_d128_k$m 9; tail -1 $R/gpurun_out/r04_isort_d128_k$m.log
done
python3 $R/scripts/ktimeline2.py $R/gpurun_out/r04_isort_d128_k65536 2>/dev/null | head -40
