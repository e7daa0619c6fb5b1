R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04_shard_static -- python3 $R/scripts/shard_static_prof.py > $R/gpurun_out/r04_shard_static.log 2>&1
cd $R
python3 scripts/kstats.py gpurun_out/r04_shard_static 16
python3 scripts/ktimeline2.py gpurun_out/r04_shard_static 2000 40
tail -3 gpurun_out/r04_shard_static.log
