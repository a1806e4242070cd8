cd $GRAFT_REPO_ROOT
TAG=${1:-r03_e}
bash tools/collect_profiles.sh $TAG > gpurun_out/collect_$TAG.log 2>&1
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$TAG/st --output-format csv -- python3 bench.py --workload offline_batch --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/prof_$TAG/offline_stats_run.log 2>&1
find gpurun_out/prof_$TAG/st -name '*kernel_stats.csv' -exec cp {} gpurun_out/prof_$TAG/bench_offline_batch_kernel_stats.csv \;
rm -rf gpurun_out/prof_$TAG/st
tail -3 gpurun_out/collect_$TAG.log
