cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
mkdir -p gpurun_out/po
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/po/st --output-format csv -- python3 bench.py --workload offline_batch --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/po/run.log 2>&1
find gpurun_out/po/st -name '*kernel_stats.csv' -exec cp {} gpurun_out/po/offline_batch_kernel_stats.csv \;
rm -rf gpurun_out/po/st
head -30 gpurun_out/po/offline_batch_kernel_stats.csv | cut -c1-200
