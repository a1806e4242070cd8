#!/bin/bash
# Everything profiles/ holds for one round, in one GPU-box call:  gpurun -- 'tools/collect_profiles.sh g'
# Writes gpurun_out/prof_<tag>/ ; copy what should be judged into profiles/r01_<tag>_*.
TAG=${1:-x}
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG; mkdir -p $OUT
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python3 bench.py --no-block-fused > $OUT/bench_default_three_kernel.json 2>> $OUT/bench_default.err
python3 bench.py --workload offline_batch > $OUT/bench_offline_batch.json 2>> $OUT/bench_default.err
python3 bench.py --workload offline_long > $OUT/bench_offline_long.json 2>> $OUT/bench_default.err
python3 tools/bench_median.py --suite path > $OUT/median_path_shapes.jsonl 2>> $OUT/bench_default.err
# per-kernel durations of the default bench command (the averages must agree with bench.py's HIP events)
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats --output-format csv -- python3 bench.py --no-cpu-baseline --no-realtime > $OUT/stats_run.log 2>&1
find $OUT/stats -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats_off --output-format csv -- python3 bench.py --workload offline_long --steps 5 --warmup 2 --no-cpu-baseline > $OUT/stats_off_run.log 2>&1
find $OUT/stats_off -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats_offline_long.csv \;
rm -rf $OUT/stats $OUT/stats_off
# hardware counters of the dominant kernel (separate passes)
timeout 700 tools/pmc.sh rt_fused_kernel > $OUT/pmc_rt_fused.json 2> $OUT/pmc.err
rm -rf gpurun_out/pmc_*
[ -x tools/bin/ubench_valu2 ] && timeout 60 tools/bin/ubench_valu2 > $OUT/ubench_valu2.txt
ls -la $OUT
