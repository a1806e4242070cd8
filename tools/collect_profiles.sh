#!/bin/bash
# Everything profiles/ holds for one round, in one GPU-box call:  gpurun -- 'tools/collect_profiles.sh r02_a'
# Writes gpurun_out/prof_<tag>/ ; copy what should be judged into profiles/<tag>_*.
TAG=${1:-x}
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG; mkdir -p $OUT
REV=$(cat .build_rev 2>/dev/null || echo unknown)
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python3 bench.py --no-block-fused --no-cpu-baseline --no-realtime > $OUT/bench_default_three_kernel.json 2>> $OUT/bench_default.err
python3 bench.py --outputs HPR --no-cpu-baseline --no-realtime > $OUT/bench_default_all_outputs.json 2>> $OUT/bench_default.err
python3 bench.py --workload offline_batch > $OUT/bench_offline_batch.json 2>> $OUT/bench_default.err
python3 bench.py --workload offline_long > $OUT/bench_offline_long.json 2>> $OUT/bench_default.err
python3 tools/bench_median.py --suite path > $OUT/median_path_shapes.jsonl 2>> $OUT/bench_default.err
# per-kernel durations of the default bench command (the averages must agree with bench.py's HIP events)
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats --output-format csv -- python3 bench.py --no-cpu-baseline --no-realtime > $OUT/stats_run.log 2>&1
find $OUT/stats -name '*kernel_stats.csv' -exec cp {} $OUT/bench_default_kernel_stats.csv \;
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats_off --output-format csv -- python3 bench.py --workload offline_long --steps 5 --warmup 2 --no-cpu-baseline > $OUT/stats_off_run.log 2>&1
find $OUT/stats_off -name '*kernel_stats.csv' -exec cp {} $OUT/bench_offline_long_kernel_stats.csv \;
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats_ob --output-format csv -- python3 bench.py --workload offline_batch --steps 5 --warmup 2 --no-cpu-baseline > $OUT/stats_ob_run.log 2>&1
find $OUT/stats_ob -name '*kernel_stats.csv' -exec cp {} $OUT/bench_offline_batch_kernel_stats.csv \;
rm -rf $OUT/stats $OUT/stats_off $OUT/stats_ob
# hardware counters (separate passes): the fused kernel under the default bench, the 47-tap kernel under the three-kernel leg
timeout 900 tools/pmc_cmd.sh rt_fused_kernel python3 bench.py --steps 3 --warmup 1 --settle-ms 0 --no-cpu-baseline --no-realtime > $OUT/pmc_rt_fused.json 2> $OUT/pmc.err
timeout 900 tools/pmc_cmd.sh median47_dpp_kernel python3 bench.py --no-block-fused --steps 3 --warmup 1 --settle-ms 0 --no-cpu-baseline --no-realtime > $OUT/pmc_median47.json 2>> $OUT/pmc.err
python3 tools/traffic_json.py $OUT/pmc_rt_fused.json "rt_fused_kernel<12,47>" 25840 4096 "$REV" > $OUT/fused_hbm_traffic.json
python3 tools/traffic_json.py $OUT/pmc_median47.json "median47_dpp_kernel<nonneg>" 25840 4096 "$REV" > $OUT/median47_hbm_traffic.json
# micro-benchmarks the design decisions lean on
mkdir -p /tmp/ub
hipcc --offload-arch=gfx950 -O3 tools/ubench_valu3.hip -o /tmp/ub/v3 2>/dev/null && /tmp/ub/v3 > $OUT/ubench_valu3.txt 2>&1
hipcc --offload-arch=gfx950 -O3 tools/ubench_copy.hip -o /tmp/ub/cp 2>/dev/null && /tmp/ub/cp > $OUT/ubench_copy.txt 2>&1
g++ -O2 -std=c++17 -I include tools/rt_latency.cpp -o /tmp/ub/rt -L zen_amd -lzen_hip -Wl,-rpath,$PWD/zen_amd && /tmp/ub/rt 3000 --stamps > $OUT/rt_latency.jsonl 2>&1
for opt in "" "median47_variant=2" "median47_variant=3" "median47_variant=4" "no_median47_dpp=1"; do
  echo "== ZEN_HIP_OPTIONS=$opt" >> $OUT/median47_variants.txt
  ZEN_HIP_OPTIONS="$opt" python3 tools/bench_median.py --suite one --rows 25840 --cols 4096 --len 47 --iters 12 >> $OUT/median47_variants.txt 2>&1
done
ls -la $OUT
