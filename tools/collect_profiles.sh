#!/bin/bash
# Everything profiles/ holds for one build, in one GPU-box call:  gpurun -- 'tools/collect_profiles.sh r03_a'
# Writes gpurun_out/prof_<tag>/ ; copy what should be judged into profiles/<tag>_* (and hbm_traffic.json -> r03_hbm_traffic.json).
TAG=${1:-x}
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG; mkdir -p $OUT
REV=$(cat .build_rev 2>/dev/null || echo unknown)
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python3 bench.py --workload offline_batch > $OUT/bench_offline_batch.json 2>> $OUT/bench_default.err
python3 bench.py --workload offline_long > $OUT/bench_offline_long.json 2>> $OUT/bench_default.err
python3 tools/bench_median.py --suite path --nonneg > $OUT/median_path_shapes.jsonl 2>> $OUT/bench_default.err
# per-kernel durations of the default bench command, legs included (the averages must agree with bench.py's HIP events)
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats --output-format csv -- python3 bench.py --no-cpu-baseline --no-realtime > $OUT/stats_run.log 2>&1
find $OUT/stats -name '*kernel_stats.csv' -exec cp {} $OUT/bench_default_kernel_stats.csv \;
rm -rf $OUT/stats
# the 47-tap kernel back to back for > 1 s: every dispatch's duration
timeout 300 rocprofv3 --kernel-trace -d $OUT/kt --output-format csv -- python3 tools/bench_median.py --suite one --rows 25840 --cols 4096 --len 47 --iters 8000 --nonneg > $OUT/kt_run.log 2>&1
python3 tools/dispatch_durations.py $OUT/kt median47_dpp_kernel > $OUT/median47_dispatch_durations.txt 2>&1
rm -rf $OUT/kt
# hardware counters (separate passes per group), per exact kernel name
B="python3 bench.py --no-legs --steps 3 --warmup 1 --settle-ms 0 --no-cpu-baseline --no-realtime"
timeout 1300 tools/pmc_cmd.sh rt_fused_kernel $B > $OUT/pmc_fused_p.json 2> $OUT/pmc.err
timeout 1300 tools/pmc_cmd.sh rt_fused_kernel $B --outputs HPR > $OUT/pmc_fused_hpr.json 2>> $OUT/pmc.err
timeout 1300 tools/pmc_cmd.sh rt_fused_kernel $B --fused-minb 1 > $OUT/pmc_fused_p_minb2.json 2>> $OUT/pmc.err
timeout 1300 tools/pmc_cmd.sh median47_dpp_kernel $B --no-block-fused > $OUT/pmc_median47_half.json 2>> $OUT/pmc.err
timeout 1300 tools/pmc_cmd.sh median47_dpp_kernel python3 tools/bench_median.py --suite one --rows 25840 --cols 4096 --len 47 --iters 6 --nonneg > $OUT/pmc_median47_whole.json 2>> $OUT/pmc.err
T=$OUT/hbm_traffic.json; rm -f $T
python3 tools/traffic_json.py $T "$REV" $OUT/pmc_fused_p.json "rt_fused_kernel<12, 47, 3, true, true, true>" $((25840*4096)) 25840 4096 "one workgroup per hop, percussive output" >> $OUT/pmc.err 2>&1
python3 tools/traffic_json.py $T "$REV" $OUT/pmc_fused_hpr.json "rt_fused_kernel<12, 47, 3, false, true, true>" $((25840*4096)) 25840 4096 "one workgroup per hop, three outputs" >> $OUT/pmc.err 2>&1
python3 tools/traffic_json.py $T "$REV" $OUT/pmc_fused_p_minb2.json "rt_fused_kernel<12, 47, 1, true, true, true>" $((25840*4096)) 25840 4096 "the register-rich single-hop build run on the block: no scratch -- what the write traffic is without spills" >> $OUT/pmc.err 2>&1
python3 tools/traffic_json.py $T "$REV" $OUT/pmc_median47_half.json "median47_dpp_kernel<true, 0, true>" $((25840*2072)) 25840 4096 "engine launch: bins 0..2048 and 4073..4095 of every row" >> $OUT/pmc.err 2>&1
python3 tools/traffic_json.py $T "$REV" $OUT/pmc_median47_whole.json "median47_dpp_kernel<true, 0, false>" $((25840*4096)) 25840 4096 "whole rows (BASELINE's median metric)" >> $OUT/pmc.err 2>&1
# micro-benchmarks the design decisions lean on
mkdir -p /tmp/ub
g++ -O2 -std=c++17 -I include tools/rt_latency.cpp -o /tmp/ub/rt -L zen_amd -lzen_hip -Wl,-rpath,$PWD/zen_amd && /tmp/ub/rt 3000 --stamps > $OUT/rt_latency.jsonl 2>&1
# the shape sweeps of the reference's three bench harnesses on this build
python3 tools/bench_sweeps.py > $OUT/sweeps.jsonl 2>> $OUT/bench_default.err
ls -la $OUT
