#!/bin/bash
# Everything profiles/ holds for one build, in one GPU-box call:  gpurun -- 'tools/collect_profiles.sh r06'
# Writes gpurun_out/prof_<tag>/ ; copy what should be judged into profiles/<tag>_* (hbm_traffic.json and offline_batch_pmc.json also go
# to profiles/<round>_*.json, where bench.py looks for them: <round> = the tag up to its first underscore).
# The build label comes from .build_rev, which zen_amd/build.py writes from `git rev-parse` where the library is built.
TAG=${1:-x}
RND=${TAG%%_*}
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG; mkdir -p $OUT
REV=$(cat .build_rev 2>/dev/null || echo unknown)
echo "$REV" > $OUT/build_rev.txt
cat .build_kernel_rev >> $OUT/build_rev.txt 2>/dev/null
# per-kernel durations of the default bench command, legs included (the averages must agree with bench.py's HIP events)
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats --output-format csv -- python3 bench.py --no-cpu-baseline --no-realtime > $OUT/stats_run.log 2>&1
# (the legs start helper processes that use the GPU too -- tools/offline_host.cpp, tools/rt_latency.cpp -- and each gets a
# stats file of its own: the one of bench.py itself is the one that holds the headline kernel)
cp "$(grep -l 'rt_fused_kernel<12, 47, 3, true, true, true>' $(find $OUT/stats -name '*kernel_stats.csv') | head -1)" $OUT/bench_default_kernel_stats.csv
rm -rf $OUT/stats
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats --output-format csv -- python3 bench.py --workload offline_batch --steps 10 --warmup 2 --no-cpu-baseline > $OUT/stats_ob_run.log 2>&1
find $OUT/stats -name '*kernel_stats.csv' -exec cp {} $OUT/bench_offline_batch_kernel_stats.csv \;
rm -rf $OUT/stats
# the 47-tap kernel back to back for > 1 s through the plain wrapper (no promise): every dispatch's duration
timeout 300 rocprofv3 --kernel-trace -d $OUT/kt --output-format csv -- python3 tools/bench_median.py --suite one --rows 25840 --cols 4096 --len 47 --iters 8000 > $OUT/kt_run.log 2>&1
python3 tools/dispatch_durations.py $OUT/kt median47_dpp_kernel > $OUT/median47_dispatch_durations.txt 2>&1
rm -rf $OUT/kt
# hardware counters (separate passes per group), per exact kernel name
B="python3 bench.py --no-legs --steps 3 --warmup 1 --settle-ms 0 --no-cpu-baseline --no-realtime"
timeout 1300 tools/pmc_cmd.sh rt_fused_kernel $B > $OUT/pmc_fused_p.json 2> $OUT/pmc.err
timeout 1300 tools/pmc_cmd.sh rt_fused_kernel $B --outputs HPR > $OUT/pmc_fused_hpr.json 2>> $OUT/pmc.err
timeout 1300 tools/pmc_cmd.sh median47_dpp_kernel $B --no-block-fused > $OUT/pmc_median47_half.json 2>> $OUT/pmc.err
timeout 1300 tools/pmc_cmd.sh median47_dpp_kernel python3 tools/bench_median.py --suite one --rows 25840 --cols 4096 --len 47 --iters 6 > $OUT/pmc_median47_whole.json 2>> $OUT/pmc.err
timeout 1300 tools/pmc_cmd.sh sse_synth_kernel python3 tools/sse_ab.py > $OUT/pmc_sse_synth.json 2>> $OUT/pmc.err
# the network median kernels on the path shapes (round 6: what bounds the time-direction kernel)
{ for sh in "103360 1024 11" "25840 4096 3" "51680 2048 7" "16384 16384 11"; do set -- $sh; echo "== time direction, $1 x $2, $3 taps"; timeout 600 tools/pmc_cmd.sh median_net_time_kernel python3 tools/bench_median.py --suite one --rows $1 --cols $2 --len $3 --dir time --iters 6; done
  for sh in "103360 1024 13" "51680 2048 23" "16384 16384 11"; do set -- $sh; echo "== frequency direction, $1 x $2, $3 taps (plain wrapper)"; timeout 600 tools/pmc_cmd.sh median_net_freq_kernel python3 tools/bench_median.py --suite one --rows $1 --cols $2 --len $3 --iters 6; done; } > $OUT/pmc_median_net_kernels.txt 2>> $OUT/pmc.err
# every kernel of the offline batch step (valu_issue_frac of its VALU-bound kernels: bench.py offline_valu_issue)
timeout 1300 tools/pmc_cmd.sh _kernel python3 bench.py --workload offline_batch --steps 2 --warmup 1 --settle-ms 0 --no-cpu-baseline > $OUT/pmc_offline_batch_raw.json 2>> $OUT/pmc.err
python3 - $OUT/pmc_offline_batch_raw.json "$REV" > $OUT/offline_batch_pmc.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
d.update({"build": sys.argv[2], "clips": 64, "clip_seconds": 30.0,
          "workload": "python3 bench.py --workload offline_batch (64 x 30 s clips, HPRIOffline 4096/256 hard masks); per-dispatch means"})
print(json.dumps(d, indent=1))
PY
T=$OUT/hbm_traffic.json; rm -f $T
python3 tools/traffic_json.py $T "$REV" $OUT/pmc_fused_p.json "rt_fused_kernel<12, 47, 3, true, true, true>" $((25840*4096)) 25840 4096 "one workgroup per hop, percussive output" >> $OUT/pmc.err 2>&1
python3 tools/traffic_json.py $T "$REV" $OUT/pmc_fused_hpr.json "rt_fused_kernel<12, 47, 3, false, true, true>" $((25840*4096)) 25840 4096 "one workgroup per hop, three outputs" >> $OUT/pmc.err 2>&1
python3 tools/traffic_json.py $T "$REV" $OUT/pmc_median47_half.json "median47_dpp_kernel<true, 0, true>" $((25840*2072)) 25840 4096 "engine launch: bins 0..2048 and 4073..4095 of every row" >> $OUT/pmc.err 2>&1
python3 tools/traffic_json.py $T "$REV" $OUT/pmc_median47_whole.json "median47_dpp_kernel<false, 0, false>" $((25840*4096)) 25840 4096 "whole rows through plain zen_hip_mfilt_run: the build that checks the sign bits of what it stages (BASELINE's median metric)" >> $OUT/pmc.err 2>&1
python3 tools/traffic_json.py $T "$REV" $OUT/pmc_sse_synth.json "sse_synth_kernel<11>" $((51680*2048)) 51680 2048 "config 5: time box + frequency box + Wiener mask + inverse transform per frame" >> $OUT/pmc.err 2>&1
# the records bench.py quotes (traffic, valu_issue_frac) are the ones just collected: put them where it looks for them, then the lines
cp $T profiles/${RND}_hbm_traffic.json
cp $OUT/offline_batch_pmc.json profiles/${RND}_offline_batch_pmc.json
python3 bench.py > $OUT/bench_default_line.json 2> $OUT/bench_default.err
cp gpurun_out/bench_detail.json $OUT/bench_default.json
python3 bench.py --workload offline_batch --detail > $OUT/bench_offline_batch.json 2>> $OUT/bench_default.err
python3 bench.py --workload offline_long --detail > $OUT/bench_offline_long.json 2>> $OUT/bench_default.err
python3 bench.py --workload offline_host --host-variants --detail > $OUT/bench_offline_host.json 2>> $OUT/bench_default.err
# BASELINE's second metric, every listed shape, the sustained protocol of the bench line (plain wrapper, then with the promise)
python3 tools/bench_median.py --suite sustained > $OUT/median_shapes_sustained.jsonl 2>> $OUT/bench_default.err
python3 tools/bench_median.py --suite sustained --nonneg >> $OUT/median_shapes_sustained.jsonl 2>> $OUT/bench_default.err
python3 tools/probe_lowrate.py > $OUT/lowrate_257_taps.jsonl 2>> $OUT/bench_default.err
python3 tools/ab_block_host.py > $OUT/block_host_piece_lengths.txt 2>> $OUT/bench_default.err
tools/ab_offline_host.sh > $OUT/offline_host_cpp_ab.txt 2>> $OUT/bench_default.err
# micro-benchmarks the design decisions lean on
mkdir -p /tmp/ub
g++ -O2 -std=c++17 -I include tools/rt_latency.cpp -o /tmp/ub/rt -L zen_amd -lzen_hip -Wl,-rpath,$PWD/zen_amd && { /tmp/ub/rt 3000 --stamps; ZEN_RT_RESIDENT=100 /tmp/ub/rt 3000; } > $OUT/rt_latency.jsonl 2>&1
# the single-hop kernels of rounds 1-4 (rt_fused.hip / rt_sse.hip single-hop builds) on the same box, for the latency layout's A/B
{ ZEN_RT_OPT=no_sse_lat=1,no_hop_lat=1 /tmp/ub/rt 3000 --stamps; ZEN_RT_OPT=no_sse_lat=1,no_hop_lat=1 ZEN_RT_RESIDENT=100 /tmp/ub/rt 3000; } > $OUT/rt_latency_round4_kernels.jsonl 2>&1
# the harmonic output alone (the single-hop builds that carry every mask variant), both generations of kernels
{ ZEN_RT_OUTPUT=H /tmp/ub/rt 3000; ZEN_RT_OUTPUT=H ZEN_RT_RESIDENT=100 /tmp/ub/rt 3000; echo '{"option": "no_sse_lat=1,no_hop_lat=1"}'; ZEN_RT_OUTPUT=H ZEN_RT_OPT=no_sse_lat=1,no_hop_lat=1 /tmp/ub/rt 3000; ZEN_RT_OUTPUT=H ZEN_RT_OPT=no_sse_lat=1,no_hop_lat=1 ZEN_RT_RESIDENT=100 /tmp/ub/rt 3000; } > $OUT/rt_latency_harmonic.jsonl 2>&1
tools/bin/probe_pcie 2048 > $OUT/probe_pcie.json 2>&1
tools/bin/ubench_rowwrite > $OUT/ubench_rowwrite.jsonl 2>&1
tools/bin/ubench_copy --json > $OUT/ubench_copy.json 2>&1
tools/bin/check_div > $OUT/check_div.txt 2>&1
tools/bin/probe_alloc > $OUT/probe_alloc.jsonl 2>&1
# the shape sweeps of the reference's three bench harnesses on this build
python3 tools/bench_sweeps.py > $OUT/sweeps.jsonl 2>> $OUT/bench_default.err
ls -la $OUT
