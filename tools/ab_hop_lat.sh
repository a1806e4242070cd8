#!/bin/bash
# ab_hop_lat.sh -- the single-hop kernels of the median path in both layouts (rt_hop_lat.hip / rt_fused.hip, option "no_hop_lat")
# on the box: parity tests first, then tools/rt_latency.cpp per launch and resident, with the phase stamps of a hop.
#   gpurun -- tools/ab_hop_lat.sh [hops]        -> gpurun_out/hop_lat_ab.jsonl
cd "$(dirname "$0")/.."
HOPS=${1:-3000}
mkdir -p gpurun_out
g++ -O2 -std=c++17 -I include tools/rt_latency.cpp -o /tmp/rtl -L zen_amd -lzen_hip -Wl,-rpath,$PWD/zen_amd || exit 1
python -m pytest tests/test_gpu_round5.py tests/test_gpu_round4.py tests/test_gpu_parity.py -q -x -k "median_single_hops or median_resident_kernel_in_both or sse or resident or hop_by_hop or publication" > gpurun_out/hop_lat_tests.log 2>&1
tail -6 gpurun_out/hop_lat_tests.log
for opt in no_hop_lat=0 no_hop_lat=1 no_hop_lat=0; do
	echo "{\"option\": \"$opt\"}"
	ZEN_RT_OPT=$opt /tmp/rtl $HOPS --stamps
	ZEN_RT_OPT=$opt ZEN_RT_RESIDENT=100 /tmp/rtl $HOPS
	ZEN_RT_OPT=$opt ZEN_RT_RESIDENT=100 /tmp/rtl $HOPS --stamps
done > gpurun_out/hop_lat_ab.jsonl 2>&1
grep -v '"hop": 2048\|"hop": 4096\|"sse": 1' gpurun_out/hop_lat_ab.jsonl | cut -c1-260
for so in zen_amd/libzen_hip_hop*.so; do
	[ -e "$so" ] || continue
	g++ -O2 -std=c++17 -I include tools/rt_latency.cpp -o /tmp/rtl_v -L zen_amd -l:$(basename $so) -Wl,-rpath,$PWD/zen_amd || continue
	for rep in 1 2; do
		echo "{\"variant\": \"$so\"}"
		/tmp/rtl_v $HOPS --stamps
		ZEN_RT_RESIDENT=100 /tmp/rtl_v $HOPS
	done
done > gpurun_out/hop_lat_variants.jsonl 2>&1
grep -v '"hop": 2048\|"hop": 4096\|"sse": 1' gpurun_out/hop_lat_variants.jsonl | cut -c1-260
