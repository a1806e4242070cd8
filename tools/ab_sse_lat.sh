#!/bin/bash
# ab_sse_lat.sh -- the single-hop SSE kernels in both layouts (rt_sse_lat.hip / rt_sse.hip, option "no_sse_lat") on the box:
# parity tests first, then tools/rt_latency.cpp per launch and resident, with the phase stamps of a hop of either kind.
#   gpurun -- tools/ab_sse_lat.sh [hops]        -> gpurun_out/sse_lat_ab.jsonl
cd "$(dirname "$0")/.."
HOPS=${1:-3000}
mkdir -p gpurun_out
g++ -O2 -std=c++17 -I include tools/rt_latency.cpp -o /tmp/rtl -L zen_amd -lzen_hip -Wl,-rpath,$PWD/zen_amd || exit 1
python -m pytest tests/test_gpu_round5.py -q -x -k "sse_single_hops or sse_resident_kernel_in_both" 2>&1 | tail -4
python -m pytest tests/test_gpu_parity.py tests/test_gpu_round4.py tests/test_gpu_round5.py -q -x -k "sse" 2>&1 | tail -4
for opt in no_sse_lat=0 no_sse_lat=1 no_sse_lat=0; do
	echo "{\"option\": \"$opt\"}"
	ZEN_RT_ONLY_SSE=1 ZEN_RT_OPT=$opt /tmp/rtl $HOPS --stamps
	ZEN_RT_ONLY_SSE=1 ZEN_RT_OPT=$opt ZEN_RT_RESIDENT=100 /tmp/rtl $HOPS
	ZEN_RT_ONLY_SSE=1 ZEN_RT_OPT=$opt ZEN_RT_RESIDENT=100 /tmp/rtl $HOPS --stamps
done > gpurun_out/sse_lat_ab.jsonl 2>&1
grep -v '"hop": 2048' gpurun_out/sse_lat_ab.jsonl | cut -c1-300
# variant builds of rt_sse_lat.hip (zen_amd/libzen_hip_lat*.so: other values-per-thread choices), new layout only
for so in zen_amd/libzen_hip_lat*.so; do
	[ -e "$so" ] || continue
	g++ -O2 -std=c++17 -I include tools/rt_latency.cpp -o /tmp/rtl_v -L zen_amd -l:$(basename $so) -Wl,-rpath,$PWD/zen_amd || continue
	for rep in 1 2; do
		echo "{\"variant\": \"$so\"}"
		ZEN_RT_ONLY_SSE=1 /tmp/rtl_v $HOPS --stamps
		ZEN_RT_ONLY_SSE=1 ZEN_RT_RESIDENT=100 /tmp/rtl_v $HOPS
	done
done > gpurun_out/sse_lat_variants.jsonl 2>&1
grep -v '"hop": 2048' gpurun_out/sse_lat_variants.jsonl | cut -c1-330
