#!/bin/bash
# ab_hop_variants.sh -- tools/rt_latency.cpp (median path only, with stamps) against every zen_amd/libzen_hip_hop*.so variant build
cd "$(dirname "$0")/.."
HOPS=${1:-3000}
mkdir -p gpurun_out
for so in zen_amd/libzen_hip_hop*.so; do
	[ -e "$so" ] || continue
	g++ -O2 -std=c++17 -I include tools/rt_latency.cpp -o /tmp/rtl_v -L zen_amd -l:$(basename $so) -Wl,-rpath,$PWD/zen_amd || continue
	for rep in 1 2; do
		echo "{\"variant\": \"$so\"}"
		/tmp/rtl_v $HOPS --stamps
		ZEN_RT_RESIDENT=100 /tmp/rtl_v $HOPS --stamps
	done
done > gpurun_out/hop_lat_variants.jsonl 2>&1
grep '"hop": 1024\|variant' gpurun_out/hop_lat_variants.jsonl | grep -v '"sse": 1' | cut -c1-330
