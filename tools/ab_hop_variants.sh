#!/bin/bash
# ab_hop_variants.sh -- tools/rt_latency.cpp (median path, hop 1024 shown) with the shipped build and with every
# zen_amd/libzen_hip_hop*.so variant build on the same box, per launch and resident, without and with the stamps
cd "$(dirname "$0")/.."
HOPS=${1:-3000}
mkdir -p gpurun_out
for so in zen_amd/libzen_hip.so zen_amd/libzen_hip_hop*.so; do
	[ -e "$so" ] || continue
	g++ -O2 -std=c++17 -I include tools/rt_latency.cpp -o /tmp/rtl_v -L zen_amd -l:$(basename $so) -Wl,-rpath,$PWD/zen_amd || continue
	for rep in 1 2; do
		echo "{\"variant\": \"$so\"}"
		/tmp/rtl_v $HOPS
		ZEN_RT_RESIDENT=100 /tmp/rtl_v $HOPS
		/tmp/rtl_v 500 --stamps | grep phase_us
		ZEN_RT_RESIDENT=100 /tmp/rtl_v 500 --stamps | grep phase_us
	done
done > gpurun_out/hop_lat_variants.jsonl 2>&1
grep '"hop": 1024\|variant' gpurun_out/hop_lat_variants.jsonl | grep -v '"sse": 1' | cut -c1-330
