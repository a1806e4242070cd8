#!/bin/bash
# ab_lat_flags.sh -- the latency-layout kernels built with other scheduler flags (zen_amd/libzen_hip_hopilp*.so) against the shipped
# build: tools/rt_latency.cpp per launch and resident, every hop
cd "$(dirname "$0")/.."
HOPS=${1:-3000}
mkdir -p gpurun_out
for so in zen_amd/libzen_hip.so zen_amd/libzen_hip_hopilp*.so zen_amd/libzen_hip.so; do
	[ -e "$so" ] || continue
	g++ -O2 -std=c++17 -I include tools/rt_latency.cpp -o /tmp/rtl_v -L zen_amd -l:$(basename $so) -Wl,-rpath,$PWD/zen_amd || continue
	echo "{\"variant\": \"$so\"}"
	/tmp/rtl_v $HOPS | grep -v '"hop": 2048\|"hop": 4096'
	ZEN_RT_RESIDENT=100 /tmp/rtl_v $HOPS | grep -v '"hop": 2048\|"hop": 4096'
done > gpurun_out/lat_flags.jsonl 2>&1
cut -c1-140 gpurun_out/lat_flags.jsonl
