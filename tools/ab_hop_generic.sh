#!/bin/bash
# ab_hop_generic.sh -- the single-hop kernels on the HARMONIC output (the builds that carry every mask variant), shipped build and
# zen_amd/libzen_hip_hop*.so variants, per launch and resident
cd "$(dirname "$0")/.."
HOPS=${1:-3000}
mkdir -p gpurun_out
python -m pytest tests/test_gpu_round5.py tests/test_gpu_round4.py tests/test_gpu_parity.py -q -x -k "median_single_hops or median_resident_kernel_in_both or resident or hop_by_hop" > gpurun_out/hop_generic_tests.log 2>&1
tail -3 gpurun_out/hop_generic_tests.log
for so in zen_amd/libzen_hip.so zen_amd/libzen_hip_hop*.so; do
	[ -e "$so" ] || continue
	g++ -O2 -std=c++17 -I include tools/rt_latency.cpp -o /tmp/rtl_v -L zen_amd -l:$(basename $so) -Wl,-rpath,$PWD/zen_amd || continue
	for rep in 1 2; do
		echo "{\"variant\": \"$so\"}"
		ZEN_RT_OUTPUT=H /tmp/rtl_v $HOPS
		ZEN_RT_OUTPUT=H ZEN_RT_RESIDENT=100 /tmp/rtl_v $HOPS
		echo "{\"option\": \"no_hop_lat=1\"}"
		ZEN_RT_OPT=no_hop_lat=1 ZEN_RT_OUTPUT=H /tmp/rtl_v $HOPS
		ZEN_RT_OPT=no_hop_lat=1 ZEN_RT_OUTPUT=H ZEN_RT_RESIDENT=100 /tmp/rtl_v $HOPS
	done
done > gpurun_out/hop_generic.jsonl 2>&1
grep '"hop": 256\|"hop": 512\|"hop": 1024\|variant\|option' gpurun_out/hop_generic.jsonl | grep -v '"sse": 1' | cut -c1-200
