#!/bin/bash
# ab_wide.sh -- hop 2048 / 4096 hop by hop (rt_wide.hip), shipped build and zen_amd/libzen_hip_wide*.so variants, per launch and
# resident, percussive and harmonic output; the long-hop parity tests first
cd "$(dirname "$0")/.."
HOPS=${1:-2000}
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_round5.py tests/test_gpu_round4.py tests/test_gpu_round3.py -q -x -k "long_hop or long_hops" > gpurun_out/wide_tests.log 2>&1
tail -3 gpurun_out/wide_tests.log
for so in zen_amd/libzen_hip.so zen_amd/libzen_hip_wide*.so; do
	[ -e "$so" ] || continue
	g++ -O2 -std=c++17 -I include tools/rt_latency.cpp -o /tmp/rtl_v -L zen_amd -l:$(basename $so) -Wl,-rpath,$PWD/zen_amd || continue
	for rep in 1 2; do
		for o in P H; do
			echo "{\"variant\": \"$so\", \"output\": \"$o\"}"
			ZEN_RT_OUTPUT=$o /tmp/rtl_v $HOPS | grep '"hop": 2048, "sse": 0\|"hop": 4096'
			ZEN_RT_OUTPUT=$o ZEN_RT_RESIDENT=100 /tmp/rtl_v $HOPS | grep '"hop": 2048, "sse": 0\|"hop": 4096'
		done
		/tmp/rtl_v 500 --stamps | grep '"hop": 2048, "sse": 0, "phase\|"hop": 4096, "sse": 0, "phase'
	done
done > gpurun_out/wide_ab.jsonl 2>&1
cut -c1-330 gpurun_out/wide_ab.jsonl
