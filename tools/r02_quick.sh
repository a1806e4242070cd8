#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
OUT=gpurun_out/r02_aa; mkdir -p $OUT
g++ -O2 -std=c++17 -I include tools/rt_latency.cpp -o /tmp/rt_latency -L zen_amd -lzen_hip -Wl,-rpath,$PWD/zen_amd
/tmp/rt_latency 3000 --stamps
timeout 2700 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
