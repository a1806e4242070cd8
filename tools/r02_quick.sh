#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
OUT=gpurun_out/r02_v; mkdir -p $OUT
timeout 2700 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
for i in 1 2; do python3 bench.py --no-cpu-baseline --no-realtime | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['value'], j['kernel_ms_per_step'], 'median', j['roofline_median']['avg_launch_ms'], j['roofline_median']['frac'], j['three_kernel_path']['kernel_ms_per_step'], j['three_kernel_path']['whole_rows']['kernel_ms_per_step'])"; done
