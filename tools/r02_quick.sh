#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
OUT=gpurun_out/r02_w; mkdir -p $OUT
timeout 2700 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
for opt in "" "mask_divide=1" "" "mask_divide=1"; do
ZEN_HIP_OPTIONS="$opt" python3 bench.py --no-cpu-baseline --no-realtime | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('$opt', j['value'], j['kernel_ms_per_step'])"
done
