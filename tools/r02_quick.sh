#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
OUT=gpurun_out/r02_ab; mkdir -p $OUT
timeout 2700 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.log
tail -12 $OUT/pytest_gpu.log
