#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
OUT=gpurun_out/r02_ah; mkdir -p $OUT
timeout 2700 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
for w in offline_batch offline_long; do python3 bench.py --workload $w --no-cpu-baseline | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$w', j['value'], j['x_realtime'], j['ms_per_step']); [print('   ',k, round(v['ms_per_step'],3), round(v['frac'],3)) for k,v in j['kernels'].items()]"; done
