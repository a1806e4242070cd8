#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
for opt in "" "rt_fused_diag=1" "rt_fused_diag=2" "no_median47_dpp=1"; do
ZEN_HIP_OPTIONS="$opt" python3 bench.py --no-cpu-baseline --no-realtime | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('$opt', j['value'], j['kernel_ms_per_step'])"
done
