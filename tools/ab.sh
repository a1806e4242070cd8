#!/bin/bash
# A/B two builds of libzen_hip.so on the default bench, interleaved (DVFS makes single runs drift).
for i in 1 2 3; do
  for v in A B; do
    ZEN_HIP_SO=$PWD/zen_amd/libzen_hip_$v.so python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-realtime 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value']/1e6,2), 'Mhops/s', {k: round(v,3) for k,v in d['kernel_ms_per_step'].items()})"
  done
done
