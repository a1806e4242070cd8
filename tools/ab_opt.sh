#!/bin/bash
# A/B one zen_hip_set_option switch on the default bench, interleaved (DVFS makes single runs drift).
# usage: tools/ab_opt.sh "<options for B, e.g. no_median47_neighbour=1>" [bench.py args...]
OPT=$1; shift
for i in 1 2 3; do
  for v in A B; do
    if [ $v = B ]; then export ZEN_HIP_OPTIONS=$OPT; else unset ZEN_HIP_OPTIONS; fi
    python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-realtime "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value']/1e6,2), 'Mhops/s', {k: round(v,4) for k,v in d['kernel_ms_per_step'].items()}, (d.get('three_kernel_path') or {}).get('kernel_ms_per_step',{}).get('freq_filter'))"
  done
done
