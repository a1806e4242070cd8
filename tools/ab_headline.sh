#!/bin/bash
# A/B of library builds on the headline block (bench.py --no-legs): value, fused launch ms, roofline frac; all-outputs leg too.
# usage (GPU box): tools/ab_headline.sh "" foldaddr ...
for v in "$@"; do
  so=zen_amd/libzen_hip${v:+_$v}.so
  [ -f "$so" ] || { echo "$so missing"; continue; }
  for outs in P HPR; do
    ZEN_HIP_SO=$so python bench.py --no-legs --steps 60 --warmup 10 --no-cpu-baseline --no-realtime --outputs $outs 2> gpurun_out/ab_h.err | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); r=j['roofline']
print('%-10s %-3s value %.2f M hops/s  step %.4f ms  fused %.4f ms  frac %.4f' % ('${v:-base}','$outs', j['value']/1e6, j['ms_per_step'], r['avg_launch_ms'], r['frac']))"
  done
done
