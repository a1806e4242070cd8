#!/bin/bash
# compare builds of the long-mask median kernel (ZEN_BIG_MINB = workgroups per CU the compiler must allow)
for i in 1 2; do
  for v in B1 B2 B3; do
    for c in "6460 16384 187" "6460 16384 171" "12920 8192 93" "12920 8192 129" "6460 16384 255" "25840 4096 65"; do
      set -- $c
      ZEN_HIP_SO=$PWD/zen_amd/libzen_hip_$v.so python tools/bench_median.py --suite one --rows $1 --cols $2 --len $3 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['filter_len'], round(d['ms'],3), 'ms', round(d['GBps']), 'GB/s')"
    done
  done
done
