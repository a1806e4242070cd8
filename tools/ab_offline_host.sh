#!/bin/bash
# kernels-alone and wall time of the one-hour host-vector pipeline with / without the synthesis in runs for its pass 1
CFGS=("$@")
[ ${#CFGS[@]} -eq 0 ] && CFGS=(no_istft_runs=2 no_istft_runs=0)
for cfg in "${CFGS[@]}"; do
  ZEN_HIP_OPTIONS=$cfg python bench.py --workload offline_host --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/ab_h.json 2> gpurun_out/ab_h.err
  python - "$cfg" <<EOF
import json,sys
d=json.load(open("gpurun_out/bench_detail.json"))
print(sys.argv[1], "wall_ms", round(d.get("wall_ms",0),3), "kernels_alone_ms", round(d.get("compute_ms",0),3))
EOF
done
