#!/bin/bash
# A/B of the offline batch step with / without the synthesis in runs of pass 2 (istft_run_kernel), and its run length
python -m pytest tests/test_gpu_round4.py -x -q -k "synthesised_in_runs" 2>&1 | tail -5
for cfg in "no_istft_runs=1" "no_istft_runs=0" "istft_run=12" "istft_run=24" "istft_run=32" "no_istft_runs=1" "no_istft_runs=0"; do
  ZEN_HIP_OPTIONS=$cfg python bench.py --workload offline_batch --steps 10 --warmup 2 --detail > gpurun_out/ab_r.json 2> gpurun_out/ab_r.err
  python - "$cfg" <<EOF
import json,sys
d=json.loads(open("gpurun_out/ab_r.json").read().strip().splitlines()[-1])
k=d.get("kernels") or {}
print(sys.argv[1], round(d.get("ms_per_step"),4), round(d.get("x_realtime")), d.get("checksum"), {a:round(b["ms_per_step"],4) for a,b in k.items() if a.startswith("pass2")}, {a:round(b["frac"],3) for a,b in k.items() if a.startswith("pass2")})
EOF
done
