#!/bin/bash
# A/B of the offline batch step: synthesis in runs (istft_run_kernel: pass 2; istft_run_wide_kernel: pass 1) against the
# synthesis + overlap-add launches, and the run lengths.  "no_istft_runs": 1 = neither pass, 2 = pass 2 only.
python -m pytest tests/test_gpu_round4.py -x -q -k "synthesised_in_runs" 2>&1 | tail -3

for cfg in "no_istft_runs=1" "no_istft_runs=0"; do
  ZEN_HIP_OPTIONS=$cfg python bench.py --workload offline_batch --steps 10 --warmup 2 --detail > gpurun_out/ab_r.json 2> gpurun_out/ab_r.err
  python - "$cfg" <<EOF
import json,sys
d=json.loads(open("gpurun_out/ab_r.json").read().strip().splitlines()[-1])
k=d.get("kernels") or {}
print(sys.argv[1], round(d.get("ms_per_step"),4), round(d.get("x_realtime")), d.get("checksum"), {a:round(b["ms_per_step"],4) for a,b in k.items() if "istft" in a or "finalize" in a})
EOF
done
