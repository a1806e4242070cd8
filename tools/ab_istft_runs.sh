#!/bin/bash
# A/B of the offline steps: synthesis in runs (istft_run_kernel / istft_run_wide_kernel) against the synthesis + overlap-add
# launches.  "no_istft_runs": 1 = neither pass, 2 = not for nfft >= 2048.  Usage: tools/ab_istft_runs.sh [workload] [cfg ...]
W=${1:-offline_batch}; shift
CFGS=("$@"); [ ${#CFGS[@]} -eq 0 ] && CFGS=("no_istft_runs=1" "no_istft_runs=2" "no_istft_runs=0")
for cfg in "${CFGS[@]}"; do
  ZEN_HIP_OPTIONS=$cfg python bench.py --workload $W --steps 10 --warmup 2 --detail > gpurun_out/ab_r.json 2> gpurun_out/ab_r.err
  python - "$cfg" <<EOF
import json,sys
d=json.loads(open("gpurun_out/ab_r.json").read().strip().splitlines()[-1])
k=d.get("kernels") or {}
print(sys.argv[1], round(d.get("ms_per_step"),4), round(d.get("x_realtime")), d.get("checksum"), {a:round(b["ms_per_step"],4) for a,b in k.items() if "istft" in a or "finalize" in a or "freq" in a})
EOF
done
