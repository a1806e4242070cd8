#!/bin/bash
# usage: ab_opts.sh "opt1=1" "" "opt2=1,opt3=2" ...
for o in "$@"; do
  ZEN_HIP_OPTIONS="$o" python bench.py --workload offline_batch --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/ab_v.json 2> gpurun_out/ab_v.err || tail -3 gpurun_out/ab_v.err
  python - "${o:-default}" <<'PY'
import json, sys
d = json.load(open("gpurun_out/bench_detail.json"))
k = d.get("kernels", {})
print("%-22s step %.3f ms |" % (sys.argv[1], d.get("ms_per_step", 0)), " ".join("%s %.3f" % (n, v["ms_per_step"]) for n, v in sorted(k.items())))
PY
done
