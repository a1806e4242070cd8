#!/bin/bash
# A/B of variant builds of the library (ZEN_HIP_VARIANT=<name> python zen_amd/build.py -> zen_amd/libzen_hip_<name>.so) on the
# offline batch: per-kernel ms per step of both passes.  Usage on the GPU box: tools/ab_variants.sh "" twpre pad4 ...
# ("" = the shipped library); ZEN_HIP_OPTIONS is passed through.
mkdir -p gpurun_out
for v in "$@"; do
  so=zen_amd/libzen_hip${v:+_$v}.so
  [ -f "$so" ] || { echo "$so missing"; continue; }
  ZEN_HIP_SO=$so python bench.py --workload offline_batch --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/ab_v.json 2> gpurun_out/ab_v.err || tail -3 gpurun_out/ab_v.err
  python - "${v:-base}" <<'PY'
import json, sys
d = json.load(open("gpurun_out/bench_detail.json"))
k = d.get("kernels", {})
print("%-10s step %.3f ms |" % (sys.argv[1], d.get("ms_per_step", 0)), " ".join("%s %.3f" % (n, v["ms_per_step"]) for n, v in sorted(k.items())))
PY
done
