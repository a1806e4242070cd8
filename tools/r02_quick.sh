#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
OUT=gpurun_out/r02_f; mkdir -p $OUT
for opt in "" "median47_variant=4"; do
  export ZEN_HIP_OPTIONS="$opt"
  rm -rf $OUT/kt
  timeout 200 rocprofv3 --kernel-trace -d $OUT/kt --output-format csv -- python3 tools/bench_median.py --suite one --rows 25840 --cols 4096 --len 47 --iters 60 > $OUT/kt.log 2>&1
  echo "== $opt" >> $OUT/durations.txt
  python3 - $OUT/kt >> $OUT/durations.txt <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "median" in r["Kernel_Name"]]
    print(rows[-1]["Kernel_Name"][:90])
    print("dur_us", " ".join("%.0f" % x for x in d))
PY
done
rm -rf $OUT/kt
cat $OUT/durations.txt
unset ZEN_HIP_OPTIONS
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python3 - <<'PY'
import json
j = json.load(open("gpurun_out/r02_f/bench_default.json"))
print("value", j["value"], "ms/step", j["ms_per_step"], j["kernel_ms_per_step"])
print("roofline_median", j["roofline_median"]["avg_launch_ms"], j["roofline_median"]["frac"], "copyGBps", j["roofline_median"]["device_copy_GBps"])
print("three", j["three_kernel_path"])
print("rt", j["realtime"]["us_per_hop"])
PY
