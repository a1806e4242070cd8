for cfg in "no_istft_runs=1" "no_istft_runs=0" "istft_run=8" "istft_run=4"; do
  ZEN_HIP_OPTIONS=$cfg python bench.py --workload offline_host --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/ab_h.json 2> gpurun_out/ab_h.err
  python - "$cfg" <<EOF
import json,sys
d=json.loads(open("gpurun_out/ab_h.json").read().strip().splitlines()[-1])
print(sys.argv[1], d.get("ms_per_step"), d.get("config",{}).get("offline_host_x_realtime"), json.dumps(d.get("legs") or d.get("roofline"))[:300])
EOF
done
python - <<EOF
import json
d=json.load(open("gpurun_out/bench_detail.json"))
print({k:d.get(k) for k in ("compute_ms","wall_ms","value")})
EOF
