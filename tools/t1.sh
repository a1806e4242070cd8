cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu -k "mask_bits or offline or golden or config or sharded or soft" 2>&1 | tail -4
for opt in "" "no_median_bits=1"; do
ZEN_HIP_OPTIONS="$opt" python bench.py --workload offline_long --steps 20 --warmup 3 --no-cpu-baseline | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print(j['ms_per_step'], j['x_realtime'], {k: round(v['ms_per_step'],3) for k,v in j['kernels'].items()})"
done
