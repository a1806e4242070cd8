cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu -k "median or mfilt or offline or golden or config or sharded or anticausal or params or mask_bits" 2>&1 | tail -3
python bench.py --workload offline_batch --steps 20 --warmup 3 --no-cpu-baseline | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print(j['ms_per_step'], j['x_realtime'], {k: round(v['ms_per_step'],3) for k,v in j['kernels'].items()})"
python bench.py --workload offline_long --steps 20 --warmup 3 --no-cpu-baseline | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print(j['ms_per_step'], j['x_realtime'], {k: round(v['ms_per_step'],3) for k,v in j['kernels'].items()})"
python tools/bench_median.py --suite path --nonneg 2>/dev/null | tail -8
