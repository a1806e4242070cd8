cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python tools/fuzz_parity.py --seconds 100 --seed 302 2>&1 | tail -1
for opt in ""; do
  echo "== offline_batch $opt"; ZEN_HIP_OPTIONS="$opt" python bench.py --workload offline_batch --steps 10 --warmup 3 --no-cpu-baseline | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print(j['ms_per_step'], j['x_realtime'], {k: round(v['ms_per_step'],3) for k,v in j['kernels'].items()})"
  echo "== offline_long $opt"; ZEN_HIP_OPTIONS="$opt" python bench.py --workload offline_long --steps 10 --warmup 3 --no-cpu-baseline | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print(j['ms_per_step'], j['x_realtime'], {k: round(v['ms_per_step'],3) for k,v in j['kernels'].items()})"
done
