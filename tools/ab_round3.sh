cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu -k "offline or sharded or config or golden or persistent or largest or long_hops or hpr_params or hard_mask_outputs or blocking or anticausal or drain or median or mfilt or fft" 2>&1 | tail -3
for opt in "" "no_median_bits=1"; do
  echo "== offline_batch $opt"; ZEN_HIP_OPTIONS="$opt" python bench.py --workload offline_batch --steps 10 --warmup 3 --no-cpu-baseline | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print(j['ms_per_step'], j['x_realtime'], {k: round(v['ms_per_step'],3) for k,v in j['kernels'].items()})"
done
echo "== offline_long"; python bench.py --workload offline_long --steps 10 --warmup 3 --no-cpu-baseline | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print(j['ms_per_step'], j['x_realtime'], {k: round(v['ms_per_step'],3) for k,v in j['kernels'].items()})"
python bench.py --no-cpu-baseline --no-realtime --no-legs | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print(j['value'], j['ms_per_step'], j['roofline'])"
