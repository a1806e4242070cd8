cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu -k "offline or golden or blocking or drain or params or stream" 2>&1 | tail -3
for so in "" "zen_amd/libzen_hip_base.so" "" "zen_amd/libzen_hip_base.so"; do
echo "== $so"
ZEN_HIP_SO=$so python bench.py --workload offline_batch --steps 20 --warmup 3 --no-cpu-baseline | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print(j['ms_per_step'], j['x_realtime'], {k: round(v['ms_per_step'],3) for k,v in j['kernels'].items()})"
done
