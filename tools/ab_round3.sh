cd $GRAFT_REPO_ROOT
for so in "" "zen_amd/libzen_hip_local.so" "" "zen_amd/libzen_hip_local.so"; do
  echo "== offline_batch $so"; ZEN_HIP_SO=$so python bench.py --workload offline_batch --steps 10 --warmup 3 --no-cpu-baseline | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print(j['ms_per_step'], j['x_realtime'], {k: round(v['ms_per_step'],3) for k,v in j['kernels'].items()})"
done
