cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu -k "sse or box" 2>&1 | tail -3
for so in "" "zen_amd/libzen_hip_base.so" "" ; do
ZEN_HIP_SO=$so python bench.py --no-cpu-baseline --no-realtime --steps 20 --leg-steps 10 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); v=j['sse_block']; print(v['value'], v['ms_per_step'], {k:round(x['ms_per_step'],3) for k,x in v['kernels'].items()})"
done
