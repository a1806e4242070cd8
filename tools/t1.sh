cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu -k "median or mfilt or offline or golden or mask_bits" 2>&1 | tail -3
for so in "" "zen_amd/libzen_hip_base.so"; do
echo "== $so"
ZEN_HIP_SO=$so python tools/bench_median.py --suite path --nonneg 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    j=json.loads(l); print(j['rows'],j['cols'],j['filter_len'],j['direction'],round(j['ms'],4),round(j['frac_of_8TBps'],3))"
ZEN_HIP_SO=$so python bench.py --workload offline_batch --steps 20 --warmup 3 --no-cpu-baseline | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print(j['ms_per_step'], j['x_realtime'], {k: round(v['ms_per_step'],3) for k,v in j['kernels'].items()})"
done
