cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python tools/fuzz_parity.py --seconds 150 --seed 301 2>&1 | tail -2
python bench.py > gpurun_out/b2.json 2> gpurun_out/b2.err
python - <<'PY'
import json
j=json.load(open('gpurun_out/b2.json'))
print('value', j['value'], 'ms', j['ms_per_step'], j['kernel_ms_per_step'], 'frac', j['roofline']['frac'])
print('three', j['three_kernel_path']['hops_per_s'], j['three_kernel_path']['kernel_ms_per_step'])
print('all_outputs', j['all_outputs']['value'], j['all_outputs']['kernel_ms_per_step'])
print('sse', j['sse_block']['value'], {k: round(v['ms_per_step'],3) for k,v in j['sse_block']['kernels'].items()})
print('ob', j['offline_batch']['x_realtime'], j['offline_batch']['ms_per_step'], {k: round(v['ms_per_step'],3) for k,v in j['offline_batch']['kernels'].items()})
print('ol', j['offline_long']['x_realtime'], j['offline_long']['ms_per_step'], {k: round(v['ms_per_step'],3) for k,v in j['offline_long']['kernels'].items()})
print('rt', j['realtime']['per_hop_us_by_hop'])
PY
